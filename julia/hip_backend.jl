# hip_backend.jl -- methods of SequentialMonteCarlo.jl's own generic functions over libsmchip.so (include/smc_hip.h).
#
# Usage: `include("hip_backend.jl")` after `include("smc_samplers.jl")` in src/SequentialMonteCarlo.jl (INTEGRATION.md).
# UNEXECUTED in the build image (no Julia there).  It is kept mechanical - one `ccall` per entry point, every one with a
# literal `(:name, LIBSMC), Ret, (ArgTypes...)` triple - and tests/test_host.py::test_julia_binding_matches_header
# parses each triple and checks name, arity and every argument type against the declarations in include/smc_hip.h, and that
# the sampler entry points below keep the reference's signatures.
# The Python binding sequential_monte_carlo_amd/_lib.py + smc_samplers.py is the tested twin of what is written here: every
# outer-level number (reweight, the window walk, the tempering bisection, resample!, the random-walk factor) comes from the
# SAME library routine in both hosts, so a Julia-hosted run and a Python-hosted run agree bit for bit.
#
# How the drop-in works.  The reference's samplers are methods on `SMC` (src/smc_samplers.jl:5-27), whose last type
# parameter KT is the type of its `kernel` field.  A GPU-backed sampler is an ordinary `SMC` whose kernel is a `HipKernel`
# (a callable that still returns the reference's random-walk kernel): `HipSMC = SMC{SSM,XT,θT,HipKernel}`.  Methods written for
# `HipSMC` are strictly more specific than the reference's, so `density_tempered(smc,y)`, `smc²(smc,y)`, `smc²!(smc,y,t)`,
# `resample!(smc)`, `rejuvenate!(smc,y,ξ,verbose)`, `exchange!(smc,y,verbose)` - called exactly as in README.md:93-101 and
# examples/inflation_example.jl - reach the GPU path with NO change at the call site, and nothing of the reference is
# overwritten.  `SMC(N,M,model,prior,chain,ess_threshold)` itself builds a HipSMC when `model(θ)` is one of the GPU's
# model families (the more specific constructor below falls through to the reference's otherwise); `HipSMC(...)` asks for it
# explicitly.
#
# Order matters in Julia: types first, then the methods that name them.

const LIBSMC = "libsmchip"            # sequential_monte_carlo_amd/lib/libsmchip.so on LD_LIBRARY_PATH

smc_check(rc) = rc == 0 || error(unsafe_string(ccall((:smc_last_error, LIBSMC), Cstring, ())))

# ---- types -----------------------------------------------------------------------------------------------------------
# device-resident filters: n_theta bootstrap filters of N particles (one opaque smc_handle)
mutable struct HipFilter
    h::Ptr{Cvoid}; N::Int; M::Int; d::Int; id::Cint
    xcache::Union{Nothing,Array{Float64}}      # host copy of the cloud, valid until the filters move again
    wcache::Union{Nothing,Array{Float64}}
    function HipFilter(id::Cint, M::Int, N::Int; seed::UInt64=rand(UInt64), device::Int=0, flags::UInt32=UInt32(0))
        out = Ref{Ptr{Cvoid}}(C_NULL)
        smc_check(ccall((:smc_create, LIBSMC), Cint,
            (Cint, Int64, Int64, Cint, UInt64, Cint, UInt32, Ref{Ptr{Cvoid}}),
            id, M, N, 0, seed, device, flags, out))
        f = new(out[], N, M, id == 3 ? 3 : 1, id, nothing, nothing)
        finalizer(x -> ccall((:smc_destroy, LIBSMC), Cint, (Ptr{Cvoid},), x.h), f)
        f
    end
end
invalidate!(f::HipFilter) = (f.xcache = nothing; f.wcache = nothing; f)

# x and w of ONE filter as the reference returns them: AbstractVectors that materialise on demand (README.md:41,51
# `quantile(x, ...)` keeps working) and are read from the device at most once per filter step
struct HipParticles <: AbstractVector{Float64}; f::HipFilter; m::Int; end
struct HipWeights   <: AbstractVector{Float64}; f::HipFilter; m::Int; end
Base.size(p::Union{HipParticles,HipWeights}) = (p.f.N,)

# theta sharded over the GPUs of a node (one Julia process per GPU; SURVEY 8e)
mutable struct HipComm
    c::Ptr{Cvoid}; rank::Int; world::Int
    function HipComm(id::Vector{UInt8}, rank::Int, world::Int, device::Int)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve id smc_check(ccall((:smc_comm_create, LIBSMC), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
                                        id, rank, world, device, out))
        c = new(out[], rank, world)
        finalizer(x -> ccall((:smc_comm_destroy, LIBSMC), Cint, (Ptr{Cvoid},), x.c), c)
        c
    end
end

# what a GPU-backed sampler carries besides the reference's SMC fields
mutable struct HipSampler
    main::Union{Nothing,HipFilter}     # the online filters: smc.x[m], smc.w[m] are views of its slot m
    prop::Union{Nothing,HipFilter}     # the proposal filters of rejuvenate!
    device_pmmh::Bool                  # prior and model closure are of the enumerated kind: rejuvenate! runs on the device
    prior_family::Vector{Int32}        # SMC_PRIOR_* per component of smc.prior
    prior_par::Matrix{Float64}         # [SMC_PRIOR_NPAR x d_theta] column-major == [d_theta][SMC_PRIOR_NPAR]
    raw_from::Vector{Int32}            # smc.model(theta): raw[k] = theta[raw_from[k]+1] (0-based in C; -1: constant)
    raw_const::Vector{Float64}
    logw::Vector{Float64}              # UN-NORMALISED outer log-weights (smc.ω = reweight(logw)[2] after every public call)
    comm::Union{Nothing,HipComm}       # theta sharded over GPUs: this rank filters the slice lo:hi
    lo::Int; hi::Int
    calls::UInt64                      # evaluation counter -> a fresh Philox seed per batched evaluation
    seed::UInt64
    device::Int
end
next_seed!(s::HipSampler) = (s.calls += 1; (s.seed << 20) + s.calls)

# the kernel field of a GPU-backed SMC: still the reference's random-walk kernel when called (smc_samplers.jl:87-101)
struct HipKernel <: Function
    hs::HipSampler
end
(k::HipKernel)(θ) = random_walk_kernel(θ)
const HipSMC{SSM,XT,θT} = SMC{SSM,XT,θT,HipKernel}
hip(smc::HipSMC) = getfield(smc, :kernel).hs

function fetch_state!(f::HipFilter)
    if f.xcache === nothing
        x = Array{Float64}(undef, f.N, f.M, f.d)       # C layout [d][n_theta][n_x] == column-major N x M x d
        w = Array{Float64}(undef, f.N, f.M)
        GC.@preserve x w smc_check(ccall((:smc_get_state, LIBSMC), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}), f.h, x, w, C_NULL))
        f.xcache = x; f.wcache = w
    end
    f
end
Base.getindex(p::HipParticles, i::Int) = fetch_state!(p.f).xcache[i, p.m, 1]          # first state coordinate
Base.getindex(w::HipWeights, i::Int) = fetch_state!(w.f).wcache[i, w.m]
Base.collect(p::HipParticles) = (x = fetch_state!(p.f).xcache; p.f.d == 1 ? x[:, p.m, 1] : x[:, p.m, :])
Base.collect(w::HipWeights) = fetch_state!(w.f).wcache[:, w.m]

# model families the GPU implements (include/smc_hip.h: SMC_MODEL_*), and their parameter rows
const HipLG = LinearModel{Float64,Float64,Float64,Float64,Float64,Float64}
const HipModels = Union{HipLG,UCSV}
hip_model(m::HipLG) = (Cint(1), Float64[m.A, m.B, m.Q, m.R, m.x0, m.σ0])                      # state_space_models.jl:46-58
hip_model(m::UCSV) = (Cint(3), Float64[m.γ[1], m.γ[2], m.x0, m.log_σ0[1], m.log_σ0[2]])        # :215-222
# [n_raw x M] column-major == the C ABI's [n_theta][n_raw] row-major
hip_rows(models::Vector{<:HipModels}) = reduce(hcat, last.(hip_model.(models)))

function set_models!(f::HipFilter, models::Vector{<:HipModels})
    raw = hip_rows(models)
    GC.@preserve raw smc_check(ccall((:smc_set_params, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Float64}), f.h, raw))
    f
end
function set_streams!(f::HipFilter, streams::Vector{UInt32})                # Philox stream id = GLOBAL theta index
    GC.@preserve streams smc_check(ccall((:smc_set_streams, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{UInt32}), f.h, streams))
    f
end

# ---- particles.jl --------------------------------------------------------------------------------------------------
# bootstrap_filter(N, y, model) -> (x, w, logμ)                      particles.jl:87-105
function bootstrap_filter(N::Int64, y::Float64, model::HipModels)
    f = set_models!(HipFilter(hip_model(model)[1], 1, N), [model])
    logμ = Ref{Float64}()
    smc_check(ccall((:smc_init, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}), f.h, y, logμ))
    return HipParticles(f, 1), HipWeights(f, 1), logμ[]
end

# bootstrap_filter!(x, w, y, model) -> (logμ, w, ess)                particles.jl:107-129
function bootstrap_filter!(states::HipParticles, weights::HipWeights, y::Float64, model::HipModels)
    f = invalidate!(states.f)
    logμ = Ref{Float64}(); ess = Ref{Float64}()
    smc_check(ccall((:smc_step, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), f.h, y, logμ, ess))
    return logμ[], HipWeights(f, states.m), ess[]
end

# log_likelihood(N, y, model) -> (x, w, logZ)                         particles.jl:132-147
function log_likelihood(N::Int64, y::Vector{Float64}, model::HipModels)
    f = set_models!(HipFilter(hip_model(model)[1], 1, N), [model])
    logZ = Ref{Float64}()
    GC.@preserve y smc_check(ccall((:smc_log_likelihood, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        f.h, y, length(y), logZ, C_NULL, C_NULL))
    return HipParticles(f, 1), HipWeights(f, 1), logZ[]
end

# the README loop (README.md:33-61: bootstrap_filter!, then quantile(x, ...) at every observation) as ONE call: the weighted
# quantiles of state coordinate `component` (0-based) at the levels p and / or mean and variance of every coordinate after
# every step, recorded on the device inside the filter loop -> (x, w, logZ, q [np x T], mean [d x T], var [d x T])
function log_likelihood(N::Int64, y::Vector{Float64}, model::HipModels, p::Vector{Float64}; component::Int=0, moments::Bool=true)
    f = set_models!(HipFilter(hip_model(model)[1], 1, N), [model])
    T = length(y); logZ = Ref{Float64}()
    GC.@preserve p smc_check(ccall((:smc_set_summaries, LIBSMC), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Cint),
                                    f.h, component, p, length(p), moments ? 1 : 0))
    GC.@preserve y smc_check(ccall((:smc_log_likelihood, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        f.h, y, T, logZ, C_NULL, C_NULL))
    q = Matrix{Float64}(undef, length(p), T); mean = Matrix{Float64}(undef, f.d, T); var = Matrix{Float64}(undef, f.d, T)
    GC.@preserve q mean var smc_check(ccall((:smc_get_summaries, LIBSMC), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        f.h, T, isempty(p) ? C_NULL : pointer(q), moments ? pointer(mean) : C_NULL, moments ? pointer(var) : C_NULL))
    smc_check(ccall((:smc_set_summaries, LIBSMC), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Cint), f.h, 0, C_NULL, 0, 0))
    return HipParticles(f, 1), HipWeights(f, 1), logZ[], q, mean, var
end

# batched: what the Threads.@threads loops of smc_samplers.jl:112-121,174-180,223-229 become -- ONE call.
# Returns the handle too (its slot m is smc.x[m], smc.w[m]).  skip[m] != 0: filter m is not run, logZ[m] = -Inf (:116).
function log_likelihood(N::Int64, y::Vector{Float64}, models::Vector{<:HipModels}; seed::UInt64=rand(UInt64), f=nothing,
                        streams=nothing, skip=nothing, device::Int=0)
    M = length(models)
    f = f === nothing ? HipFilter(hip_model(models[1])[1], M, N; seed=seed, device=device) : invalidate!(f)
    smc_check(ccall((:smc_reseed, LIBSMC), Cint, (Ptr{Cvoid}, UInt64), f.h, seed))
    set_models!(f, models)
    streams === nothing || set_streams!(f, streams)
    logZ = Vector{Float64}(undef, M)
    if skip !== nothing
        GC.@preserve skip smc_check(ccall((:smc_set_skip, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{UInt8}), f.h, skip))
    end
    try
        GC.@preserve y logZ smc_check(ccall((:smc_log_likelihood, LIBSMC), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            f.h, y, length(y), logZ, C_NULL, C_NULL))
    finally
        skip === nothing || smc_check(ccall((:smc_set_skip, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{UInt8}), f.h, C_NULL))
    end
    return f, logZ
end

# quantile(x, weights(w), p) of examples/inflation_example.jl:45 without moving the cloud: the weights live in the same
# handle, so the method ignores the values of `w` and selects on the device (all filters of the handle; row m is ours)
function StatsBase.quantile(x::HipParticles, ::Union{HipWeights,StatsBase.AbstractWeights}, p::AbstractVector{<:Real}; component=0)
    pp = Float64.(p); out = Matrix{Float64}(undef, length(pp), x.f.M)
    GC.@preserve pp out smc_check(ccall((:smc_get_quantiles, LIBSMC), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}), x.f.h, component, pp, length(pp), out))
    return out[:, x.m]
end

# normalize / resample on whole clouds (particles.jl:5-19), on the device
function normalize(logw::Vector{Float64}, ::Val{:hip})
    w = similar(logw); logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw w smc_check(ccall((:smc_normalize, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint), logw, length(logw), w, logμ, ess, 0))
    return (logμ[], w, ess[])
end
function resample(w::Vector{Float64}, N::Int64, ::Val{:hip}; seed::UInt64=rand(UInt64))
    a = Vector{Int32}(undef, N)
    GC.@preserve w a smc_check(ccall((:smc_resample, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Int64, UInt64, UInt32, UInt32, Ptr{Int32}, Cint), w, length(w), N, seed, 0, 0, a, 0))
    return Int.(a) .+ 1            # the C ABI is 0-based
end

# ---- the OUTER level: one specification for every host (include/smc_hip.h "the OUTER level") ----------------------------
# reweight(logω) of the samplers (undefined in the reference's tree; == normalize): the library's integer normalize over
# segments of 8 entries - the same bits on every host, rank and number of ranks
function reweight(logw::Vector{Float64})
    w = similar(logw); logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw w smc_check(ccall((:smc_host_reweight, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), logw, length(logw), w, logμ, ess))
    return (logμ[], w, ess[])
end
# the bisection of density_tempered for the next exponent (smc_samplers.jl:240-266) -> (ξ, ess, resample_flag, logω)
function temper(logZ::Vector{Float64}, ξ::Float64, ess_min::Float64)
    newξ = Ref{Float64}(); ess = Ref{Float64}(); flag = Ref{Cint}(); logw = similar(logZ)
    GC.@preserve logZ logw smc_check(ccall((:smc_host_outer_temper, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Cint}, Ptr{Float64}),
        logZ, length(logZ), ξ, ess_min, newξ, ess, flag, logw))
    return newξ[], ess[], flag[] != 0, logw
end
# a = resample(ω) of resample!(smc) (smc_samplers.jl:74-84), ascending, 1-based; Philox pick numbers keyed by `seed`
function outer_resample(logw::Vector{Float64}, m::Int, seed::UInt64)
    a = Vector{Int32}(undef, m)
    GC.@preserve logw a smc_check(ccall((:smc_host_outer_resample, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Int64, UInt64, Ptr{Int32}), logw, length(logw), m, seed, a))
    return Int.(a) .+ 1
end
# random_walk_kernel(θ) (smc_samplers.jl:87-101) as the lower Cholesky factor the device wants, row-major [d][d]
function rw_factor(θm::Matrix{Float64})                                      # [dθ x M] column-major == [n][d] row-major
    d = size(θm, 1); L = Matrix{Float64}(undef, d, d); uni = Ref{Cint}()
    GC.@preserve θm L smc_check(ccall((:smc_host_rw_factor, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Cint, Ptr{Float64}, Ptr{Cint}), θm, size(θm, 2), d, L, uni))
    return L, uni[] != 0                                                     # L as C wrote it: row-major [d][d]
end
# the host half of k smc²! steps (smc_samplers.jl:323-338) for the entries this rank holds: records out, then the walk
function outer_window(logw_local::Vector{Float64}, lik::Matrix{Float64})      # lik [n_local x k] column-major == [k][n_local]
    n = length(logw_local); k = size(lik, 2); nseg = cld(n, 8)
    rec = Array{UInt64}(undef, 4, nseg, k)
    GC.@preserve logw_local lik rec smc_check(ccall((:smc_host_outer_window, LIBSMC), Cint,
        (Ptr{Float64}, Ptr{Float64}, Cint, Int64, Ptr{UInt64}), logw_local, lik, k, n, rec))
    return rec
end
function outer_walk(rec::Array{UInt64,3}, n_total::Int, ess_min::Float64)
    k = size(rec, 3); ess = Vector{Float64}(undef, k); j = Ref{Cint}()
    GC.@preserve rec ess smc_check(ccall((:smc_host_outer_walk, LIBSMC), Cint,
        (Ptr{UInt64}, Cint, Int64, Int64, Float64, Ptr{Float64}, Ptr{Cint}), rec, k, size(rec, 2), n_total, ess_min, ess, j))
    return ess[1:j[]], Int(j[])
end
function outer_advance!(logw::Vector{Float64}, logZ::Vector{Float64}, lik::Matrix{Float64}, j::Int)
    GC.@preserve logw logZ lik smc_check(ccall((:smc_host_outer_advance, LIBSMC), Cint,
        (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Int64), logw, logZ, lik, j, length(logw)))
end

# ---- theta sharded over the GPUs of a node ---------------------------------------------------------------------------
function hip_unique_id()                                        # rank 0; ship the bytes to the other ranks (Distributed, a file, ...)
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id smc_check(ccall((:smc_comm_unique_id, LIBSMC), Cint, (Ptr{Cvoid},), id))
    id
end
function all_gather(c::HipComm, v::Vector{Float64})
    out = Vector{Float64}(undef, length(v) * c.world)
    GC.@preserve v out smc_check(ccall((:smc_comm_all_gather, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                                       c.c, v, length(v), out))
    out
end
all_gather(::Nothing, v::Vector{Float64}) = v
# reweight(logZ) with the entries sharded over the ranks (smc_samplers.jl:232,249,265,298,338): smc_host_reweight of the
# concatenated vector, bit for bit; want_w = false and whole-segment slices: only segment records travel
function reweight(c::HipComm, logw_local::Vector{Float64}; want_w::Bool=true)
    n = length(logw_local) * c.world
    allw = want_w ? Vector{Float64}(undef, n) : Float64[]; w = want_w ? Vector{Float64}(undef, n) : Float64[]
    logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw_local allw w smc_check(ccall((:smc_outer_reweight, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.c, logw_local, length(logw_local), want_w ? pointer(allw) : C_NULL, want_w ? pointer(w) : C_NULL, logμ, ess))
    return logμ[], w, ess[], allw
end
# resample!(smc) of the online sampler with sharded filters: a = GLOBAL ancestors (1-based here, the same on every rank)
function exchange_slots!(c::HipComm, f::HipFilter, a::Vector{Int})
    a0 = Int32.(a .- 1)
    GC.@preserve a0 smc_check(ccall((:smc_comm_exchange_slots, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Int64),
                                    c.c, invalidate!(f).h, a0, length(a0)))
end

# ---- building a GPU-backed sampler from the reference's SMC ------------------------------------------------------------
# prior components the device knows (README.md:81-85, examples/inflation_example.jl:33-37,234-239)
hip_prior(p::Uniform) = (Int32(1), Float64[p.a, p.b, 0, 0, 0])
hip_prior(p::Normal) = (Int32(2), Float64[p.μ, p.σ, 0, 0, 0])
hip_prior(p::Truncated{<:Normal}) = (Int32(3), Float64[p.untruncated.μ, p.untruncated.σ, p.lower, p.upper, p.logtp])
hip_prior(p::LogNormal) = (Int32(4), Float64[p.μ, p.σ, 0, 0, 0])
hip_prior(::Any) = nothing
# the components of smc.prior: product_distribution([...]) (README.md:81-85) or one univariate distribution
prior_components(p::Distributions.Product) = p.v
prior_components(p::UnivariateDistribution) = [p]
prior_components(::Any) = nothing
if isdefined(Distributions, :ProductDistribution)      # what product_distribution returns in newer Distributions.jl
    @eval prior_components(p::Distributions.ProductDistribution) = collect(p.dists)
end

# smc.model(θ) as data: raw[k] = θ[raw_from[k]] or a constant.  Found by probing the closure: the parameter row at θ0 and at θ0
# with one component changed - an entry that moves must BE that component (README.md:75-79, examples:227-230 are of this kind)
function infer_theta_map(model, θ0::Vector{Float64}, scalar::Bool=false)
    arg(θ) = scalar ? θ[1] : θ                                 # a univariate prior hands the closure a number
    raw0 = hip_model(model(arg(θ0)))[2]
    raw_from = fill(Int32(-1), length(raw0)); raw_const = copy(raw0)
    for i in eachindex(θ0)
        θp = copy(θ0); θp[i] = 1.25 * θ0[i] + 0.125
        rawp = hip_model(model(arg(θp)))[2]
        for k in eachindex(raw0)
            rawp[k] == raw0[k] && continue
            (rawp[k] == θp[i] && raw0[k] == θ0[i] && raw_from[k] == -1) || return nothing     # not a selection of components
            raw_from[k] = Int32(i - 1); raw_const[k] = 0.0
        end
    end
    return raw_from, raw_const
end

# HipSampler(smc): what rejuvenate! on the device needs, assembled from smc.prior and smc.model; `nothing` fields and
# device_pmmh = false when the prior or the closure is not of the enumerated kind (rejuvenate! then proposes on the host)
function HipSampler(θ::Vector, model, prior; seed::UInt64=rand(UInt64), comm=nothing, device::Int=0)
    M = length(θ); dθ = length(θ[1])
    comps = prior_components(prior)
    specs = comps === nothing ? nothing : hip_prior.(comps)
    tmap = θ[1] isa Number ? infer_theta_map(model, [Float64(θ[1])], true) : infer_theta_map(model, Float64.(collect(θ[1])))
    ok = specs !== nothing && all(!isnothing, specs) && length(specs) == dθ && tmap !== nothing
    fam = ok ? Int32[s[1] for s in specs] : Int32[]
    par = ok ? reduce(hcat, [s[2] for s in specs]) : zeros(5, 0)
    world = comm === nothing ? 1 : comm.world; rank = comm === nothing ? 0 : comm.rank
    M % world == 0 || error("n_theta must be a multiple of the number of ranks")
    per = M ÷ world
    HipSampler(nothing, nothing, ok, fam, par, ok ? tmap[1] : Int32[], ok ? tmap[2] : Float64[], zeros(M), comm,
               rank * per + 1, (rank + 1) * per, UInt64(0), seed, device)
end

# HipSMC(N, M, model, prior, chain, ess_threshold, min_ar=-1.0): the reference's constructor (smc_samplers.jl:29-59), GPU-backed
function HipSMC(N::Int64, M::Int64, model::SSM, prior::Sampleable, chain::Int64, ess_threshold::Float64, min_ar::Float64=-1.0;
                seed::UInt64=rand(UInt64), comm=nothing, device::Int=0) where SSM
    θ = map(m -> rand(prior), 1:M)
    ω = (1 / M) * ones(Float64, M)
    x = [Float64[] for _ in 1:M]; w = [Float64[] for _ in 1:M]         # the clouds live on the device: smc.x / smc.w are views (below)
    hs = HipSampler(θ, model, prior; seed=seed, comm=comm, device=device)
    return SMC{SSM,Float64,eltype(θ),HipKernel}(θ, ω, x, w, 1.0 * M, M * ess_threshold, N, M, chain, zeros(Float64, M),
                                                  model, prior, HipKernel(hs), min_ar, 0.0)
end
# SMC(N, M, model, prior, ...) with a product prior (README.md:81-88, examples/inflation_example.jl:58,256): GPU-backed when the
# closure yields one of the GPU's model families, the reference's own constructor otherwise
function SMC(N::Int64, M::Int64, model::SSM, prior::MultivariateDistribution, chain::Int64, ess_threshold::Float64,
             min_ar::Float64=-1.0) where SSM
    model(rand(prior)) isa HipModels && return HipSMC(N, M, model, prior, chain, ess_threshold, min_ar)
    return invoke(SMC, Tuple{Int64,Int64,SSM,Sampleable,Int64,Float64,Float64}, N, M, model, prior, chain, ess_threshold, min_ar)
end

# smc.x[m], smc.w[m] of a GPU-backed sampler: views of slot m of the online filters (global slot m on the rank that holds it)
function Base.getproperty(smc::HipSMC, s::Symbol)
    if s === :x || s === :w
        hs = hip(smc)
        hs.main === nothing && return getfield(smc, s)
        return s === :x ? [HipParticles(hs.main, m) for m in 1:hs.main.M] : [HipWeights(hs.main, m) for m in 1:hs.main.M]
    end
    return getfield(smc, s)
end

local_models(smc::HipSMC) = (hs = hip(smc); smc.model.(smc.θ[hs.lo:hs.hi]))
local_streams(hs::HipSampler) = UInt32.((hs.lo - 1):(hs.hi - 1))                 # global theta index, 0-based
# full logZ (or any per-particle vector) from this rank's slice
gather(hs::HipSampler, v::Vector{Float64}) = all_gather(hs.comm, v)
# ω of the reference's struct = the normalised outer weights
sync_omega!(smc::HipSMC) = (smc.ω = reweight(hip(smc).logw)[2]; smc)

function expected_parameters(smc::HipSMC)                                    # smc_samplers.jl:61-65 with normalised ω
    ω = reweight(hip(smc).logw)[2]
    return sum(reduce(hcat, smc.θ .* ω), dims=2)
end

# ---- smc_samplers.jl: the sampler entry points, with the reference's signatures ---------------------------------------
# resample!(smc)   smc_samplers.jl:74-84 -- value copies of the filter slots (on one GPU, or between GPUs)
function resample!(smc::HipSMC)
    hs = hip(smc)
    a = outer_resample(hs.logw, smc.M, next_seed!(hs))
    smc.θ = smc.θ[a]; hs.logw = hs.logw[a]; smc.logZ = smc.logZ[a]
    if hs.main !== nothing
        if hs.comm === nothing
            a0 = Int32.(a .- 1)
            GC.@preserve a0 smc_check(ccall((:smc_permute, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Int32}), invalidate!(hs.main).h, a0))
        else
            exchange_slots!(hs.comm, hs.main, a)
        end
    end
    return a
end

function ensure_prop!(smc::HipSMC)
    hs = hip(smc)
    per = hs.hi - hs.lo + 1
    if hs.prop === nothing || hs.prop.N != smc.N
        hs.prop = set_streams!(HipFilter(hip_model(smc.model(smc.θ[1]))[1], per, smc.N; device=hs.device), local_streams(hs))
        if hs.device_pmmh
            GC.@preserve hs smc_check(ccall((:smc_pmmh_configure, LIBSMC), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
                hs.prop.h, length(hs.prior_family), hs.prior_family, hs.prior_par, hs.raw_from, hs.raw_const))
        end
    end
    hs.prop
end

# rejuvenate!(smc, y, ξ, verbose)   smc_samplers.jl:103-146
function rejuvenate!(smc::HipSMC, y::Vector{Float64}, ξ::Float64, verbose::Bool)
    if verbose @printf("\t[rejuvenating]") end
    hs = hip(smc)
    accepted = hs.device_pmmh ? rejuvenate_device!(smc, y, ξ) : rejuvenate_host!(smc, y, ξ)
    hs.logw = zeros(smc.M)                                          # smc.ω[m] = 1.0  (:139)
    smc.ω = ones(smc.M)
    smc.acc_ratio = sum(accepted) / smc.M
    if verbose @printf("\tacc_rate: %1.5f", smc.acc_ratio) end
    return smc
end
rejuvenate!(smc::HipSMC, y::Vector{Float64}, verbose::Bool) = rejuvenate!(smc, y, 1.0, verbose)     # :148

# the whole `for m ... for c in 1:chain` loop (:112-138) in ONE device call per rank; proposals, accept uniforms: Philox,
# keyed by the global theta index
function rejuvenate_device!(smc::HipSMC, y::Vector{Float64}, ξ::Float64)
    hs = hip(smc); dθ = length(smc.θ[1]); per = hs.hi - hs.lo + 1
    L, uni = rw_factor(reduce(hcat, smc.θ))                         # :87-101 from the full cloud every rank holds
    scales = 0.5 * reverse(1:smc.chain)                             # :108
    s = uni ? scales .^ 2 : collect(scales)
    seeds = UInt64[next_seed!(hs) for _ in 1:smc.chain]; move_seed = next_seed!(hs)
    prop = ensure_prop!(smc)
    θm = reduce(hcat, smc.θ[hs.lo:hs.hi]); logZ = smc.logZ[hs.lo:hs.hi]; acc = zeros(UInt8, per); nrun = Ref{Int64}(0)
    GC.@preserve y L s seeds θm logZ acc smc_check(ccall((:smc_pmmh_rejuvenate, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{UInt64}, UInt64,
         Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Ptr{Int64}),
        prop.h, hs.main === nothing ? C_NULL : invalidate!(hs.main).h, y, length(y), ξ, L, s, smc.chain, seeds,
        move_seed, θm, logZ, acc, nrun))
    packed = gather(hs, vcat(vec(θm), logZ, Float64.(acc)))         # ONE all-gather of the moved slices
    blk = reshape(packed, per * (dθ + 2), :)
    smc.θ = eltype(smc.θ) <: Number ? [blk[m, r] for r in 1:size(blk, 2) for m in 1:per] :
                                      [blk[(m - 1) * dθ .+ (1:dθ), r] for r in 1:size(blk, 2) for m in 1:per]
    smc.logZ = vec(blk[per * dθ .+ (1:per), :])
    return vec(blk[per * (dθ + 1) .+ (1:per), :]) .!= 0.0
end

# priors / model closures outside the enumerated families: proposals, insupport, prior ratio and the accept test on the host
# (Julia's own random numbers, as in the reference), the filters of ALL parameter particles of a chain position in one batched
# call - proposals outside the support are skipped on the device (:116) - and the accepted clouds copied on the device (:132-133)
function rejuvenate_host!(smc::HipSMC, y::Vector{Float64}, ξ::Float64)
    hs = hip(smc); M = smc.M
    pmmh_kernel = random_walk_kernel(smc.θ); scales = 0.5 * reverse(1:smc.chain)
    accepted = falses(M)
    for c in 1:smc.chain
        θ_prop = [rand(pmmh_kernel(smc.θ[m], scales[c])) for m in 1:M]
        u = rand(M)
        ok = [insupport(smc.prior, θ_prop[m]) for m in 1:M]
        safe = [ok[m] ? θ_prop[m] : smc.θ[m] for m in hs.lo:hs.hi]
        prop = ensure_prop!(smc)
        _, logZ_loc = log_likelihood(smc.N, y, smc.model.(safe); seed=next_seed!(hs), f=prop, streams=local_streams(hs),
                                     skip=UInt8.(.!ok[hs.lo:hs.hi]))
        logZ_prop = gather(hs, logZ_loc)
        acc = falses(M)
        for m in 1:M
            ok[m] || continue
            prior_ratio = logpdf(smc.prior, θ_prop[m]) - logpdf(smc.prior, smc.θ[m])
            acc[m] = (logZ_prop[m] + logpdf(smc.prior, θ_prop[m]) > -Inf) && log(u[m]) < ξ * (logZ_prop[m] - smc.logZ[m]) + prior_ratio
            if acc[m] smc.logZ[m] = logZ_prop[m]; smc.θ[m] = θ_prop[m] end
        end
        if hs.main !== nothing && any(acc[hs.lo:hs.hi])
            mask = UInt8.(acc[hs.lo:hs.hi])
            GC.@preserve mask smc_check(ccall((:smc_copy_from, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}),
                                              invalidate!(hs.main).h, prop.h, mask))
        end
        accepted .|= acc
    end
    return accepted
end

# exchange!(smc, y, verbose)   smc_samplers.jl:163-189
function exchange!(smc::HipSMC, y::Vector{Float64}, verbose::Bool)
    if smc.acc_ratio < smc.acc_threshold
        if smc.N <= 4096
            smc.N *= 2
            if verbose @printf("\t%d particles added", smc.N) end
            hs = hip(smc)
            f, new_loc = log_likelihood(smc.N, y, local_models(smc); seed=next_seed!(hs), streams=local_streams(hs), device=hs.device)
            new_logZ = gather(hs, new_loc)
            hs.main = f; hs.prop = nothing                          # the superseded N-particle filters are released by their finalizers
            hs.logw = new_logZ .- smc.logZ                          # :183
            _, smc.ω, smc.ess = reweight(hs.logw)
            smc.logZ = new_logZ
        else
            print("\n\t[cannot exceed max state particles]")
        end
    end
end

# density_tempered(smc, y, verbose=true)   smc_samplers.jl:222-281
function density_tempered(smc::HipSMC, y::Vector{Float64}, verbose=true)
    hs = hip(smc)
    _, loc = log_likelihood(smc.N, y, local_models(smc); seed=next_seed!(hs), f=ensure_prop!(smc), streams=local_streams(hs))   # :223-229
    smc.logZ = gather(hs, loc)
    hs.logw = copy(smc.logZ)
    _, smc.ω, smc.ess = reweight(hs.logw)                          # :232
    ξ = 0.0
    while ξ < 1.0
        ξ, smc.ess, resample_flag, hs.logw = temper(smc.logZ, ξ, smc.ess_min)      # :240-266 in one library call
        if verbose @printf("ξ = %1.5f\tess = %4.3f", ξ, smc.ess) end
        if resample_flag
            resample!(smc)                                         # :274
            rejuvenate!(smc, y, ξ, verbose)                        # :277
        end
        if verbose print("\n") end
    end
    sync_omega!(smc)
end

# smc²(smc, y)   smc_samplers.jl:288-301
function smc²(smc::HipSMC, y::Vector{Float64})
    hs = hip(smc); models = local_models(smc)
    hs.main = set_streams!(set_models!(HipFilter(hip_model(models[1])[1], length(models), smc.N; seed=next_seed!(hs), device=hs.device), models),
                           local_streams(hs))
    logμ = Vector{Float64}(undef, length(models))
    GC.@preserve logμ smc_check(ccall((:smc_init, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}), hs.main.h, y[1], logμ))
    smc.logZ = gather(hs, logμ)                                    # :297
    hs.logw = copy(smc.logZ)
    _, smc.ω, smc.ess = reweight(hs.logw)                          # :298
    return smc
end

# the host half of k online steps over this rank's likelihood increments lik [n_local x k]: segment records out, ONE
# all-gather, the walk on every rank, the kept steps added to this rank's slices; -> (ess of the walked steps, j)
function window_walk!(smc::HipSMC, lik::Matrix{Float64}, ess_min::Float64)
    hs = hip(smc); per = hs.hi - hs.lo + 1; k = size(lik, 2); W = hs.comm === nothing ? 1 : hs.comm.world
    (W == 1 || per % 8 == 0) || error("theta slices of whole segments (multiples of 8) are required for sharded online steps")
    lw = hs.logw[hs.lo:hs.hi]; lz = smc.logZ[hs.lo:hs.hi]
    rec = outer_window(lw, lik)                                    # [4 x nseg_local x k]
    if W > 1
        allr = reinterpret(UInt64, all_gather(hs.comm, collect(reinterpret(Float64, vec(rec)))))
        rec = permutedims(reshape(allr, 4, size(rec, 2), k, W), (1, 2, 4, 3))       # [4][nseg_local][W][k]
        rec = reshape(rec, 4, size(rec, 2) * W, k)
    end
    ess, j = outer_walk(collect(rec), smc.M, ess_min)
    outer_advance!(lw, lz, lik, j)
    if W > 1                                                       # every rank keeps the full (small) vectors current
        both = reshape(all_gather(hs.comm, vcat(lw, lz)), 2 * per, W)
        hs.logw = vec(both[1:per, :]); smc.logZ = vec(both[per+1:2per, :])
    else
        hs.logw = lw; smc.logZ = lz
    end
    return ess, j
end

# smc²!(smc, y, t, verbose=true)   smc_samplers.jl:308-340 -- the serial loop :325-335 is ONE batched smc_step
function smc²!(smc::HipSMC, y::Vector{Float64}, t::Int64, verbose::Bool=true)
    if verbose @printf("t = %4d\tess = %4.3f", t - 1, smc.ess) end
    hs = hip(smc)
    if smc.ess < smc.ess_min
        resample!(smc)
        rejuvenate!(smc, y[1:(t-1)], verbose)
        exchange!(smc, y[1:(t-1)], verbose)
    end
    set_models!(invalidate!(hs.main), local_models(smc))
    per = hs.hi - hs.lo + 1
    lik = Matrix{Float64}(undef, per, 1); ess = Vector{Float64}(undef, per)
    GC.@preserve lik ess smc_check(ccall((:smc_step, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}),
                                         hs.main.h, y[t], lik, ess))
    e, _ = window_walk!(smc, lik, 0.0)                             # logω .+= lik; logZ .+= lik; reweight   (:324-338)
    smc.ess = e[1]
    sync_omega!(smc)
    if verbose print("\n") end
end

# `for t in t1:t2 smc²!(smc, y, t) end` (README.md:93-101) with up to `window` propagation steps per device call
# (smc_step_window / smc_step_commit): the same results bit for bit; see smc_samplers.py smc2_run for the tested twin
function smc²_run!(smc::HipSMC, y::Vector{Float64}, t1::Int64, t2::Int64; window::Int=16, verbose::Bool=true)
    hs = hip(smc); t = t1
    while t <= t2
        k = min(window, t2 - t + 1)
        if verbose @printf("t = %4d\tess = %4.3f", t - 1, smc.ess) end
        if smc.ess < smc.ess_min
            resample!(smc)
            rejuvenate!(smc, y[1:(t-1)], verbose)
            exchange!(smc, y[1:(t-1)], verbose)
        end
        set_models!(invalidate!(hs.main), local_models(smc))
        per = hs.hi - hs.lo + 1
        yk = y[t:(t+k-1)]; lik = Matrix{Float64}(undef, per, k); ess = Matrix{Float64}(undef, per, k)
        GC.@preserve yk lik ess smc_check(ccall((:smc_step_window, LIBSMC), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}), hs.main.h, yk, k, lik, ess))
        e, j = window_walk!(smc, lik, smc.ess_min)
        smc.ess = e[end]
        if verbose
            print("\n")
            for i in 1:(j-1) @printf("t = %4d\tess = %4.3f\n", t + i - 1, e[i]) end
        end
        smc_check(ccall((:smc_step_commit, LIBSMC), Cint, (Ptr{Cvoid}, Cint), hs.main.h, j))
        t += j
    end
    sync_omega!(smc)
end
