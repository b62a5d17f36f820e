# hip_backend.jl -- methods of SequentialMonteCarlo.jl's own generic functions over libsmchip.so (include/smc_hip.h).
#
# Usage: `include("hip_backend.jl")` after `include("smc_samplers.jl")` in src/SequentialMonteCarlo.jl (INTEGRATION.md).
# UNEXECUTED in the build image (no Julia there).  It is kept mechanical - one `ccall` per entry point, every one with a
# literal `(:name, LIBSMC), Ret, (ArgTypes...)` triple - and tests/test_host.py::test_julia_binding_matches_header
# parses each triple and checks name, arity and every argument type against the declarations in include/smc_hip.h.
# The Python binding sequential_monte_carlo_amd/_lib.py + smc_samplers.py is the tested twin of what is written here.
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

# batched: what the Threads.@threads loops of smc_samplers.jl:112-121,174-180,223-229 become -- ONE call.
# Returns the handle too (its slot m is smc.x[m], smc.w[m]).
function log_likelihood(N::Int64, y::Vector{Float64}, models::Vector{<:HipModels}; seed::UInt64=rand(UInt64), f=nothing)
    M = length(models)
    f = f === nothing ? HipFilter(hip_model(models[1])[1], M, N; seed=seed) : invalidate!(f)
    smc_check(ccall((:smc_reseed, LIBSMC), Cint, (Ptr{Cvoid}, UInt64), f.h, seed))
    set_models!(f, models)
    logZ = Vector{Float64}(undef, M)
    GC.@preserve y logZ smc_check(ccall((:smc_log_likelihood, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        f.h, y, length(y), logZ, C_NULL, C_NULL))
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

# normalize / resample on n_theta-vectors (the samplers' `reweight`, smc_samplers.jl:232,...)
function normalize(logw::Vector{Float64}, ::Val{:hip})
    w = similar(logw); logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw w smc_check(ccall((:smc_normalize, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint), logw, length(logw), w, logμ, ess, 0))
    return (logμ[], w, ess[])
end
# reweight(logω) of the samplers (undefined in the reference's tree; == normalize): the library's host routine, the same
# bits on every rank
function reweight(logw::Vector{Float64})
    w = similar(logw); logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw w smc_check(ccall((:smc_host_reweight, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), logw, length(logw), w, logμ, ess))
    return (logμ[], w, ess[])
end
function resample(w::Vector{Float64}, N::Int64, ::Val{:hip}; seed::UInt64=rand(UInt64))
    a = Vector{Int32}(undef, N)
    GC.@preserve w a smc_check(ccall((:smc_resample, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Int64, UInt64, UInt32, UInt32, Ptr{Int32}, Cint), w, length(w), N, seed, 0, 0, a, 0))
    return Int.(a) .+ 1            # the C ABI is 0-based
end

# the index draw of resample!(smc) in ascending order from uniforms of the caller's generator (one linear merge on the host):
# with theta sharded over GPUs the ascending order keeps most filter copies on their rank (hip_exchange! below)
function resample_sorted(w::Vector{Float64}, N::Int64=length(w))
    u = sort!(rand(N)); a = Vector{Int32}(undef, N)
    GC.@preserve w u a smc_check(ccall((:smc_host_resample_sorted, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Int32}), w, length(w), u, N, a))
    return Int.(a) .+ 1
end

# ---- smc_samplers.jl: the SMC container keeps theta / omega / logZ on the host; x and w live in `main` ----------------
# A sampler whose model closure yields one of the GPU families carries two extra fields (or a side table keyed by the
# SMC object): `main::HipFilter` (the online filters smc.x, smc.w) and `prop::HipFilter` (the proposal filters of
# rejuvenate!), both with M filters of N particles, and for the device-side PMMH the description of prior and model map.
mutable struct HipSampler
    main::Union{Nothing,HipFilter}
    prop::Union{Nothing,HipFilter}
    prior_family::Vector{Int32}        # SMC_PRIOR_* per component of smc.prior
    prior_par::Matrix{Float64}         # [SMC_PRIOR_NPAR x d_theta] column-major == [d_theta][SMC_PRIOR_NPAR]
    raw_from::Vector{Int32}            # smc.model(theta): raw[k] = theta[raw_from[k]+1] (0-based in C; -1: constant)
    raw_const::Vector{Float64}
    calls::UInt64                      # evaluation counter -> a fresh Philox seed per batched evaluation
    seed::UInt64
end
next_seed!(s::HipSampler) = (s.calls += 1; (s.seed << 20) + s.calls)

# prior components the device knows (README.md:81-85, examples/inflation_example.jl:33-37,234-239)
hip_prior(p::Uniform) = (Int32(1), Float64[p.a, p.b, 0, 0, 0])
hip_prior(p::Normal) = (Int32(2), Float64[p.μ, p.σ, 0, 0, 0])
hip_prior(p::Truncated{<:Normal}) = (Int32(3), Float64[p.untruncated.μ, p.untruncated.σ, p.lower, p.upper, p.logtp])
hip_prior(p::LogNormal) = (Int32(4), Float64[p.μ, p.σ, 0, 0, 0])

# resample!(smc)   smc_samplers.jl:74-84 -- one GPU: filter slot m <- slot a[m] (value copies)
function resample!(smc::SMC, hs::HipSampler)
    a = resample(smc.ω)
    smc.θ = smc.θ[a]; smc.ω = smc.ω[a]; smc.logZ = smc.logZ[a]
    if hs.main !== nothing
        a0 = Int32.(a .- 1)
        GC.@preserve a0 smc_check(ccall((:smc_permute, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Int32}), invalidate!(hs.main).h, a0))
    end
    return a
end

# rejuvenate!(smc, y, ξ)   smc_samplers.jl:103-146 -- the whole `for m ... for c in 1:chain` loop on the device
function rejuvenate!(smc::SMC, hs::HipSampler, y::Vector{Float64}, ξ::Float64, verbose::Bool)
    if verbose @printf("\t[rejuvenating]") end
    dθ = length(smc.θ[1]); M = smc.M
    θm = reduce(hcat, smc.θ)                                   # [dθ x M] column-major == [n_theta][d_theta]
    Σ = norm(cov(θm')) < 1.e-8 ? Matrix(1.e-2I, dθ, dθ) : (2.83^2 / dθ) * cov(θm') + 1.e-10I     # :95-100
    L = Matrix(cholesky(Symmetric(Σ)).L)
    Lrow = Matrix(L')                                           # row-major [d][d] lower factor for C
    scales = 0.5 * reverse(1:smc.chain)                         # :108
    seeds = UInt64[next_seed!(hs) for _ in 1:smc.chain]; move_seed = next_seed!(hs)
    if hs.prop === nothing
        hs.prop = HipFilter(Cint(hs.main === nothing ? 1 : hs.main.id), M, smc.N)
        GC.@preserve hs smc_check(ccall((:smc_pmmh_configure, LIBSMC), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
            hs.prop.h, dθ, hs.prior_family, hs.prior_par, hs.raw_from, hs.raw_const))
    end
    logZ = copy(smc.logZ); accepted = zeros(UInt8, M); nrun = Ref{Int64}(0)
    GC.@preserve y Lrow scales seeds θm logZ accepted smc_check(ccall((:smc_pmmh_rejuvenate, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{UInt64}, UInt64,
         Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Ptr{Int64}),
        hs.prop.h, hs.main === nothing ? C_NULL : invalidate!(hs.main).h, y, length(y), ξ, Lrow, scales, smc.chain, seeds,
        move_seed, θm, logZ, accepted, nrun))
    smc.θ = [θm[:, m] for m in 1:M]; smc.logZ = logZ; smc.ω = ones(M)
    smc.acc_ratio = sum(accepted) / M
    if verbose @printf("\tacc_rate: %1.5f", smc.acc_ratio) end
    return smc
end

# smc²(smc, y)   smc_samplers.jl:288-301
function smc²(smc::SMC, hs::HipSampler, y::Vector{Float64})
    models = smc.model.(smc.θ)
    hs.main = set_models!(HipFilter(hip_model(models[1])[1], smc.M, smc.N; seed=next_seed!(hs)), models)
    logμ = Vector{Float64}(undef, smc.M)
    GC.@preserve logμ smc_check(ccall((:smc_init, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}), hs.main.h, y[1], logμ))
    smc.logZ = logμ
    _, smc.ω, smc.ess = normalize(copy(logμ))
    return smc
end

# smc²!(smc, y, t)   smc_samplers.jl:308-340 -- the serial loop :325-335 is ONE batched smc_step
function smc²!(smc::SMC, hs::HipSampler, y::Vector{Float64}, t::Int64, verbose::Bool=true)
    if verbose @printf("t = %4d\tess = %4.3f", t - 1, smc.ess) end
    if smc.ess < smc.ess_min
        resample!(smc, hs)
        rejuvenate!(smc, hs, y[1:(t-1)], 1.0, verbose)
    end
    set_models!(invalidate!(hs.main), smc.model.(smc.θ))
    lik = Vector{Float64}(undef, smc.M); ess = Vector{Float64}(undef, smc.M)
    GC.@preserve lik ess smc_check(ccall((:smc_step, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}),
                                         hs.main.h, y[t], lik, ess))
    logω = log.(smc.ω) .+ lik
    smc.logZ .+= lik
    _, smc.ω, smc.ess = normalize(logω)
    if verbose print("\n") end
end

# the same loop with up to k propagation steps per device call (smc_step_window / smc_step_commit): bit-identical to
# `for t in t1:t2 smc²!(smc, hs, y, t) end`; see sequential_monte_carlo_amd/smc_samplers.py smc2_run for the host logic
function smc²_window!(smc::SMC, hs::HipSampler, y::Vector{Float64}, t::Int64, k::Int64)
    set_models!(invalidate!(hs.main), smc.model.(smc.θ))
    yk = y[t:(t+k-1)]; lik = Matrix{Float64}(undef, smc.M, k); ess = Matrix{Float64}(undef, smc.M, k)
    GC.@preserve yk lik ess smc_check(ccall((:smc_step_window, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}), hs.main.h, yk, k, lik, ess))
    j = 0
    while j < k
        j += 1
        logω = log.(smc.ω) .+ lik[:, j]
        smc.logZ .+= lik[:, j]
        _, smc.ω, smc.ess = normalize(logω)
        smc.ess < smc.ess_min && break
    end
    smc_check(ccall((:smc_step_commit, LIBSMC), Cint, (Ptr{Cvoid}, Cint), hs.main.h, j))
    return j                                                   # steps kept; the caller continues at t + j
end

# ---- theta sharded over the GPUs of a node (one Julia process per GPU; SURVEY 8e) -------------------------------------
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
function hip_unique_id()                                        # rank 0; ship the bytes to the other ranks (Distributed, a file, ...)
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id smc_check(ccall((:smc_comm_unique_id, LIBSMC), Cint, (Ptr{Cvoid},), id))
    id
end
# reweight(logZ) with the entries sharded over the ranks (smc_samplers.jl:232,249,265,298,338)
function reweight(c::HipComm, logw_local::Vector{Float64})
    n = length(logw_local) * c.world
    allw = Vector{Float64}(undef, n); w = Vector{Float64}(undef, n); logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw_local allw w smc_check(ccall((:smc_outer_reweight, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.c, logw_local, length(logw_local), allw, w, logμ, ess))
    return logμ[], w, ess[], allw
end
function all_gather(c::HipComm, v::Vector{Float64})
    out = Vector{Float64}(undef, length(v) * c.world)
    GC.@preserve v out smc_check(ccall((:smc_comm_all_gather, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                                       c.c, v, length(v), out))
    out
end
# resample!(smc) of the online sampler with sharded filters: a = GLOBAL ancestors (1-based here, the same on every rank)
function exchange_slots!(c::HipComm, f::HipFilter, a::Vector{Int})
    a0 = Int32.(a .- 1)
    GC.@preserve a0 smc_check(ccall((:smc_comm_exchange_slots, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Int64),
                                    c.c, invalidate!(f).h, a0, length(a0)))
end
