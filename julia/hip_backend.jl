# hip_backend.jl -- methods of SequentialMonteCarlo.jl's own generic functions over libsmchip.so
# (include/smc_hip.h).  UNEXECUTED in the build image (no Julia there); mechanical: one ccall per entry
# point, mirroring the tested Python binding sequential_monte_carlo_amd/_lib.py + particles.py.
# Usage: add `include("hip_backend.jl")` after `include("particles.jl")` in src/SequentialMonteCarlo.jl
# (INTEGRATION.md explains the boundary).
# src/hip_backend.jl  -- methods of the reference's own generic functions over libsmchip.so
const LIBSMC = "libsmchip"            # sequential_monte_carlo_amd/lib/libsmchip.so on LD_LIBRARY_PATH

smc_check(rc) = rc == 0 || error(unsafe_string(ccall((:smc_last_error, LIBSMC), Cstring, ())))

# model families the GPU implements (include/smc_hip.h: SMC_MODEL_*), and their parameter rows
hip_model(m::LinearModel{Float64,Float64,Float64,Float64,Float64,Float64}) =
    (Cint(1), Float64[m.A, m.B, m.Q, m.R, m.x0, m.σ0])            # state_space_models.jl:46-58
hip_model(m::UCSV) =
    (Cint(3), Float64[m.γ[1], m.γ[2], m.x0, m.log_σ0[1], m.log_σ0[2]])   # :215-222

mutable struct HipFilter                # device-resident (x, w) of one bootstrap_filter call
    h::Ptr{Cvoid}; N::Int; d::Int
    function HipFilter(id, raw, N; seed=rand(UInt64), device=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        smc_check(ccall((:smc_create, LIBSMC), Cint,
            (Cint, Int64, Int64, Cint, UInt64, Cint, UInt32, Ref{Ptr{Cvoid}}),
            id, 1, N, 0, seed, device, 0, out))
        f = new(out[], N, id == 3 ? 3 : 1)
        finalizer(x -> ccall((:smc_destroy, LIBSMC), Cint, (Ptr{Cvoid},), x.h), f)
        GC.@preserve raw smc_check(ccall((:smc_set_params, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Float64}), f.h, raw))
        f
    end
end

const HipModels = Union{LinearModel{Float64,Float64,Float64,Float64,Float64,Float64},UCSV}

# bootstrap_filter(N, y, model) -> (x, w, logμ)                      particles.jl:87-105
function bootstrap_filter(N::Int64, y::Float64, model::HipModels)
    f = HipFilter(hip_model(model)..., N)
    logμ = Ref{Float64}()
    smc_check(ccall((:smc_init, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ref{Float64}), f.h, y, logμ))
    return HipParticles(f), HipWeights(f), logμ[]
end

# bootstrap_filter!(x, w, y, model) -> (logμ, w, ess)                particles.jl:107-129
function bootstrap_filter!(states::HipParticles, weights::HipWeights, y::Float64, model::HipModels)
    logμ = Ref{Float64}(); ess = Ref{Float64}()
    smc_check(ccall((:smc_step, LIBSMC), Cint, (Ptr{Cvoid}, Float64, Ref{Float64}, Ref{Float64}),
                    states.f.h, y, logμ, ess))
    return logμ[], HipWeights(states.f), ess[]
end

# log_likelihood(N, y, model) -> (x, w, logZ)                         particles.jl:132-147
function log_likelihood(N::Int64, y::Vector{Float64}, model::HipModels)
    f = HipFilter(hip_model(model)..., N)
    logZ = Ref{Float64}()
    GC.@preserve y smc_check(ccall((:smc_log_likelihood, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ref{Float64}, Ptr{Float64}, Ptr{Float64}),
        f.h, y, length(y), logZ, C_NULL, C_NULL))
    return HipParticles(f), HipWeights(f), logZ[]
end

# batched: what the Threads.@threads loops of smc_samplers.jl:112-121,223-229 become -- ONE call
function log_likelihood(N::Int64, y::Vector{Float64}, models::Vector{<:HipModels}; seed=rand(UInt64))
    id = hip_model(models[1])[1]
    raw = reduce(hcat, last.(hip_model.(models)))      # n_raw × M column-major == [M][n_raw] row-major
    out = Ref{Ptr{Cvoid}}(C_NULL); M = length(models)
    smc_check(ccall((:smc_create, LIBSMC), Cint, (Cint, Int64, Int64, Cint, UInt64, Cint, UInt32, Ref{Ptr{Cvoid}}),
                    id, M, N, 0, seed, 0, 0, out))
    logZ = Vector{Float64}(undef, M)
    GC.@preserve raw y logZ begin
        smc_check(ccall((:smc_set_params, LIBSMC), Cint, (Ptr{Cvoid}, Ptr{Float64}), out[], raw))
        smc_check(ccall((:smc_log_likelihood, LIBSMC), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            out[], y, length(y), logZ, C_NULL, C_NULL))
    end
    ccall((:smc_destroy, LIBSMC), Cint, (Ptr{Cvoid},), out[])
    return logZ
end

# x materialises on demand: README.md:41,51 `quantile(x, ...)` keeps working through AbstractVector
struct HipParticles <: AbstractVector{Float64}; f::HipFilter; end
struct HipWeights   <: AbstractVector{Float64}; f::HipFilter; end
Base.size(p::Union{HipParticles,HipWeights}) = (p.f.N,)
function Base.collect(p::HipParticles)
    x = Matrix{Float64}(undef, p.f.N, p.f.d)          # [d][1][N] row-major == N × d column-major
    GC.@preserve x smc_check(ccall((:smc_get_state, LIBSMC), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}), p.f.h, x, C_NULL, C_NULL))
    p.f.d == 1 ? vec(x) : x
end
Base.getindex(p::HipParticles, i::Int) = collect(p)[i]

# quantile(x, weights(w), p) of examples/inflation_example.jl:45 without moving the cloud: the weights live
# in the same handle, so the method ignores the values of `w` and selects on the device
function StatsBase.quantile(x::HipParticles, ::Union{HipWeights,StatsBase.AbstractWeights}, p::AbstractVector{<:Real}; component=0)
    out = Vector{Float64}(undef, length(p)); pp = Float64.(p)
    GC.@preserve pp out smc_check(ccall((:smc_get_quantiles, LIBSMC), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}), x.f.h, component, pp, length(pp), out))
    return out
end

# normalize / resample on n_theta-vectors (the samplers' `reweight`, smc_samplers.jl:232,...)
function normalize(logw::Vector{Float64}, ::Val{:hip})
    w = similar(logw); logμ = Ref{Float64}(); ess = Ref{Float64}()
    GC.@preserve logw w smc_check(ccall((:smc_normalize, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Ptr{Float64}, Ref{Float64}, Ref{Float64}, Cint), logw, length(logw), w, logμ, ess, 0))
    return (logμ[], w, ess[])
end
function resample(w::Vector{Float64}, N::Int64, ::Val{:hip}; seed=rand(UInt64))
    a = Vector{Int32}(undef, N)
    GC.@preserve w a smc_check(ccall((:smc_resample, LIBSMC), Cint,
        (Ptr{Float64}, Int64, Int64, UInt64, UInt32, UInt32, Ptr{Int32}, Cint), w, length(w), N, seed, 0, 0, a, 0))
    return Int.(a) .+ 1            # the C ABI is 0-based
end
