# julia/reference_restatement.jl -- CPU baseline (ii) of SURVEY 8(d): a <=150-line Julia restatement of the
# reference's hot path (src/particles.jl:5-147 + src/state_space_models.jl:74-109) with `reweight -> normalize`,
# because the package itself cannot be `using`-ed (dangling includes, undefined symbols; SURVEY 0).
# bench.py runs it with `julia -t auto` ONLY if a `julia` with Distributions + StatsBase exists on the box
# (it does not in this image); otherwise the baseline is the C oracle ("port").  Written for this repo.
using Distributions, StatsBase, Random

struct LG; A::Float64; B::Float64; Q::Float64; R::Float64; x0::Float64; s0::Float64; end
initial_dist(m::LG) = Normal(m.x0, sqrt(m.s0))
transition(m::LG, x::Float64) = Normal(m.A * x, sqrt(m.Q))
observation(m::LG, x::Float64) = Normal(m.B * x, sqrt(m.R))

function normalize_w(logw::Vector{Float64})
    maxw = maximum(logw); w = exp.(logw .- maxw); sumw = sum(w)
    return (maxw + log(sumw) - log(length(logw)), w / sumw, 1.0 / sum((w / sumw) .^ 2))
end
resample_w(w::Vector{Float64}, N::Int = length(w)) = sample(1:length(w), Weights(w), N)

function bootstrap_filter_r(N::Int, y::Float64, m::LG)
    x = zeros(N); logw = zeros(N)
    for i in 1:N
        x[i] = rand(initial_dist(m)); logw[i] = logpdf(observation(m, x[i]), y)
    end
    logmu, w, _ = normalize_w(logw)
    return x, w, logmu
end

function bootstrap_filter_r!(x::Vector{Float64}, w::Vector{Float64}, y::Float64, m::LG)
    logw = similar(w); a = resample_w(w); xp = x[a]
    for i in eachindex(x)
        x[i] = rand(transition(m, xp[i])); logw[i] = logpdf(observation(m, x[i]), y)
    end
    return normalize_w(logw)
end

function log_likelihood_r(N::Int, y::Vector{Float64}, m::LG)
    x, w, logZ = bootstrap_filter_r(N, y[1], m)
    for t in 2:length(y)
        logmu, w, _ = bootstrap_filter_r!(x, w, y[t], m); logZ += logmu
    end
    return logZ
end

if abspath(PROGRAM_FILE) == @__FILE__
    N = parse(Int, get(ARGS, 1, "1048576")); T = parse(Int, get(ARGS, 2, "16"))
    m = LG(0.5, 1.0, 0.9, 0.8, 0.0, 1.0); Random.seed!(1998)
    y = randn(T)
    log_likelihood_r(1024, y[1:2], m)                       # compile
    t = @elapsed z = log_likelihood_r(N, y, m)
    println("{\"psteps_per_s\": ", N * T / t, ", \"logZ\": ", z, ", \"threads\": 1}")
end
