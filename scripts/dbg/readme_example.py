import sys; sys.path.insert(0, "/root/repo")
import time
import sequential_monte_carlo_amd as smc
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 1000)
smc.log_likelihood(2**20, y, m)
t0 = time.perf_counter(); x, w, logZ = smc.log_likelihood(2**20, y, m); print("log_likelihood %.1f ms logZ %.3f" % ((time.perf_counter() - t0) * 1e3, logZ))
x, w, logmu = smc.bootstrap_filter(1024, y[0], m)
for t in range(1, 50):
    logmu, w, ess = smc.bootstrap_filter_(x, w, y[t], m)
    q25, q50, q75 = x.quantile([0.25, 0.5, 0.75])
    mean, var = x.moments()
print("filter loop ok", q25, q50, q75, mean, var)
# the same loop, 999 observations: step by step with the summaries fetched per observation, against ONE call that records them on the device
x, w, logmu = smc.bootstrap_filter(1024, y[0], m, seed=5)
t0 = time.perf_counter()
for t in range(1, 1000):
    logmu, w, ess = smc.bootstrap_filter_(x, w, y[t], m)
    q = x.quantile([0.25, 0.5, 0.75]); mv = x.moments()
loop_ms = (time.perf_counter() - t0) * 1e3
for rep in range(4):   # a lone workgroup does not wake the clocks: the first calls after idling run 2-3x slower (scripts/dbg/small_filter.py)
    t0 = time.perf_counter(); x1, w1, logZ1, s1 = smc.log_likelihood(1024, y, m, seed=5, quantiles=[0.25, 0.5, 0.75], moments=True); one_ms = (time.perf_counter() - t0) * 1e3
for rep in range(4):
    t0 = time.perf_counter(); smc.log_likelihood(1024, y, m, seed=5); plain_ms = (time.perf_counter() - t0) * 1e3
print("README loop, Nx = 1024, 999 steps with 3 quantiles + mean/var per step: step-by-step %.1f ms (%.1f us per observation); ONE call %.2f ms (%.2f us per observation); "
      "the same call without summaries %.2f ms; last quantiles equal: %s" % (loop_ms, loop_ms, one_ms, one_ms, plain_ms, (s1["quantiles"][-1] == q).all()))
prior = smc.product_distribution([smc.TruncatedNormal(0, 1, -1, 1), smc.LogNormal(), smc.LogNormal()])
mod = lambda th: smc.UnivariateLinearGaussian(A=th[0], B=1.0, Q=th[1], R=th[2])
tmap = smc.ThetaMap(smc.LinearModel.model_id, [0, -1, 1, 2, -1, -1], [0, 1, 0, 0, 0, 1])
y = y[:200]
for rep in range(3):
    s = smc.SMC(1024, 512, mod, prior, 3, 0.5, theta_map=tmap)
    t0 = time.perf_counter(); smc.density_tempered(s, y, verbose=False); dt = time.perf_counter() - t0
    s2 = smc.SMC(1024, 512, mod, prior, 3, 0.5, theta_map=tmap)
    t0 = time.perf_counter(); smc.smc2(s2, y); smc.smc2_run(s2, y, 2, len(y), verbose=False); d2 = time.perf_counter() - t0
print("density_tempered %.1f ms, smc2 %.1f ms; posterior mean" % (dt * 1e3, d2 * 1e3), (s.theta * s.omega[:, None]).sum(axis=0))
