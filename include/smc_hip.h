/*
 * smc_hip.h -- C ABI of libsmchip.so: the MI355X (gfx950) particle-filter hot path.
 *
 * Drop-in boundary for charlesknipp/sequential_monte_carlo (reference @ v1).  The reference has
 * no FFI: its filters call Julia model methods per particle (src/particles.jl:97-98,123-124).
 * A GPU cannot call Julia closures, so the boundary is "enumerated model family + parameter
 * rows"; a Julia wrapper adds methods of the SAME generic functions for these model types and
 * `ccall`s the entry points below (INTEGRATION.md shows the stub).  Each entry point names the
 * reference function it replaces.
 *
 * Conventions: plain pointers and sizes only.  Host arrays are caller-owned and borrowed for
 * the duration of the call; device state is owned by the opaque handle.  Every function
 * returns 0 on success, a negative SMC_E* code otherwise; smc_last_error() returns a
 * thread-local message.  Never aborts.  Indices are 0-based int32 (the Julia wrapper adds 1).
 * Layouts: params [n_theta][n_raw] row-major; x [d][n_theta][n_x]; w, anc [n_theta][n_x].
 * A handle is used from one host thread at a time; different handles are independent.
 */
#ifndef SMC_HIP_H
#define SMC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* model families (src/state_space_models.jl) */
#define SMC_MODEL_LG1D 1   /* UnivariateLinearGaussian :74-109  raw = (A,B,Q,R,x0,sigma0), Q R sigma0 variances */
#define SMC_MODEL_SV1D 2   /* stochastic volatility (SURVEY A7') raw = (mu,rho,sigma)                            */
#define SMC_MODEL_UCSV3D 3 /* UCSV :215-263                     raw = (gamma_eps,gamma_eta,x0,log_s_eps0,log_s_eta0) */

#define SMC_OK 0
#define SMC_EINVAL (-1)
#define SMC_EHIP (-2)
#define SMC_ESTATE (-3)
#define SMC_ENOMEM (-4)

/* flags of smc_create */
#define SMC_FLAG_ANCESTORS 1u /* keep the ancestor vector `a` of the last step (particles.jl:117)      */
#define SMC_FLAG_NO_RESIDENT 2u /* never use the LDS-resident whole-series kernel (testing)            */
#define SMC_FLAG_SYSTEMATIC 4u /* OPT-IN: systematic resampling instead of the reference's multinomial
                                * resample (particles.jl:17-19 draws iid): one uniform per step, child j takes
                                * the point (j + u)/N of the weight CDF.  Same expectation N w_i of every
                                * particle's children, lower variance, a different law - never the default. */

typedef struct smc_filter_s* smc_handle;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* n_theta independent bootstrap filters of n_x particles each (the batched callers
 * src/smc_samplers.jl:112-121,223-229,289-295,325-335 become ONE handle with n_theta > 1).
 * seg: particles per segment (power of two in [256,8192]); 0 = automatic (smc_auto_seg). It is part of the
 * random-number contract like the seed.  A filter has at most 16384 segments (n_x <= 2^27 with seg = 8192; the automatic
 * choice keeps segments of 2048 up to 2^25 particles).  Filter m uses Philox stream id m until smc_set_streams says otherwise. */
int smc_create(int model_id, int64_t n_theta, int64_t n_x, int seg, uint64_t seed, int device, uint32_t flags,
               smc_handle* out);
/* waits for the handle's stream, then releases it.  The device slab, the stream, the events and the pinned mirrors of up to 8
 * destroyed handles (512 MB of device memory at most) are kept for the next smc_create that fits them: creating and destroying
 * a handle per call costs 0.02 ms instead of 0.6 ms. */
int smc_destroy(smc_handle h);

/* smc.model(theta[m]) for every m (src/smc_samplers.jl:120,178,227,293,330): raw parameter rows. */
int smc_set_params(smc_handle h, const double* raw /*[n_theta][n_raw]*/);
/* Philox stream id per filter (e.g. the GLOBAL theta index when theta is sharded over GPUs). */
int smc_set_streams(smc_handle h, const uint32_t* stream /*[n_theta]*/);
int smc_reseed(smc_handle h, uint64_t seed);

/* ---- the hot path ----------------------------------------------------------------------------*/
/* bootstrap_filter(N, y, model) -> (x, w, logmu)            src/particles.jl:87-105 */
int smc_init(smc_handle h, double y1, double* logmu /*[n_theta]*/);
/* bootstrap_filter!(x, w, y, model) -> (logmu, w, ess)      src/particles.jl:107-129 */
int smc_step(smc_handle h, double y_t, double* logmu /*[n_theta]*/, double* ess /*[n_theta] or NULL*/);
/* k consecutive bootstrap_filter! calls (the loop `for t in 2:T smc²!(smc,y,t)` of the online sampler,
 * src/smc_samplers.jl:325-335, between two resample-move decisions) in ONE launch with the clouds resident in LDS:
 * returns (logmu, ess) of every step, [k][n_theta] each, but does NOT advance the filters - the caller looks at the
 * k outer ESS values and then keeps the first j steps with smc_step_commit(h, j) (j < k re-runs those j steps: the
 * random numbers are counter based, the bits are the same; j = 0 keeps nothing).  Bit-identical to k (or j) smc_step
 * calls.  Needs single-segment filters that fit the LDS-resident kernel (n_x <= 8192); k <= 64. */
int smc_step_window(smc_handle h, const double* y /*[k]*/, int k, double* logmu /*[k][n_theta]*/, double* ess /*[k][n_theta] or NULL*/);
int smc_step_commit(smc_handle h, int j);
/* log_likelihood(N, y, model) -> (x, w, logZ)               src/particles.jl:132-147
 * logmu_trace / ess_trace: [T][n_theta] or NULL. */
int smc_log_likelihood(smc_handle h, const double* y, int64_t T, double* logZ /*[n_theta]*/,
                       double* logmu_trace, double* ess_trace);
/* Per-step filtered summaries INSIDE the multi-step calls: what README.md:33-61 and examples/inflation_example.jl:39-55 compute
 * on the host after every bootstrap_filter! - quantile(x, weights(w), p), the weighted mean and variance - recorded at every
 * step of the following smc_log_likelihood / smc_step_window calls of the handle, on the device, without a host round trip per
 * observation (the README loop becomes ONE call).  component: state coordinate of the quantiles; p [np], np <= 8 (0: no
 * quantiles); moments != 0: mean and variance of every coordinate.  np = 0 and moments = 0 switch it off again.
 * Quantile definition: that of smc_get_quantiles (inverse of the weighted empirical CDF in the filter's integer weights).
 * smc_get_summaries hands over the first T steps of the last such call: q [T][n_theta][np], mean / var [T][d][n_theta]
 * (NULL: not wanted).  Single-segment filters compute them inside the LDS-resident kernel; larger ones by trailing kernels
 * on the handle's stream after every step. */
int smc_set_summaries(smc_handle h, int component, const double* p /*[np]*/, int np, int moments);
int smc_get_summaries(smc_handle h, int64_t T, double* q /*[T][n_theta][np]*/, double* mean /*[T][d][n_theta]*/, double* var /*[T][d][n_theta]*/);
/* the (x, w) the reference returns / mutates; any pointer may be NULL. w is the normalised
 * weight vector of normalize() (particles.jl:11); anc needs SMC_FLAG_ANCESTORS. */
int smc_get_state(smc_handle h, double* x /*[d][n_theta][n_x]*/, double* w /*[n_theta][n_x]*/,
                  int32_t* anc /*[n_theta][n_x]*/);
/* accumulated log-likelihood so far and ESS of the current weights */
int smc_get_logZ(smc_handle h, double* logZ /*[n_theta]*/, double* ess /*[n_theta] or NULL*/);
/* resample!(smc) of the OUTER sampler (src/smc_samplers.jl:74-84): filter slot m <- slot a[m]
 * (value copy of x cloud, weights and logZ; stream ids stay with the slot).  The indices are copied before the call returns;
 * the device work is enqueued on the handle's stream and NOT waited for (every later call on the handle is ordered behind it). */
int smc_permute(smc_handle h, const int32_t* a /*[n_theta]*/);
/* PMMH accept step of rejuvenate! (src/smc_samplers.jl:129-136): for every m with mask[m] != 0 the
 * filter state of slot m (x cloud, weights, logZ) is overwritten by slot m of `src` (the proposal
 * filters, same geometry).  Value copy on the device; streams and parameters are not copied. */
int smc_copy_from(smc_handle dst, smc_handle src, const uint8_t* mask /*[n_theta]*/);

/* Proposals outside the prior's support: the reference never filters them (src/smc_samplers.jl:116).  Filters m with
 * skip[m] != 0 are left out by the following smc_log_likelihood calls (their logZ reads -inf, their state is not
 * touched); NULL runs every filter again. */
int smc_set_skip(smc_handle h, const uint8_t* skip /*[n_theta] or NULL*/);

/* ---- rejuvenate!(smc, y, xi) on the device: src/smc_samplers.jl:103-146 -------------------------
 * The handle `prop` holds the proposal filters of this rank's parameter particles (same geometry as the online
 * filters `main`, if any).  smc_pmmh_configure describes what a GPU cannot call as closures:
 *   prior  = product_distribution of d_theta enumerated components (smc.prior; README.md:81-85,
 *            examples/inflation_example.jl:33-37,234-239), par rows of SMC_PRIOR_NPAR doubles;
 *   model  = smc.model(theta): raw parameter row k is theta[raw_from[k]] (raw_from[k] >= 0) or raw_const[k].
 * smc_pmmh_rejuvenate then runs the whole `for c in 1:chain` loop (:113-137) for every parameter particle without
 * a host round trip: theta' ~ MvNormal(theta, scales[c] * L L') (:114; L = lower Cholesky factor of the random-walk
 * covariance :95-100, row-major [d][d]), insupport (:116), log_likelihood(N, y, model(theta')) with Philox seed
 * filter_seeds[c] for the in-support proposals only (:117-121), the accept test log(rand()) < xi (logZ' - logZ) +
 * logprior(theta') - logprior(theta) (:123-129), and theta / logZ / x / w of the accepted particles (:130-133;
 * x, w are copied into `main` when it is not NULL).  Proposal normals and accept uniforms are Philox draws keyed by
 * (move_seed, stream id of the filter = global theta index, chain position): independent of the sharding.
 * theta [n_theta][d_theta] and logZ [n_theta] are read and updated in place; accepted[m] = 1 if particle m moved at
 * least once (acc_array :135); *filters_run = number of proposal filters actually executed. */
#define SMC_PRIOR_UNIFORM 1     /* par = (lo, hi)                                                     */
#define SMC_PRIOR_NORMAL 2      /* par = (mu, sigma)                                                  */
#define SMC_PRIOR_TRUNCNORMAL 3 /* par = (mu, sigma, lo, hi, log(Phi((hi-mu)/sigma) - Phi((lo-mu)/sigma))) */
#define SMC_PRIOR_LOGNORMAL 4   /* par = (mu, sigma) of log x                                          */
#define SMC_PRIOR_NPAR 5
#define SMC_MAX_DTHETA 8
int smc_pmmh_configure(smc_handle prop, int d_theta, const int32_t* prior_family /*[d_theta]*/,
                       const double* prior_par /*[d_theta][SMC_PRIOR_NPAR]*/, const int32_t* raw_from /*[n_raw]*/,
                       const double* raw_const /*[n_raw]*/);
int smc_pmmh_rejuvenate(smc_handle prop, smc_handle main /*or NULL*/, const double* y, int64_t T, double xi,
                        const double* chol /*[d_theta][d_theta]*/, const double* scales /*[chain]*/, int chain,
                        const uint64_t* filter_seeds /*[chain]*/, uint64_t move_seed, double* theta /*[n_theta][d_theta]*/,
                        double* logZ /*[n_theta]*/, uint8_t* accepted /*[n_theta] or NULL*/, int64_t* filters_run /*or NULL*/);
/* ---- the OUTER level of the samplers: one integer, order-free, shardable specification (host code, no GPU needed) --------
 * reweight (undefined in the reference's tree; == normalize, src/particles.jl:5-15) is the inner filter's normalize with
 * segments of SMC_OUTER_SEG consecutive entries: per segment the record (kb, S, S2hi, S2lo) = (largest binary exponent, sum
 * and sum of squares of the 48-bit fixed-point weights relative to it), combined by shifts against the largest kb.  Integer
 * sums only: the same bits for any order of evaluation and for any dealing of whole segments to ranks (a rank computes the
 * records of the segments it holds, the ranks exchange records, every rank combines them).  The online sampler carries the
 * UN-NORMALISED outer log-weights logw: smc²!'s `log.(omega) .+ lik` (:324) of re-normalised weights (:338) is the same
 * weight vector up to a common factor, which reweight removes. */
#define SMC_OUTER_SEG 8
int smc_outer_seg(void);
/* reweight(logw) -> (logmu, w, ess)   src/smc_samplers.jl:232,249,265,298,338; w [n] or NULL */
int smc_host_reweight(const double* logw, int64_t n, double* w, double* logmu, double* ess);
/* the records of the whole segments a rank holds (n_local entries beginning at a multiple of SMC_OUTER_SEG; the last segment
 * of the whole vector may be short): rec [ceil(n_local / SMC_OUTER_SEG)][4] 8-byte words (bits of kb, S, S2hi, S2lo) */
int smc_host_outer_records(const double* logw_local, int64_t n_local, uint64_t* rec);
/* (logmu, ess) of a vector of n_total entries from the records of ALL its segments, in segment order */
int smc_host_outer_combine(const uint64_t* rec, int64_t nseg, int64_t n_total, double* logmu, double* ess);
/* the host half of up to k consecutive smc²! steps (src/smc_samplers.jl:323-338) over the log-likelihood increments
 * lik [k][n_local] of a window of inner-filter steps, in three pieces so that only records cross the ranks:
 *   smc_host_outer_window   records of logw + lik_1, logw + lik_1 + lik_2, ... (nothing modified): rec [k][nseg_local][4]
 *   smc_host_outer_walk     ess of every step from the records of ALL segments, rec [k][nseg][4]; stops after the first step with
 *                           ess < ess_min: *j_out = steps walked, ess_out [k]
 *   smc_host_outer_advance  keep the first j steps: logw .+= lik_t, logZ .+= lik_t, t = 1..j in step order (:333-334) */
int smc_host_outer_window(const double* logw_local, const double* lik /*[k][n_local]*/, int k, int64_t n_local, uint64_t* rec);
int smc_host_outer_walk(const uint64_t* rec, int k, int64_t nseg, int64_t n_total, double ess_min, double* ess_out /*[k]*/, int* j_out);
int smc_host_outer_advance(double* logw, double* logZ, const double* lik /*[k][n]*/, int j, int64_t n);
/* the bisection for the next tempering exponent of density_tempered (src/smc_samplers.jl:240-266) in one call: *xi_new, the
 * ESS of reweight((xi_new - xi) .* logZ), *resample_flag = 0 at the corner solution xi_new = 1 (:261-266); logw_out [n] or
 * NULL = (xi_new - xi) .* logZ */
int smc_host_outer_temper(const double* logZ, int64_t n, double xi, double ess_min, double* xi_new, double* ess, int* resample_flag,
                          double* logw_out);
/* the index draw of resample!(smc) (src/smc_samplers.jl:74-84: sample(1:n, Weights(w), m)) for the weights exp(logw): m iid
 * draws through the inverse of the integer weight CDF, pick numbers = 64-bit Philox draws keyed by `seed`; ancestors in ASCENDING
 * order, 0-based (the order of a resampled population carries no information; ascending keeps most filter copies of a sharded
 * online sampler on their rank).  All weights zero: the identity. */
int smc_host_outer_resample(const double* logw, int64_t n, int64_t m, uint64_t seed, int32_t* a /*[m]*/);
/* random_walk_kernel(theta) (src/smc_samplers.jl:87-101): lower Cholesky factor L [d][d] (row-major) of the PMMH proposal
 * covariance 2.83^2/d cov(theta) + 1e-10 I (1e-2 I when norm(cov) < 1e-8) from the cloud theta [n][d], fixed order of
 * operations; d = 1: L = [[2.83^2 var + 1e-10]] handed to Normal() as a standard deviation (:87-92), *univariate = 1 */
int smc_host_rw_factor(const double* theta, int64_t n, int d, double* L /*[d][d]*/, int* univariate);
/* the spec's PMMH pieces on the host (parity tests): proposal, log prior (NaN-free; -inf outside the support) */
int smc_host_pmmh_propose(int d_theta, uint64_t move_seed, uint32_t stream, uint32_t c, const double* theta,
                          const double* chol, double scale, double* prop);
double smc_host_pmmh_log_uniform(uint64_t move_seed, uint32_t stream, uint32_t c);
double smc_host_prior_logpdf(int family, const double* par /*[SMC_PRIOR_NPAR]*/, double x);

/* Moving whole filters between handles / GPUs (outer resample! of the online sampler when theta is
 * sharded, src/smc_samplers.jl:74-84 + SURVEY 8e/8f.2): pack k slots (x cloud, weights, segment records,
 * logZ) into / out of a caller-provided DEVICE buffer of k * smc_slot_bytes() bytes (e.g. a torch tensor
 * that is then exchanged with an RCCL all-to-all).  idx: local slot indices, host array. */
int smc_slot_bytes(smc_handle h, int64_t* bytes);
int smc_pack_slots(smc_handle h, const int32_t* idx, int64_t k, void* device_buf);
int smc_unpack_slots(smc_handle h, const int32_t* idx, int64_t k, const void* device_buf);

/* ---- theta sharded over the GPUs of one node, for hosts without torch.distributed ---------------------
 * One process per GPU.  Rank 0 calls smc_comm_unique_id and hands the SMC_COMM_ID_BYTES bytes to the other ranks by
 * whatever means the host has (a file, a socket, Julia's Distributed); every rank then calls smc_comm_create.  RCCL
 * (xGMI) underneath, opened with dlopen at the first call.  The collectives are the ones the samplers have:
 *   smc_outer_reweight       reweight(logZ) / reweight(logw) of src/smc_samplers.jl:232,249,265,298,338 with the
 *                            entries sharded over the ranks: the SAME function as smc_host_reweight on the concatenated
 *                            vector, bit for bit, for any number of ranks.  w_all and logw_all NULL and n_local a multiple of
 *                            SMC_OUTER_SEG: the ranks exchange segment records only; otherwise one all-gather of the slices
 *   smc_comm_all_gather      n doubles per rank -> [world][n] on every rank (theta / logZ / accepted after rejuvenate!)
 *   smc_comm_exchange_slots  resample!(smc) of the online sampler (src/smc_samplers.jl:74-84) when the filters of `h`
 *                            are sharded: a[m] is the GLOBAL ancestor of GLOBAL slot m (same vector on every rank, rank r
 *                            holds slots [r M/world, (r+1) M/world)); whole filters travel device to device */
#define SMC_COMM_ID_BYTES 128
typedef struct smc_comm_s* smc_comm;
int smc_comm_unique_id(void* id /*[SMC_COMM_ID_BYTES] out*/);
int smc_comm_create(const void* id, int rank, int world, int device, smc_comm* out);
int smc_comm_destroy(smc_comm c);
int smc_comm_rank(smc_comm c, int* rank, int* world);
int smc_comm_all_gather(smc_comm c, const double* local /*[n]*/, int64_t n, double* all /*[world][n]*/);
int smc_outer_reweight(smc_comm c, const double* logw_local /*[n_local]*/, int64_t n_local, double* logw_all /*or NULL*/,
                       double* w_all /*[n_local*world] or NULL*/, double* logmu, double* ess);
int smc_comm_exchange_slots(smc_comm c, smc_handle h, const int32_t* a /*[M]*/, int64_t M);
/* the plan smc_comm_exchange_slots follows, as pure host arithmetic (no GPU; tested on CPU against the Python twin):
 * send_idx [<= M] local slots to pack, grouped by destination rank (send_cnt [world], *n_send in total); dest_idx [M/world]
 * local slots to unpack into, grouped by source rank (recv_cnt [world]) */
int smc_comm_plan_exchange(const int32_t* a /*[M]*/, int64_t M, int rank, int world, int32_t* send_idx, int64_t* send_cnt /*[world]*/,
                           int64_t* n_send, int32_t* dest_idx, int64_t* recv_cnt /*[world]*/);

/* raw fixed-point weight state (tests): C [n_theta][nseg*seg], m/S/S2hi/S2lo [n_theta][nseg] */
int smc_get_weights_raw(smc_handle h, uint64_t* C, double* m, uint64_t* S, uint64_t* S2hi, uint64_t* S2lo);
int smc_get_geometry(smc_handle h, int* seg, int* nseg, int* d, int* resident);
/* device time (HIP events on the handle's stream) of the last log_likelihood / step_window call; after smc_init / smc_step (which
 * place no events on the stream: their results arrive through a ticket in pinned host memory) the host time spent waiting */
int smc_last_elapsed_ms(smc_handle h, double* ms);
int smc_synchronize(smc_handle h);
/* roofline measurement: runs log_likelihood through the one-launch-per-step path and brackets
 * `nsample` evenly spaced runs of 32 consecutive k_step launches (8 when T < 513, one when T < 65) with HIP events
 * on the handle's stream; returns the average / minimum bracketed duration PER k_step launch in ms. */
int smc_time_step_kernel(smc_handle h, const double* y, int64_t T, int nsample, double* avg_ms, double* min_ms);
/* what an EMPTY HIP-event bracket measures on the handle's stream (average of nsample brackets with a
 * trivial kernel before them): the fixed cost contained in every bracketed figure above. */
int smc_event_overhead_ms(smc_handle h, int nsample, double* avg_ms);

/* ---- stand-alone A1 / A2 ---------------------------------------------------------------------*/
/* normalize(logw) -> (logmu, w, ess)                        src/particles.jl:5-15
 * (also the samplers' `reweight`, src/smc_samplers.jl:232,249,265,298,338) */
int smc_normalize(const double* logw, int64_t n, double* w, double* logmu, double* ess, int device);
/* resample(w, N) -> N iid Categorical(w) indices, unsorted  src/particles.jl:17-19 */
int smc_resample(const double* w, int64_t n, int64_t ndraw, uint64_t seed, uint32_t stream, uint32_t t, int32_t* a,
                 int device);

/* ---- SURVEY 8(f) "next" rows --------------------------------------------------------------------*/
/* log_likelihood(y, model::LinearModel) -> (x_T, Sigma_T, logZ): exact scalar Kalman filter,
 * src/kalman_filter.jl:29-70, for n_theta parameter rows at once (the inner "filter" of the IBIS
 * sampler src/ibis.jl:134-189, and a self-check of linear-Gaussian particle runs).
 * raw [n_theta][6] = (A,B,Q,R,x0,sigma0); out [n_theta][3]. predict_first != 0 is the literal
 * reference loop; 0 starts at x_1 ~ N(x0, sigma0) like bootstrap_filter. */
int smc_kalman_log_likelihood(const double* raw, int64_t n_theta, const double* y, int64_t T, int predict_first,
                              double* out /*[n_theta][3]*/, int device);
/* filtered mean and variance of every state coordinate under the current weights, on the device
 * (README.md:41,51 summaries; src/plotting_utils.jl:116-124 estimated_trend). mean, var: [d][n_theta]. */
int smc_get_moments(smc_handle h, double* mean, double* var);

/* weighted quantiles of state coordinate `component` under the current weights, per filter, on the
 * device: what quantile(smc.x[i], weights(smc.w[i]), [0.25,0.5,0.75]) computes per theta-particle in
 * examples/inflation_example.jl:45-46 (and README.md:41,51 for an unweighted cloud).  Definition: the
 * inverse of the weighted empirical CDF in the filter's integer weights - the smallest particle value v
 * with sum{W_i : x_i <= v} > floor(p * sum W) - no interpolation between particles (StatsBase
 * interpolates; it is not vendored, so that variant is unpinned).  np <= 8; out [n_theta][np];
 * NaN for a collapsed filter. */
int smc_get_quantiles(smc_handle h, int component, const double* p, int np, double* out);

/* ---- host-side helpers (no GPU needed) ---------------------------------------------------------*/
/* simulate(rng, model, T) -> (x, y)                         src/state_space_models.jl:11-26 */
int smc_simulate(int model_id, const double* raw, int64_t T, uint64_t seed, double* x /*[d][T]*/, double* y /*[T]*/);
int smc_model_dim(int model_id);
int smc_model_nraw(int model_id);
/* the segment length smc_create picks for seg = 0: a function of the model family (its state dimension) and n_x alone */
int smc_auto_seg(int model_id, int64_t n_x);
int smc_device_count(void);
/* the spec's elementary functions on the host (parity tests of the host build) */
double smc_host_exp(double x);
double smc_host_log(double x);
void smc_host_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void smc_host_box_muller(const uint32_t w[4], double* z0, double* z1);
/* systematic-resampling targets T_{j0+k} = floor(((j0+k) Dtot + mulhi64(u, Dtot)) / n), k < nk <= 8192, by the
 * division-free evaluation of the kernels; device < 0 evaluates on the host (exactness tests) */
int smc_sys_targets(uint64_t Dtot, uint32_t n, uint64_t u, uint64_t j0, int nk, uint64_t* out, int device);
/* the same functions evaluated on the device for n inputs (math parity tests) */
int smc_device_math(int which /*0 exp,1 log,2 sqrt,3 box-muller z0,4 z1,5 div a/b*/, const double* a, const double* b,
                    int64_t n, double* out, int device);

const char* smc_last_error(void);
const char* smc_version(void);

#ifdef __cplusplus
}
#endif
#endif
