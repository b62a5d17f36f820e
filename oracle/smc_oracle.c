/*
 * smc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never the product path).
 *
 * A scalar, single-threaded, plain-C restatement of the reference's particle-filter
 * hot path, function for function:
 *
 *   orc_normalize              <- normalize            src/particles.jl:5-15
 *   orc_resample               <- resample             src/particles.jl:17-19
 *   orc_bootstrap_filter       <- bootstrap_filter     src/particles.jl:87-105
 *   orc_bootstrap_filter_step  <- bootstrap_filter!    src/particles.jl:107-129
 *   orc_log_likelihood         <- log_likelihood       src/particles.jl:132-147
 *   orc_simulate               <- simulate             src/state_space_models.jl:11-26
 *   model methods              <- initial_dist/transition/observation
 *                                 src/state_space_models.jl:87-109 (LG), :233-259 (UCSV)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY PIN.  The reference is Julia (no toolchain here) and its arithmetic lives in
 * un-vendored, un-pinned Distributions.jl / StatsBase.jl / Random (Project.toml:6-13), with
 * no tests or golden vectors: bit-level parity with Julia is unpinned by construction.
 * This oracle is pinned instead by the reference's own exact Kalman log-likelihood
 * (src/kalman_filter.jl:29-70, restated in oracle/kalman.py): tests/test_oracle.py
 * (test_particle_filter_pinned_by_kalman) checks E[exp(logZ_PF - logZ_KF)] = 1 and mean(logZ_PF) - logZ_KF within MC error.
 *
 * Because the reference's random stream cannot be reproduced, the *specification* of the
 * stream and of every rounding step is defined here (DESIGN.md "Numerical specification")
 * and the HIP kernels must reproduce THIS file bit for bit:
 *   - RNG: Philox4x32-10, key = 64-bit seed, counter = (pair index, stream, t, slot).
 *   - normals: Box-Muller on two 53-bit uniforms; one Philox call serves the particle
 *     pair (2p, 2p+1): z[2p] = R cos(2 pi u2), z[2p+1] = R sin(2 pi u2).
 *   - exp / log / sincos: the polynomial kernels below (IEEE +,-,*,/,sqrt and fma only).
 *   - exp(logw_i) is split as p_i * 2^k_i (k_i integer, p_i in [0.707,1.415]); a segment of
 *     `seg` particles is normalised against the power of two 2^kb, kb = max k_i (instead of
 *     the reference's maximum(logw), particles.jl:6 -- same role, exact to undo), and weights
 *     are carried as fixed point q_i = rint(p_i * 2^(48 + k_i - kb)), so every sum and prefix
 *     sum is an exact integer and independent of the order a parallel machine adds in.
 *   - segments are combined by integer shifts (2^(K - kb), K = max kb) into a second table.
 *     resample (particles.jl:17-19) is the multinomial law of StatsBase.sample(1:n, Weights(w), n):
 *     n iid uniforms through the inverse of the integer weight CDF.  With several segments the
 *     uniforms are generated sorted BY BLOCK (block = seg consecutive children) between break
 *     points built from Gamma variates (resample_breaks below); inside a block the order is iid.
 *     The joint law of the ancestors is exactly Multinomial(n, w); only the (unobservable,
 *     exchangeable) order of the children differs from an unsorted draw.  The stand-alone
 *     resample(w, N) keeps the unsorted iid order of the reference.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; never -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------ */
/* constants                                                                            */
/* ------------------------------------------------------------------------------------ */
#define ORC_LG1D 1
#define ORC_SV1D 2
#define ORC_UCSV3D 3

#define FIX_BITS 48            /* q = rint(w~ * 2^48), w~ in [0,1] relative to segment max */
#define MAX_SEG 8192
#define SIM_STREAM 0xFFFFFFFFu /* stream id reserved for simulate()                       */
#define SLOT_RESAMPLE 0u       /* within-segment pick of child j                          */
#define SLOT_NORMAL0 1u        /* slots 1..d : state normals                              */
#define SLOT_OBS 8u            /* simulate(): observation noise                           */
#define SLOT_COUNT 9u          /* segment pick of draw i (multi-segment filters only)     */
#define SLOT_SYS 10u           /* the one uniform of a systematic resampling step (opt-in) */
#define SLOT_BREAK 16u         /* break points of the blocks: 16+2i normal, 17+2i uniform, i < 8; pair = block */

static const double HALF_LOG2PI = 0x1.d67f1c864beb5p-1;
static const double LOG2PI = 0x1.d67f1c864beb5p+0;
static const double INV_LN2 = 0x1.71547652b82fep+0;
static const double LN2_HI = 0x1.62e42fee00000p-1;
static const double LN2_LO = 0x1.a39ef35793c76p-33;
static const double SQRT2 = 0x1.6a09e667f3bcdp+0;
static const double PIO4 = 0x1.921fb54442d18p-1;
static const double RND_MAGIC = 0x1.8p52;
static const double TWO_M53 = 0x1p-53;
static const double TWO_P64 = 0x1p+64;

/* ------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator of the spec         */
/* ------------------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void draw(uint64_t seed, uint32_t pair, uint32_t stream, uint32_t t, uint32_t slot,
                 uint32_t out[4]) {
    uint32_t ctr[4] = {pair, stream, t, slot};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

/* ------------------------------------------------------------------------------------ */
/* elementary functions of the spec                                                     */
/* ------------------------------------------------------------------------------------ */
static double bits2d(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t d2bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* round to nearest-even integer: signed |v| < 2^51; and v >= 0 of any size (== rint) */
static double rne(double v) { return (v + RND_MAGIC) - RND_MAGIC; }
static double rne_pos(double v) { return v < 0x1p52 ? (v + 0x1p52) - 0x1p52 : v; }

/* exp(x) = p * 2^k with k = rint(x/ln2) (returned as an integral double), p in [0.707,1.415];
 * |x| <= 7e8 */
static double exp_parts(double x, double* kout) {
    double k = rne(x * INV_LN2);
    double r = fma(-k, LN2_HI, x);
    r = fma(-k, LN2_LO, r);
    double p = 0x1.6124613a86d09p-33;            /* 1/13! */
    p = fma(p, r, 0x1.1eed8eff8d898p-29);        /* 1/12! */
    p = fma(p, r, 0x1.ae64567f544e4p-26);
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma(p, r, 0x1.71de3a556c734p-19);
    p = fma(p, r, 0x1.a01a01a01a01ap-16);
    p = fma(p, r, 0x1.a01a01a01a01ap-13);
    p = fma(p, r, 0x1.6c16c16c16c17p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    *kout = k;
    return p;
}

double orc_exp(double x) {
    if (x != x) return x;
    if (!(x > -708.0)) return 0.0;
    if (x > 709.0) return INFINITY;
    double k;
    double p = exp_parts(x, &k);
    return p * bits2d((uint64_t)((int64_t)k + 1023) << 52);
}

/* a log-weight takes part in the normalisation iff it is a number of sane magnitude */
static int lw_alive(double l) { return l == l && fabs(l) <= 7e8; }   /* |k| < 2^30: exponent differences fit an int32 */

/* q = rint(p * 2^(bits + dk)), dk = k_i - kb <= 0 (integral doubles) */
static uint64_t fix_weight(double p, double dk, int bits) {
    if (dk < -(double)(bits + 2)) return 0;
    return (uint64_t)rne_pos(p * bits2d((uint64_t)(bits + (int)dk + 1023) << 52));
}

double orc_log(double x) {
    if (x != x || x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int64_t e = 0;
    if (x < 0x1p-1022) { x *= 0x1p54; e = -54; }
    uint64_t b = d2bits(x);
    e += (int64_t)((b >> 52) & 0x7ff) - 1023;
    double m = bits2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > SQRT2) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = 0x1.642c8590b2164p-4;             /* 2/23 */
    R = fma(R, z, 0x1.8618618618618p-4);         /* 2/21 */
    R = fma(R, z, 0x1.af286bca1af28p-4);
    R = fma(R, z, 0x1.e1e1e1e1e1e1ep-4);
    R = fma(R, z, 0x1.1111111111111p-3);
    R = fma(R, z, 0x1.3b13b13b13b14p-3);
    R = fma(R, z, 0x1.745d1745d1746p-3);
    R = fma(R, z, 0x1.c71c71c71c71cp-3);
    R = fma(R, z, 0x1.2492492492492p-2);
    R = fma(R, z, 0x1.999999999999ap-2);
    R = fma(R, z, 0x1.5555555555555p-1);         /* 2/3 */
    R = R * z;                                   /* log(1+f) = 2s + s*R = f - s*(f - R) */
    double dk = (double)e;
    return dk * LN2_HI - ((s * (f - R) - dk * LN2_LO) - f);
}

/* cos and sin of 2*pi*u, u in [0,1) a multiple of 2^-53 */
void orc_sincos2pi(double u, double* c, double* s) {
    double a = 8.0 * u;
    int oct = (int)a;
    double g = a - (double)oct;
    if (oct & 1) g = 1.0 - g;
    double y = g * PIO4;
    double z = y * y;
    double ps = 0x1.952c77030ad4ap-49;           /* 1/17! */
    ps = fma(ps, z, -0x1.ae7f3e733b81fp-41);
    ps = fma(ps, z, 0x1.6124613a86d09p-33);
    ps = fma(ps, z, -0x1.ae64567f544e4p-26);
    ps = fma(ps, z, 0x1.71de3a556c734p-19);
    ps = fma(ps, z, -0x1.a01a01a01a01ap-13);
    ps = fma(ps, z, 0x1.1111111111111p-7);
    ps = fma(ps, z, -0x1.5555555555555p-3);
    double sy = fma(y * z, ps, y);
    double pc = -0x1.6827863b97d97p-53;          /* -1/18! */
    pc = fma(pc, z, 0x1.ae7f3e733b81fp-45);
    pc = fma(pc, z, -0x1.93974a8c07c9dp-37);
    pc = fma(pc, z, 0x1.1eed8eff8d898p-29);
    pc = fma(pc, z, -0x1.27e4fb7789f5cp-22);
    pc = fma(pc, z, 0x1.a01a01a01a01ap-16);
    pc = fma(pc, z, -0x1.6c16c16c16c17p-10);
    pc = fma(pc, z, 0x1.5555555555555p-5);
    pc = fma(pc, z, -0.5);
    double cy = fma(z, pc, 1.0);
    int swap = (oct + 1) & 2;
    double cc = swap ? sy : cy;
    double ss = swap ? cy : sy;
    if ((oct + 2) & 4) cc = -cc;
    if (oct & 4) ss = -ss;
    *c = cc;
    *s = ss;
}

/* four Philox words -> two independent N(0,1) */
void orc_box_muller(const uint32_t w[4], double* z0, double* z1) {
    uint64_t n1 = (((uint64_t)w[1] << 32) | w[0]) >> 11;
    uint64_t n2 = (((uint64_t)w[3] << 32) | w[2]) >> 11;
    double u1 = (double)(n1 + 1) * TWO_M53;      /* (0,1] */
    double u2 = (double)n2 * TWO_M53;            /* [0,1) */
    double r = sqrt(-2.0 * orc_log(u1));
    double c, s;
    orc_sincos2pi(u2, &c, &s);
    *z0 = r * c;
    *z1 = r * s;
}

/* normals of the particle pair (2p, 2p+1) at (stream,t,slot): z[0] -> 2p, z[1] -> 2p+1 */
static void normal_pair(uint64_t seed, int64_t p, uint32_t stream, uint32_t t, uint32_t slot, double z[2]) {
    uint32_t w[4];
    draw(seed, (uint32_t)p, stream, t, slot, w);
    orc_box_muller(w, &z[0], &z[1]);
}

/* 64 resampling bits of each member of the pair */
static void resample_pair(uint64_t seed, int64_t p, uint32_t stream, uint32_t t, uint32_t slot, uint64_t r[2]) {
    uint32_t w[4];
    draw(seed, (uint32_t)p, stream, t, slot, w);
    r[0] = ((uint64_t)w[1] << 32) | w[0];
    r[1] = ((uint64_t)w[3] << 32) | w[2];
}

static double u128_to_double(uint64_t hi, uint64_t lo) { return (double)hi * TWO_P64 + (double)lo; }

static int ceil_log2(int64_t n) { int k = 0; while (((int64_t)1 << k) < n) ++k; return k; }

/* first j in [0,n) with C[j] > T   (C non-decreasing, C[n-1] > T guaranteed by callers) */
static int64_t upper_bound_u64(const uint64_t* C, int64_t n, uint64_t T) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (C[mid] > T) hi = mid; else lo = mid + 1;
    }
    return lo < n ? lo : n - 1;
}

/* ------------------------------------------------------------------------------------ */
/* state-space models: the 4-function contract of src/state_space_models.jl:30-42       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int id, d;
    double raw[8];
    double der[8];
} orc_model;

int orc_model_dim(int id) { return id == ORC_UCSV3D ? 3 : (id == ORC_LG1D || id == ORC_SV1D) ? 1 : -1; }
int orc_model_nraw(int id) { return id == ORC_LG1D ? 6 : id == ORC_SV1D ? 3 : id == ORC_UCSV3D ? 5 : -1; }

/* derived constants (what Normal(mu, sqrt(Q)) etc. rebuild per particle in the reference) */
static int model_init(orc_model* m, int id, const double* raw) {
    m->id = id;
    m->d = orc_model_dim(id);
    if (m->d < 0) return -1;
    memset(m->raw, 0, sizeof m->raw);
    memset(m->der, 0, sizeof m->der);
    memcpy(m->raw, raw, sizeof(double) * (size_t)orc_model_nraw(id));
    if (id == ORC_LG1D) {
        /* raw = (A,B,Q,R,x0,sigma0), Q R sigma0 are VARIANCES: ssm.jl:93,102,108 */
        double sR = sqrt(raw[3]);
        m->der[0] = sqrt(raw[2]);                 /* sQ  */
        m->der[1] = sR;
        m->der[2] = sqrt(raw[5]);                 /* s0  */
        m->der[3] = 1.0 / sR;
        m->der[4] = -HALF_LOG2PI - orc_log(sR);   /* logpdf constant */
    } else if (id == ORC_SV1D) {
        /* raw = (mu, rho, sigma): x1 ~ N(mu, sigma^2/(1-rho^2)) */
        m->der[0] = raw[2] / sqrt(1.0 - raw[1] * raw[1]);
    }
    return 0;
}

/* rand(initial_dist(model)) : ssm.jl:105-109, :249-259 */
static void model_initial(const orc_model* m, const double* z, double* x) {
    switch (m->id) {
    case ORC_LG1D: x[0] = fma(m->der[2], z[0], m->raw[4]); break;
    case ORC_SV1D: x[0] = fma(m->der[0], z[0], m->raw[0]); break;
    case ORC_UCSV3D:
        /* raw = (gamma_eps, gamma_eta, x0, lse0, lsn0); gamma is a STD-DEV: ssm.jl:239-240 */
        x[0] = fma(orc_exp(0.5 * m->raw[3]), z[0], m->raw[2]);
        x[1] = fma(m->raw[0], z[1], m->raw[3]);
        x[2] = fma(m->raw[1], z[2], m->raw[4]);
        break;
    }
}

/* rand(transition(model, xp)) : ssm.jl:87-94, :233-242 (x' uses the PREVIOUS log-vol) */
static void model_transition(const orc_model* m, const double* xp, const double* z, double* x) {
    switch (m->id) {
    case ORC_LG1D: x[0] = fma(m->der[0], z[0], m->raw[0] * xp[0]); break;
    case ORC_SV1D: x[0] = fma(m->raw[2], z[0], fma(m->raw[1], xp[0] - m->raw[0], m->raw[0])); break;
    case ORC_UCSV3D:
        x[0] = fma(orc_exp(0.5 * xp[1]), z[0], xp[0]);
        x[1] = fma(m->raw[0], z[1], xp[1]);
        x[2] = fma(m->raw[1], z[2], xp[2]);
        break;
    }
}

/* logpdf(observation(model, x), y) : ssm.jl:96-103, :244-247; Normal logpdf =
 * -(z^2 + log 2pi)/2 - log(sigma), z = (y - mean)/sigma */
static double model_logobs(const orc_model* m, const double* x, double y) {
    double z, c;
    switch (m->id) {
    case ORC_LG1D:
        z = (y - m->raw[1] * x[0]) * m->der[3];
        c = m->der[4];
        break;
    case ORC_SV1D:
        z = y * orc_exp(-0.5 * x[0]);
        c = fma(-0.5, x[0], -HALF_LOG2PI);
        break;
    default:
        z = (y - x[0]) * orc_exp(-0.5 * x[2]);
        c = fma(-0.5, x[2], -HALF_LOG2PI);
        break;
    }
    return fma(-0.5 * z, z, c);
}

/* mean and sd of the observation density (simulate only) */
static void model_obs_moments(const orc_model* m, const double* x, double* mean, double* sd) {
    switch (m->id) {
    case ORC_LG1D: *mean = m->raw[1] * x[0]; *sd = m->der[1]; break;
    case ORC_SV1D: *mean = 0.0; *sd = orc_exp(0.5 * x[0]); break;
    default: *mean = x[0]; *sd = orc_exp(0.5 * x[2]); break;
    }
}

/* simulate(rng, model, T) : ssm.jl:11-26.  x is [d][T]. */
int orc_simulate(int id, const double* raw, int T, uint64_t seed, double* x, double* y) {
    orc_model m;
    if (model_init(&m, id, raw)) return -1;
    double xc[3], xn[3], z[3], mean, sd;
    for (int t = 0; t < T; ++t) {
        double zp[2];
        for (int k = 0; k < m.d; ++k) { normal_pair(seed, 0, SIM_STREAM, (uint32_t)t, SLOT_NORMAL0 + (uint32_t)k, zp); z[k] = zp[0]; }
        if (t == 0) model_initial(&m, z, xn); else model_transition(&m, xc, z, xn);
        model_obs_moments(&m, xn, &mean, &sd);
        normal_pair(seed, 0, SIM_STREAM, (uint32_t)t, SLOT_OBS, zp);
        y[t] = fma(sd, zp[0], mean);
        for (int k = 0; k < m.d; ++k) { xc[k] = xn[k]; x[(size_t)k * T + t] = xn[k]; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* A1/A2 stand-alone: normalize(logw) and resample(w, N)    particles.jl:5-19            */
/* (single level; used for the outer theta-level `reweight`/`resample`)                  */
/* ------------------------------------------------------------------------------------ */
static int fix_bits_for(int64_t n) { int k = 61 - ceil_log2(n); return k > FIX_BITS ? FIX_BITS : k; }

static uint64_t to_fix(double wrel, double scale) { return (uint64_t)rne_pos(wrel * scale); }

int orc_normalize(const double* logw, int64_t n, double* w, double* logmu, double* ess) {
    if (n <= 0) return -1;
    const int K = fix_bits_for(n);
    double* p = (double*)malloc(sizeof(double) * (size_t)n);
    double* k = (double*)malloc(sizeof(double) * (size_t)n);
    uint64_t* q = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    double kmax = -INFINITY;                                     /* maxw = maximum(logw)     */
    for (int64_t i = 0; i < n; ++i) {
        if (lw_alive(logw[i])) { p[i] = exp_parts(logw[i], &k[i]); if (k[i] > kmax) kmax = k[i]; }
        else { p[i] = 0.0; k[i] = -INFINITY; }
    }
    uint64_t S = 0; u128 S2 = 0;
    for (int64_t i = 0; i < n; ++i) {                            /* w = exp.(logw .- maxw)   */
        q[i] = (k[i] > -INFINITY) ? fix_weight(p[i], k[i] - kmax, K) : 0;
        S += q[i];                                               /* sumw = sum(w)            */
        S2 += (u128)q[i] * q[i];
    }
    double Sd = (double)S;
    *logmu = S ? fma(kmax, LN2_HI, fma(kmax, LN2_LO, orc_log(Sd * bits2d((uint64_t)(1023 - K) << 52)))) - orc_log((double)n)
               : -INFINITY;                                      /* maxw+log(sumw)-log(N)    */
    for (int64_t i = 0; i < n; ++i) w[i] = S ? (double)q[i] / Sd : 0.0;   /* w = w/sumw      */
    *ess = S ? (Sd * Sd) / u128_to_double((uint64_t)(S2 >> 64), (uint64_t)S2) : 0.0;   /* 1/sum(w.^2) */
    free(p); free(k); free(q);
    return 0;
}

/* N iid draws from Categorical(w), unsorted: sample(1:n, Weights(w), N). 0-based output. */
int orc_resample(const double* w, int64_t n, int64_t ndraw, uint64_t seed, uint32_t stream, uint32_t t,
                 int64_t* a) {
    if (n <= 0) return -1;
    double wmax = 0.0;
    for (int64_t i = 0; i < n; ++i) if (w[i] > wmax) wmax = w[i];
    if (!(wmax > 0.0) || wmax == INFINITY) return -2;
    int K = fix_bits_for(n);
    double scale = bits2d((uint64_t)(K + 1023) << 52);
    uint64_t* C = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    uint64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        double r = w[i] / wmax;
        acc += (r == r && r > 0.0) ? to_fix(r, scale) : 0;
        C[i] = acc;
    }
    for (int64_t i = 0; i < ndraw; ++i) {
        uint64_t r[2];
        resample_pair(seed, i >> 1, stream, t, SLOT_RESAMPLE, r);
        uint64_t T = (uint64_t)(((u128)r[i & 1] * acc) >> 64);
        a[i] = upper_bound_u64(C, n, T);
    }
    free(C);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* the filter's weights object: segmented fixed-point normalisation                      */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n;      /* particles                                   */
    int seg, nseg;  /* segment length (power of two), #segments    */
    int SH;         /* extra right shift of the segment table: max(0, ceil_log2(nseg*seg) - 14) */
    uint64_t* C;    /* [nseg*seg] segment-local inclusive sums of q */
    double* kb;     /* [nseg] segment exponent (integral; -inf if no live particle) */
    uint64_t *S, *S2hi, *S2lo;   /* [nseg] sum q, sum q^2 (128 bit) */
    /* global combine */
    double K;       /* max kb */
    uint64_t* Dcum; /* [nseg] inclusive sums of Q_b = S_b >> sh_b  */
    int64_t* cnt;   /* [nseg] children per segment of the current resampling step */
    uint64_t Dtot, Rtot;
    double logmu, ess;
} orc_weights;

static void weights_alloc(orc_weights* W, int64_t n, int seg) {
    W->n = n; W->seg = seg; W->nseg = (int)((n + seg - 1) / seg);
    int sh = ceil_log2((int64_t)W->nseg * seg) - 14;
    W->SH = sh > 0 ? sh : 0;
    size_t ns = (size_t)W->nseg;
    W->C = (uint64_t*)calloc(ns * (size_t)seg, 8);
    W->kb = (double*)calloc(ns, 8);
    W->S = (uint64_t*)calloc(ns, 8);
    W->S2hi = (uint64_t*)calloc(ns, 8);
    W->S2lo = (uint64_t*)calloc(ns, 8);
    W->Dcum = (uint64_t*)calloc(ns, 8);
    W->cnt = (int64_t*)calloc(ns, sizeof(int64_t));
}
static void weights_free(orc_weights* W) {
    free(W->C); free(W->kb); free(W->S); free(W->S2hi); free(W->S2lo); free(W->Dcum); free(W->cnt);
}

/* normalize(logw) of particles.jl:5-15, segment by segment, then the integer combine */
static void weights_normalize(orc_weights* W, const double* logw) {
    const int seg = W->seg;
    double p[MAX_SEG], k[MAX_SEG];
    for (int b = 0; b < W->nseg; ++b) {
        int64_t i0 = (int64_t)b * seg, i1 = i0 + seg < W->n ? i0 + seg : W->n;
        double kb = -INFINITY;
        for (int64_t i = i0; i < i1; ++i) {
            if (lw_alive(logw[i])) { p[i - i0] = exp_parts(logw[i], &k[i - i0]); if (k[i - i0] > kb) kb = k[i - i0]; }
            else k[i - i0] = -INFINITY;
        }
        uint64_t acc = 0; u128 acc2 = 0;
        for (int64_t i = i0; i < i0 + seg; ++i) {
            uint64_t q = 0;
            if (i < i1 && k[i - i0] > -INFINITY) q = fix_weight(p[i - i0], k[i - i0] - kb, FIX_BITS);
            acc += q; acc2 += (u128)q * q;
            W->C[i] = acc;
        }
        W->kb[b] = kb; W->S[b] = acc; W->S2hi[b] = (uint64_t)(acc2 >> 64); W->S2lo[b] = (uint64_t)acc2;
    }
    /* combine: what the next step's kernel prologue (or the finalize kernel) computes */
    double K = -INFINITY;
    for (int b = 0; b < W->nseg; ++b) if (W->kb[b] > K) K = W->kb[b];
    W->K = K;
    uint64_t D = 0, R = 0;
    for (int b = 0; b < W->nseg; ++b) {
        double dk = K - W->kb[b];                        /* >= 0, integral, inf or nan if dead */
        int sh = (dk >= 0.0 && dk < 64.0) ? (int)dk + W->SH : 64;
        if (sh > 64) sh = 64;
        uint64_t Qb = sh < 64 ? W->S[b] >> sh : 0;
        int sh2 = 2 * (sh - W->SH) + 49 + W->SH;          /* shift of the 128-bit sum of squares */
        u128 s2 = ((u128)W->S2hi[b] << 64) | W->S2lo[b];
        uint64_t Rb = (sh < 64 && sh2 < 128) ? (uint64_t)(s2 >> sh2) : 0;
        D += Qb; R += Rb;
        W->Dcum[b] = D;
    }
    W->Dtot = D; W->Rtot = R;
    double Dd = (double)D * bits2d((uint64_t)(1023 + W->SH - 48) << 52);
    double Rd = (double)R * bits2d((uint64_t)(1023 + W->SH - 47) << 52);
    W->logmu = D ? fma(K, LN2_HI, fma(K, LN2_LO, orc_log(Dd))) - orc_log((double)W->n) : -INFINITY;
    W->ess = R ? Dd * Dd / Rd : 0.0;
}

/* Gamma(m, 1), integer shape m >= 1, in 2^-32 fixed point (Marsaglia & Tsang 2000: d = m - 1/3,
 * c = 1/sqrt(9d); x ~ N(0,1), v = (1 + c x)^3, accept if u < 1 - 0.0331 x^4 or
 * log u < x^2/2 + d (1 - v + log v)).  Counter-based: attempt i of block w uses slots 16+2i, 17+2i. */
static uint64_t gamma_fix(uint64_t seed, uint32_t w, uint32_t stream, uint32_t t, int64_t m) {
    const double d = (double)m - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double G = d;                                   /* (all 8 attempts rejected: probability ~1e-15) */
    for (uint32_t it = 0; it < 8; ++it) {
        double z[2];
        normal_pair(seed, w, stream, t, SLOT_BREAK + 2 * it, z);
        const double x = z[0], vv = 1.0 + c * x;
        if (!(vv > 0.0)) continue;
        const double v3 = vv * vv * vv, x2 = x * x;
        uint32_t wd[4];
        draw(seed, w, stream, t, SLOT_BREAK + 2 * it + 1, wd);
        const double u = (double)((((((uint64_t)wd[1] << 32) | wd[0]) >> 11)) + 1) * TWO_M53;
        if (u < 1.0 - 0.0331 * (x2 * x2) || orc_log(u) < 0.5 * x2 + d * ((1.0 - v3) + orc_log(v3))) { G = d * v3; break; }
    }
    const uint64_t g = (uint64_t)rne_pos(G * 0x1p32);
    return g ? g : 1;
}

/* Break points of the n sorted iid uniforms of a resampling step at the block boundaries (block =
 * the seg consecutive children a workgroup owns).  With E_i iid Exp(1), the order statistics are
 * U_(k) = (E_1+..+E_k)/(E_1+..+E_{n+1}); a block of m ranks contributes a Gamma(m) to these sums, so
 * the block's largest uniform is F_{w+1} = (g_0+..+g_w)/(g_0+..+g_{B-1}+e) and - given the break
 * points - the other m-1 uniforms of the block are iid on (F_w, F_{w+1}).  F in 2^-64 fixed point. */
static void resample_breaks(const orc_weights* W, uint64_t seed, uint32_t stream, uint32_t t, uint64_t* F /*[nseg+1]*/) {
    const int B = W->nseg;
    uint64_t S = 0;
    for (int w = 0; w < B; ++w) {
        int64_t m = W->n - (int64_t)w * W->seg;
        if (m > W->seg) m = W->seg;
        F[w + 1] = gamma_fix(seed, (uint32_t)w, stream, t, m);      /* g_w for now */
        S += F[w + 1];
    }
    uint32_t wd[4];
    draw(seed, (uint32_t)B, stream, t, SLOT_BREAK + 1, wd);
    const double u = (double)((((((uint64_t)wd[1] << 32) | wd[0]) >> 11)) + 1) * TWO_M53;
    uint64_t e = (uint64_t)rne_pos(-orc_log(u) * 0x1p32);
    S += e ? e : 1;
    uint64_t P = 0;
    F[0] = 0;
    for (int w = 0; w < B; ++w) {
        P += F[w + 1];
        F[w + 1] = (uint64_t)((((u128)P) << 64) / S);
    }
}

/* a = resample(weights): ancestors of all n children, exactly Multinomial(n, w) - the n iid uniforms
 * are generated block by block between their break points (above), so the children come out sorted by
 * block of the weight CDF; inside a block the order is iid.  One segment: n iid picks, no breaks.
 * (seed, stream, t) select the Philox counters. */
static void weights_resample(orc_weights* W, uint64_t seed, uint32_t stream, uint32_t t, int64_t* a) {
    const int64_t n = W->n;
    uint64_t r[2];
    if (W->Dtot == 0) {                                          /* collapsed filter: identity */
        for (int64_t i = 0; i < n; ++i) a[i] = i;
        return;
    }
    if (W->nseg == 1) {
        for (int64_t j = 0; j < n; ++j) {
            if (!(j & 1)) resample_pair(seed, j >> 1, stream, t, SLOT_RESAMPLE, r);
            uint64_t T2 = (uint64_t)(((u128)r[j & 1] * W->S[0]) >> 64);
            a[j] = upper_bound_u64(W->C, W->seg, T2);
        }
        return;
    }
    uint64_t* F = (uint64_t*)malloc(8 * ((size_t)W->nseg + 1));
    resample_breaks(W, seed, stream, t, F);
    for (int w = 0; w < W->nseg; ++w) {
        const uint64_t lo = (uint64_t)(((u128)F[w] * W->Dtot) >> 64), hi = (uint64_t)(((u128)F[w + 1] * W->Dtot) >> 64);
        int64_t m = n - (int64_t)w * W->seg;
        if (m > W->seg) m = W->seg;
        for (int64_t k = 0; k < m; ++k) {
            const int64_t j = (int64_t)w * W->seg + k;
            if (!(j & 1)) resample_pair(seed, j >> 1, stream, t, SLOT_RESAMPLE, r);
            const uint64_t T = (k == m - 1) ? hi : lo + (uint64_t)(((u128)r[j & 1] * (hi - lo)) >> 64);
            const int b = (int)upper_bound_u64(W->Dcum, W->nseg, T);
            const uint64_t T2 = T - (b ? W->Dcum[b - 1] : 0);
            const double dk = W->K - W->kb[b];
            int sh = (dk >= 0.0 && dk < 64.0) ? (int)dk + W->SH : 64;
            if (sh > 64) sh = 64;
            const uint64_t thr = sh < 64 ? ((T2 + 1) << sh) - 1 : 0;      /* (C >> sh) > T2  <=>  C > thr */
            a[j] = (int64_t)b * W->seg + upper_bound_u64(W->C + (size_t)b * W->seg, W->seg, thr);
        }
    }
    free(F);
}

/* OPT-IN alternative to resample(): systematic resampling (one uniform u per step; child j takes the
 * point (j + u)/n of the weight CDF).  Not the reference's law (its resample is multinomial,
 * particles.jl:17-19) - lower variance, same expectation E[#children of i] = n w_i - so never the
 * default.  Exact integer definition: with the combined table (Dcum, Dtot) and v0 = mulhi64(u, Dtot),
 *     T_j = floor((j * Dtot + v0) / n)   in [0, Dtot),  j = 0..n-1   (increasing in j)
 * picks segment b = first with Dcum[b] > T_j, and inside it the first particle a whose cumulative
 * weight in table units (C_a >> sh_b) exceeds T_j - Dcum[b-1].  Children come out sorted by ancestor. */
static void weights_resample_systematic(orc_weights* W, uint64_t seed, uint32_t stream, uint32_t t, int64_t* a) {
    const int64_t n = W->n;
    if (W->Dtot == 0) {
        for (int64_t i = 0; i < n; ++i) a[i] = i;
        return;
    }
    uint64_t r[2];
    resample_pair(seed, 0, stream, t, SLOT_SYS, r);
    const uint64_t v0 = (uint64_t)(((u128)r[0] * W->Dtot) >> 64);
    for (int64_t j = 0; j < n; ++j) {
        const uint64_t T = (uint64_t)(((u128)(uint64_t)j * W->Dtot + v0) / (u128)(uint64_t)n);
        const int b = (int)upper_bound_u64(W->Dcum, W->nseg, T);
        const uint64_t T2 = T - (b ? W->Dcum[b - 1] : 0);
        const double dk = W->K - W->kb[b];
        int sh = (dk >= 0.0 && dk < 64.0) ? (int)dk + W->SH : 64;
        if (sh > 64) sh = 64;
        const uint64_t thr = sh < 64 ? ((T2 + 1) << sh) - 1 : 0;      /* (C >> sh) > T2  <=>  C > thr */
        a[j] = (int64_t)b * W->seg + upper_bound_u64(W->C + (size_t)b * W->seg, W->seg, thr);
    }
}

/* dense normalised weights w_i (what the reference's normalize returns as `w`) */
static void weights_dense(const orc_weights* W, double* w) {
    double Dd = (double)W->Dtot * bits2d((uint64_t)(1023 + W->SH - 48) << 52);
    for (int b = 0; b < W->nseg; ++b) {
        double dk = W->K - W->kb[b];
        double sc = (dk >= 0.0 && dk < 900.0) ? bits2d((uint64_t)(1023 - 48 - (int)dk) << 52) : 0.0;
        for (int j = 0; j < W->seg; ++j) {
            int64_t i = (int64_t)b * W->seg + j;
            if (i >= W->n) break;
            uint64_t q = W->C[i] - (j ? W->C[i - 1] : 0);
            w[i] = W->Dtot ? ((double)q * sc) / Dd : 0.0;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* the filter                                                                            */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    orc_model model;
    int64_t n;
    uint64_t seed;
    uint32_t stream, t;     /* t = index of the NEXT observation (0 before bootstrap_filter) */
    double *x, *xp, *logw;  /* x is [d][n] */
    int64_t* a;
    orc_weights W;
    int systematic;         /* 0: multinomial resampling (the reference's law); 1: opt-in systematic */
} orc_filter;

void orc_filter_set_systematic(orc_filter* f, int on) { f->systematic = on ? 1 : 0; }

int orc_auto_seg(int model, int64_t n) {   /* the same rule as smc_auto_seg (the segment length is part of the spec) */
    const int d3 = model == ORC_UCSV3D;
    if (n > (int64_t)16384 * 4096) return 8192;   /* at most 16384 segments */
    if (n > (int64_t)16384 * 2048) return 4096;
    if (n > (int64_t)16384 * 1024) return 2048;
    if (n > ((int64_t)1 << 19)) return d3 ? 1024 : 2048;
    if (n > ((int64_t)1 << 17)) return 1024;
    if (n > ((int64_t)1 << 15)) return 512;
    if (n > (d3 ? 4096 : MAX_SEG)) return 256;
    int s = 256;
    while (s < n) s <<= 1;
    return s;
}

orc_filter* orc_filter_create(int model, const double* raw, int64_t n, int seg, uint64_t seed, uint32_t stream) {
    if (n <= 0) return NULL;
    if (seg == 0) seg = orc_auto_seg(model, n);
    if (seg < 2 || seg > MAX_SEG || (seg & (seg - 1))) return NULL;
    orc_filter* f = (orc_filter*)calloc(1, sizeof *f);
    if (model_init(&f->model, model, raw)) { free(f); return NULL; }
    f->n = n; f->seed = seed; f->stream = stream; f->t = 0;
    size_t d = (size_t)f->model.d;
    f->x = (double*)calloc(d * (size_t)n, 8);
    f->xp = (double*)calloc(d * (size_t)n, 8);
    f->logw = (double*)calloc((size_t)n, 8);
    f->a = (int64_t*)calloc((size_t)n, 8);
    weights_alloc(&f->W, n, seg);
    return f;
}

void orc_filter_destroy(orc_filter* f) {
    if (!f) return;
    free(f->x); free(f->xp); free(f->logw); free(f->a);
    weights_free(&f->W);
    free(f);
}

/* smc.model(theta[m]) changed (PMMH accept / per-step parameter refresh): same particles, new parameters */
int orc_filter_set_params(orc_filter* f, const double* raw) { return model_init(&f->model, f->model.id, raw); }

/* value copy of the filter STATE (x cloud, weights, t) from src into dst; dst keeps its own seed,
 * stream and parameters (resample!(smc) smc_samplers.jl:74-84 and the PMMH accept :130-133) */
int orc_filter_copy_state(orc_filter* dst, const orc_filter* src) {
    if (dst->n != src->n || dst->W.seg != src->W.seg || dst->model.d != src->model.d) return -1;
    const size_t n = (size_t)dst->n, d = (size_t)dst->model.d, ns = (size_t)dst->W.nseg;
    memcpy(dst->x, src->x, 8 * d * n);
    memcpy(dst->logw, src->logw, 8 * n);
    memcpy(dst->a, src->a, 8 * n);
    memcpy(dst->W.C, src->W.C, 8 * ns * (size_t)dst->W.seg);
    memcpy(dst->W.kb, src->W.kb, 8 * ns);
    memcpy(dst->W.S, src->W.S, 8 * ns);
    memcpy(dst->W.S2hi, src->W.S2hi, 8 * ns);
    memcpy(dst->W.S2lo, src->W.S2lo, 8 * ns);
    memcpy(dst->W.Dcum, src->W.Dcum, 8 * ns);
    dst->W.K = src->W.K; dst->W.Dtot = src->W.Dtot; dst->W.Rtot = src->W.Rtot;
    dst->W.logmu = src->W.logmu; dst->W.ess = src->W.ess;
    dst->t = src->t;
    return 0;
}

/* serialise the filter STATE (what orc_filter_copy_state copies) to / from 8-byte words: used by the tests of
 * the theta-sharded online sampler to move filters between ranks */
int64_t orc_filter_state_words(const orc_filter* f) {
    const int64_t n = f->n, d = f->model.d, ns = f->W.nseg;
    return d * n + n + n + ns * f->W.seg + 5 * ns + 8;
}
static uint64_t* put(uint64_t* b, const void* src, size_t words) { memcpy(b, src, 8 * words); return b + words; }
static const uint64_t* get(const uint64_t* b, void* dst, size_t words) { memcpy(dst, b, 8 * words); return b + words; }
void orc_filter_export(const orc_filter* f, uint64_t* b) {
    const size_t n = (size_t)f->n, d = (size_t)f->model.d, ns = (size_t)f->W.nseg;
    b = put(b, f->x, d * n); b = put(b, f->logw, n); b = put(b, f->a, n);
    b = put(b, f->W.C, ns * (size_t)f->W.seg); b = put(b, f->W.kb, ns); b = put(b, f->W.S, ns);
    b = put(b, f->W.S2hi, ns); b = put(b, f->W.S2lo, ns); b = put(b, f->W.Dcum, ns);
    b = put(b, &f->W.K, 1); b = put(b, &f->W.Dtot, 1); b = put(b, &f->W.Rtot, 1);
    b = put(b, &f->W.logmu, 1); b = put(b, &f->W.ess, 1);
    uint64_t t = f->t; b = put(b, &t, 1);
}
void orc_filter_import(orc_filter* f, const uint64_t* b) {
    const size_t n = (size_t)f->n, d = (size_t)f->model.d, ns = (size_t)f->W.nseg;
    b = get(b, f->x, d * n); b = get(b, f->logw, n); b = get(b, f->a, n);
    b = get(b, f->W.C, ns * (size_t)f->W.seg); b = get(b, f->W.kb, ns); b = get(b, f->W.S, ns);
    b = get(b, f->W.S2hi, ns); b = get(b, f->W.S2lo, ns); b = get(b, f->W.Dcum, ns);
    b = get(b, &f->W.K, 1); b = get(b, &f->W.Dtot, 1); b = get(b, &f->W.Rtot, 1);
    b = get(b, &f->W.logmu, 1); b = get(b, &f->W.ess, 1);
    uint64_t t; b = get(b, &t, 1); f->t = (uint32_t)t;
}

void orc_filter_reseed(orc_filter* f, uint64_t seed, uint32_t stream) { f->seed = seed; f->stream = stream; f->t = 0; }
/* smc_reseed / smc_set_streams in the middle of a series: new Philox key / stream id, the time index goes on */
void orc_filter_set_rng(orc_filter* f, uint64_t seed, uint32_t stream) { f->seed = seed; f->stream = stream; }

/* bootstrap_filter(N, y, model)  particles.jl:87-105  -> logmu */
double orc_bootstrap_filter(orc_filter* f, double y) {
    const int d = f->model.d;
    const int64_t n = f->n;
    double z[3], xi[3], zp[3][2];
    for (int64_t i = 0; i < n; ++i) {
        if (!(i & 1)) for (int k = 0; k < d; ++k) normal_pair(f->seed, i >> 1, f->stream, 0, SLOT_NORMAL0 + (uint32_t)k, zp[k]);
        for (int k = 0; k < d; ++k) z[k] = zp[k][i & 1];
        model_initial(&f->model, z, xi);                        /* x[i] = rand(initial_dist)        */
        for (int k = 0; k < d; ++k) f->x[(size_t)k * n + i] = xi[k];
        f->logw[i] = model_logobs(&f->model, xi, y);            /* logw[i] = logpdf(observation, y) */
        f->a[i] = i;
    }
    weights_normalize(&f->W, f->logw);                          /* logmu,w,_ = normalize(logw)      */
    f->t = 1;
    return f->W.logmu;
}

/* bootstrap_filter!(x, w, y, model)  particles.jl:107-129  -> (logmu, ess) */
double orc_bootstrap_filter_step(orc_filter* f, double y, double* ess) {
    const int d = f->model.d;
    const int64_t n = f->n;
    double z[3], xpi[3], xi[3], zp[3][2];
    if (f->systematic) weights_resample_systematic(&f->W, f->seed, f->stream, f->t, f->a);
    else weights_resample(&f->W, f->seed, f->stream, f->t, f->a);    /* a = resample(weights)            */
    for (int k = 0; k < d; ++k)                                 /* xp = deepcopy(x[a])              */
        for (int64_t i = 0; i < n; ++i) f->xp[(size_t)k * n + i] = f->x[(size_t)k * n + f->a[i]];
    for (int64_t i = 0; i < n; ++i) {
        if (!(i & 1)) for (int k = 0; k < d; ++k) normal_pair(f->seed, i >> 1, f->stream, f->t, SLOT_NORMAL0 + (uint32_t)k, zp[k]);
        for (int k = 0; k < d; ++k) {
            z[k] = zp[k][i & 1];
            xpi[k] = f->xp[(size_t)k * n + i];
        }
        model_transition(&f->model, xpi, z, xi);                /* x[i] = rand(transition(xp[i]))   */
        for (int k = 0; k < d; ++k) f->x[(size_t)k * n + i] = xi[k];
        f->logw[i] = model_logobs(&f->model, xi, y);            /* logw[i] = logpdf(observation, y) */
    }
    weights_normalize(&f->W, f->logw);                          /* return normalize(logw)           */
    f->t += 1;
    if (ess) *ess = f->W.ess;
    return f->W.logmu;
}

/* log_likelihood(N, y, model)  particles.jl:132-147 -> logZ ; traces optional */
double orc_log_likelihood(orc_filter* f, const double* y, int T, double* logmu_trace, double* ess_trace) {
    double logZ = orc_bootstrap_filter(f, y[0]);
    if (logmu_trace) logmu_trace[0] = logZ;
    if (ess_trace) ess_trace[0] = f->W.ess;
    for (int t = 1; t < T; ++t) {
        double ess, logmu = orc_bootstrap_filter_step(f, y[t], &ess);
        logZ += logmu;
        if (logmu_trace) logmu_trace[t] = logmu;
        if (ess_trace) ess_trace[t] = ess;
    }
    return logZ;
}

/* x [d][n], normalised w [n], last ancestors a [n] (0-based), logw [n]; any pointer may be NULL */
void orc_filter_get_state(const orc_filter* f, double* x, double* w, int64_t* a, double* logw) {
    if (x) memcpy(x, f->x, sizeof(double) * (size_t)f->model.d * (size_t)f->n);
    if (w) weights_dense(&f->W, w);
    if (a) memcpy(a, f->a, sizeof(int64_t) * (size_t)f->n);
    if (logw) memcpy(logw, f->logw, sizeof(double) * (size_t)f->n);
}

double orc_filter_ess(const orc_filter* f) { return f->W.ess; }
int orc_filter_seg(const orc_filter* f) { return f->W.seg; }

/* raw internal weights state, for kernel-level parity tests: C [nseg*seg], kb/S/S2hi/S2lo [nseg] */
void orc_filter_get_weights_raw(const orc_filter* f, uint64_t* C, double* m, uint64_t* S, uint64_t* S2hi,
                                uint64_t* S2lo) {
    size_t ns = (size_t)f->W.nseg;
    if (C) memcpy(C, f->W.C, 8 * ns * (size_t)f->W.seg);
    if (m) memcpy(m, f->W.kb, 8 * ns);
    if (S) memcpy(S, f->W.S, 8 * ns);
    if (S2hi) memcpy(S2hi, f->W.S2hi, 8 * ns);
    if (S2lo) memcpy(S2lo, f->W.S2lo, 8 * ns);
}

/* batched A8 convenience: n_theta independent filters, stream = stream0 + m. logZ [n_theta]. */
int orc_log_likelihood_batch(int model, const double* raw /*[n_theta][nraw]*/, int n_theta, int64_t n, int seg,
                             uint64_t seed, uint32_t stream0, const double* y, int T, double* logZ) {
    int nraw = orc_model_nraw(model);
    if (nraw < 0) return -1;
    for (int m = 0; m < n_theta; ++m) {
        orc_filter* f = orc_filter_create(model, raw + (size_t)m * nraw, n, seg, seed, stream0 + (uint32_t)m);
        if (!f) return -2;
        logZ[m] = orc_log_likelihood(f, y, T, NULL, NULL);
        orc_filter_destroy(f);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* exact scalar Kalman filter  src/kalman_filter.jl:29-53 (step), :55-70 (loop)           */
/* raw = (A,B,Q,R,x0,sigma0).  predict_first != 0 is the literal reference loop (predict   */
/* before the first update); 0 starts from x_1 ~ N(x0, sigma0) like bootstrap_filter.      */
/* out = (x_T, Sigma_T, logZ).  Operation order is part of the spec (no fma).              */
/* ------------------------------------------------------------------------------------ */
void orc_kalman_log_likelihood(const double* raw, const double* y, int64_t T, int predict_first, double out[3]) {
    const double A = raw[0], B = raw[1], Q = raw[2], R = raw[3];
    double x = raw[4], S = raw[5], logZ = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        if (predict_first || t > 0) { x = A * x; S = (A * A) * S + Q; }           /* kalman_filter.jl:39-40 */
        const double s = (B * B) * S + R, dy = y[t] - B * x;                      /* :42-43 */
        const double K = S * B, inv = 1.0 / s;
        x = x + (K * inv) * dy;                                                   /* :46 */
        S = S - (K * K) * inv;                                                    /* :47 */
        logZ += -0.5 * (LOG2PI + orc_log(s) + (dy / s) * dy);                     /* :50-52 */
    }
    out[0] = x; out[1] = S; out[2] = logZ;
}

/* ------------------------------------------------------------------------------------ */
/* PMMH rejuvenation pieces   rejuvenate!  src/smc_samplers.jl:103-146                    */
/* (the loop over parameter particles and chain positions is tests/oracle_backend.py)     */
/* ------------------------------------------------------------------------------------ */
#define SLOT_PMMH_Z 32u   /* proposal normals of a parameter particle: Philox pair k holds z[2k], z[2k+1] */
#define SLOT_PMMH_U 33u   /* the uniform of its accept test */
#define ORC_PRIOR_UNIFORM 1      /* par = (lo, hi) */
#define ORC_PRIOR_NORMAL 2       /* par = (mu, sigma) */
#define ORC_PRIOR_TRUNCNORMAL 3  /* par = (mu, sigma, lo, hi, log(Phi(b) - Phi(a))) */
#define ORC_PRIOR_LOGNORMAL 4    /* par = (mu, sigma) of log x */

/* insupport(prior_i, x)  smc_samplers.jl:116 (closed intervals as in Distributions.jl) */
int orc_prior_insupport(int fam, const double* par, double x) {
    switch (fam) {
    case ORC_PRIOR_UNIFORM: return par[0] <= x && x <= par[1];
    case ORC_PRIOR_NORMAL: return isfinite(x);
    case ORC_PRIOR_TRUNCNORMAL: return par[2] <= x && x <= par[3];
    case ORC_PRIOR_LOGNORMAL: return x > 0.0 && isfinite(x);
    }
    return 0;
}
/* logpdf(prior_i, x)  smc_samplers.jl:123;  -inf outside the support */
double orc_prior_logpdf(int fam, const double* par, double x) {
    if (!orc_prior_insupport(fam, par, x)) return -INFINITY;
    if (fam == ORC_PRIOR_UNIFORM) return -orc_log(par[1] - par[0]);
    if (fam == ORC_PRIOR_LOGNORMAL) {
        const double lx = orc_log(x), z = (lx - par[0]) / par[1];
        return (-0.5 * (z * z + LOG2PI) - orc_log(par[1])) - lx;
    }
    const double z = (x - par[0]) / par[1];
    const double l = -0.5 * (z * z + LOG2PI) - orc_log(par[1]);
    return fam == ORC_PRIOR_TRUNCNORMAL ? l - par[4] : l;
}
/* theta' = rand(MvNormal(theta, scale * L L'))  smc_samplers.jl:99-100,114: theta'_i = theta_i + sqrt(scale) sum_{k<=i} L_ik z_k */
void orc_pmmh_propose(int d, uint64_t seed, uint32_t stream, uint32_t c, const double* theta, const double* L, double scale, double* prop) {
    double z[10];
    for (int k = 0; k < d; k += 2) normal_pair(seed, k >> 1, stream, c, SLOT_PMMH_Z, z + k);
    const double sq = sqrt(scale);
    for (int i = 0; i < d; ++i) {
        double a = 0.0;
        for (int k = 0; k <= i; ++k) a = a + L[i * d + k] * z[k];
        prop[i] = theta[i] + sq * a;
    }
}
/* log(rand()) of the accept test  smc_samplers.jl:129 */
double orc_pmmh_log_uniform(uint64_t seed, uint32_t stream, uint32_t c) {
    uint32_t wd[4];
    draw(seed, 0, stream, c, SLOT_PMMH_U, wd);
    return orc_log((double)((((((uint64_t)wd[1] << 32) | wd[0]) >> 11)) + 1) * TWO_M53);
}

/* Weighted quantiles of state coordinate `comp` under the current weights: what
 * quantile(smc.x[i], weights(smc.w[i]), p) delivers per theta-particle in examples/inflation_example.jl:45
 * (StatsBase.jl, not vendored).  Definition used here (integer, order-free): with W_i = q_i >> sh_b the
 * weight of particle i in the units of the combined segment table and Wtot = sum W_i,
 *     quantile(p) = the smallest particle value v with  sum{W_i : x_i <= v} > floor(p * Wtot)
 * i.e. the inverse of the weighted empirical CDF (no interpolation between particles; values are
 * ordered by the IEEE total order of their bits, so -0.0 < +0.0).  NaN if every weight is 0.
 * This oracle SORTS; the device does a radix select - two different algorithms for the same definition. */
typedef struct { uint64_t key, w; } orc_kw;
static int kw_cmp(const void* a, const void* b) {
    uint64_t x = ((const orc_kw*)a)->key, y = ((const orc_kw*)b)->key;
    return x < y ? -1 : x > y;
}
static uint64_t order_key(double v) { uint64_t b = d2bits(v); return (b >> 63) ? ~b : (b | 0x8000000000000000ULL); }
static double key_value(uint64_t k) { return bits2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k); }
uint64_t orc_prob_to_u64(double p) {   /* floor(p * 2^64), clamped to [0, 2^64-1] */
    if (!(p > 0.0)) return 0;
    if (p >= 1.0) return ~(uint64_t)0;
    return (uint64_t)(p * 0x1p64);
}
int orc_filter_quantiles(const orc_filter* f, int comp, const double* p, int np, double* out) {
    const orc_weights* W = &f->W;
    if (comp < 0 || comp >= f->model.d || np < 1) return -1;
    const int64_t n = f->n;
    orc_kw* kw = (orc_kw*)malloc(sizeof(orc_kw) * (size_t)n);
    uint64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int b = (int)(i / W->seg), j = (int)(i % W->seg);
        const double dk = W->K - W->kb[b];
        int sh = (dk >= 0.0 && dk < 64.0) ? (int)dk + W->SH : 64;
        if (sh > 64) sh = 64;
        const uint64_t q = W->C[i] - (j ? W->C[i - 1] : 0);
        kw[i].key = order_key(f->x[(size_t)comp * n + i]);
        kw[i].w = sh < 64 ? q >> sh : 0;
        tot += kw[i].w;
    }
    qsort(kw, (size_t)n, sizeof(orc_kw), kw_cmp);
    for (int k = 0; k < np; ++k) {
        if (!tot) { out[k] = bits2d(0x7ff8000000000000ULL); continue; }
        const uint64_t T = (uint64_t)(((u128)orc_prob_to_u64(p[k]) * tot) >> 64);
        uint64_t run = 0;
        int64_t i = 0;
        for (; i < n; ++i) { run += kw[i].w; if (run > T) break; }
        out[k] = key_value(kw[i < n ? i : n - 1].key);
    }
    free(kw);
    return 0;
}

/* weighted mean and variance of every state coordinate under the current weights (plain sums) */
void orc_filter_moments(const orc_filter* f, double* mean, double* var) {
    const int64_t n = f->n;
    double* w = (double*)malloc(8 * (size_t)n);
    weights_dense(&f->W, w);
    for (int c = 0; c < f->model.d; ++c) {
        double m = 0.0, m2 = 0.0;
        for (int64_t i = 0; i < n; ++i) { const double x = f->x[(size_t)c * n + i]; m += w[i] * x; m2 += w[i] * x * x; }
        mean[c] = m; var[c] = m2 - m * m;
    }
    free(w);
}

/* ------------------------------------------------------------------------------------ */
/* The OUTER level of the samplers (src/smc_samplers.jl): reweight, the window walk of     */
/* smc²!, the tempering bisection, resample!(smc), the random-walk factor.                 */
/* `reweight` is undefined in the reference's tree; it is normalize (particles.jl:5-15).   */
/* Specification: the filter's own segmented fixed-point normalize   */
/* with segments of 8 consecutive entries - every sum an integer sum, so the result does   */
/* not depend on how the entries are dealt out to ranks.  These are the straightforward    */
/* whole-vector versions: the product computes the same numbers rank by rank.              */
/* ------------------------------------------------------------------------------------ */
#define ORC_OUTER_SEG 8
#define OUTER_STREAM 0xFFFFFFFEu   /* Philox stream id of the outer level */
#define SLOT_OUTER 34u             /* pick numbers of resample!(smc) */

/* reweight(logw) -> (logmu, w, ess)   smc_samplers.jl:232,249,265,298,338 */
int orc_outer_reweight(const double* logw, int64_t n, double* w, double* logmu, double* ess) {
    if (n <= 0) return -1;
    orc_weights W;
    weights_alloc(&W, n, ORC_OUTER_SEG);
    weights_normalize(&W, logw);
    if (w) weights_dense(&W, w);
    if (logmu) *logmu = W.logmu;
    if (ess) *ess = W.ess;
    weights_free(&W);
    return 0;
}

/* the host half of k consecutive smc²! steps (smc_samplers.jl:323-338) with un-normalised outer log-weights:
 * logw .+= lik_t; logZ .+= lik_t; ess_t = reweight(logw).ess - stopping after the first step with ess_t < ess_min.
 * logw and logZ are advanced in place by the steps walked; returns their number. */
int orc_outer_steps(double* logw, double* logZ, const double* lik /*[k][n]*/, int k, int64_t n, double ess_min, double* ess_out) {
    int j = 0;
    while (j < k) {
        const double* l = lik + (size_t)j * (size_t)n;
        for (int64_t i = 0; i < n; ++i) { logw[i] = logw[i] + l[i]; logZ[i] = logZ[i] + l[i]; }
        double e = 0.0;
        orc_outer_reweight(logw, n, NULL, NULL, &e);
        ess_out[j++] = e;
        if (e < ess_min) break;
    }
    return j;
}

/* the bisection of density_tempered for the next exponent   smc_samplers.jl:240-266 */
int orc_outer_temper(const double* logZ, int64_t n, double xi, double ess_min, double* xi_new, double* ess_new, double* logw_out) {
    double* lw = (double*)malloc(sizeof(double) * (size_t)n);
    double lower = xi, upper = 2.0, newxi = xi, e = 0.0;
    int resample_flag = 1;
    while (upper - lower > 1.e-6) {                                   /* :245 */
        newxi = (upper + lower) / 2.0;                                /* :246 */
        for (int64_t i = 0; i < n; ++i) lw[i] = (newxi - xi) * logZ[i];
        orc_outer_reweight(lw, n, NULL, NULL, &e);                    /* :249 */
        if (e == ess_min) break;                                      /* :251 */
        else if (e < ess_min) upper = newxi;                          /* :253 */
        else lower = newxi;                                           /* :255 */
    }
    if (newxi >= 1.0) {                                               /* :261 corner solution */
        resample_flag = 0;
        newxi = 1.0;
        for (int64_t i = 0; i < n; ++i) lw[i] = (newxi - xi) * logZ[i];
        orc_outer_reweight(lw, n, NULL, NULL, &e);                    /* :265 */
    }
    for (int64_t i = 0; i < n; ++i) lw[i] = (newxi - xi) * logZ[i];
    if (logw_out) memcpy(logw_out, lw, sizeof(double) * (size_t)n);
    free(lw);
    *xi_new = newxi;
    *ess_new = e;
    return resample_flag;
}

static int u64_cmp(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
/* a = resample(omega) of resample!(smc)   smc_samplers.jl:74-84: m iid draws from the weights exp(logw), the ancestors in
 * ascending order (0-based).  The m targets are drawn, SORTED, and looked up one after the other. */
int orc_outer_resample(const double* logw, int64_t n, int64_t m, uint64_t seed, int64_t* a) {
    if (n <= 0 || m < 0) return -1;
    orc_weights W;
    weights_alloc(&W, n, ORC_OUTER_SEG);
    weights_normalize(&W, logw);
    if (W.Dtot == 0) {
        for (int64_t j = 0; j < m; ++j) a[j] = j < n ? j : n - 1;
        weights_free(&W);
        return 0;
    }
    uint64_t* T = (uint64_t*)malloc(8 * (size_t)(m > 0 ? m : 1));
    uint64_t r[2];
    for (int64_t j = 0; j < m; ++j) {
        if (!(j & 1)) resample_pair(seed, j >> 1, OUTER_STREAM, 0, SLOT_OUTER, r);
        T[j] = (uint64_t)(((u128)r[j & 1] * W.Dtot) >> 64);
    }
    qsort(T, (size_t)m, 8, u64_cmp);
    for (int64_t j = 0; j < m; ++j) {
        const int b = (int)upper_bound_u64(W.Dcum, W.nseg, T[j]);
        const uint64_t T2 = T[j] - (b ? W.Dcum[b - 1] : 0);
        const double dk = W.K - W.kb[b];
        int sh = (dk >= 0.0 && dk < 64.0) ? (int)dk + W.SH : 64;
        if (sh > 64) sh = 64;
        const uint64_t thr = sh < 64 ? ((T2 + 1) << sh) - 1 : 0;
        int64_t i = (int64_t)b * W.seg + upper_bound_u64(W.C + (size_t)b * W.seg, W.seg, thr);
        a[j] = i < n ? i : n - 1;
    }
    free(T);
    weights_free(&W);
    return 0;
}

/* random_walk_kernel(theta)   smc_samplers.jl:87-101: lower Cholesky factor L [d][d] of the proposal covariance from the
 * cloud theta [n][d]; returns 1 for univariate theta (L = [[sigma]], handed to Normal() as a standard deviation, :87-92) */
int orc_rw_factor(const double* theta, int64_t n, int d, double* L) {
    double mean[8], cov[8][8], Sg[8][8];
    for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int64_t m = 0; m < n; ++m) s = s + theta[m * d + i];
        mean[i] = s / (double)n;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int64_t m = 0; m < n; ++m) s = s + (theta[m * d + i] - mean[i]) * (theta[m * d + j] - mean[j]);
            cov[i][j] = s / (double)(n - 1);
            cov[j][i] = cov[i][j];
        }
    double nrm2 = 0.0;
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) nrm2 = nrm2 + cov[i][j] * cov[i][j];
    const int collapsed = sqrt(nrm2) < 1.e-8;                                   /* norm(cov(theta)) < 1.e-8 */
    const double dth = 2.83 * 2.83;
    if (d == 1) {
        L[0] = collapsed ? 1.e-2 : dth * cov[0][0] + 1.e-10;                     /* :89 */
        return 1;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
            Sg[i][j] = collapsed ? (i == j ? 1.e-2 : 0.0) : (dth / (double)d) * cov[i][j] + (i == j ? 1.e-10 : 0.0);   /* :97-98 */
    /* Cholesky-Banachiewicz, row by row */
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) L[i * d + j] = 0.0;
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = Sg[i][j];
            for (int k = 0; k < j; ++k) s = s - L[i * d + k] * L[j * d + k];
            L[i * d + j] = (i == j) ? sqrt(s) : s / L[j * d + j];
        }
    }
    return 0;
}
