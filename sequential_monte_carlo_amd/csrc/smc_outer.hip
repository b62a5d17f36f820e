// smc_outer.hip -- the OUTER level of the samplers (src/smc_samplers.jl) in ONE numerical specification: integer, order-free,
// shardable.  Host code (no kernel here): O(n_theta) work per step, vectorised for the host's AVX2 + FMA units.
//
//   reweight(logw) -> (logmu, w, ess)   smc_samplers.jl:232,249,265,298,338 (undefined in the reference's tree; == normalize,
//                                       particles.jl:5-15)
//   the window walk of smc²!            :323-338   logw .+= lik_j ; ess_j = reweight(logw).ess, until ess_j < ess_min
//   the tempering bisection             :240-266
//   resample!(smc) index draw           :74-84     sample(1:M, Weights(w), M), ancestors in ascending order
//   random_walk_kernel covariance       :87-101    cov of the theta cloud, its Cholesky factor
//
// Specification ("outer level"): reweight IS the inner filter's normalize with segments of OSEG = 8
// consecutive entries: exp(logw_i) = p_i 2^k_i; segment b carries kb = max k_i and the 48-bit fixed-point weights
// q_i = rint(p_i 2^(48 + k_i - kb)), S_b = sum q_i, S2_b = sum q_i^2 (128 bit) - its RECORD (kb, S_b, S2_b); the records are
// combined by shifts against K = max kb exactly as smc_spec.h "segment combine" does (SH = table_shift_extra(nseg * OSEG)).
// Every sum is an integer sum: the result does not depend on the order of the entries' evaluation, nor on how the segments are
// dealt out to ranks - a rank that holds whole segments computes their records alone, the ranks exchange records (32 bytes per 8
// parameter particles and step) and every rank combines them to the same (logmu, ess).  The online sampler carries the
// UN-NORMALISED log-weights logw (the reference re-normalises at every step, :338, and takes log.(omega) again, :324: the same
// weights up to a common factor, which reweight removes anyway), so that a step is one addition per parameter particle.
#include "../../include/smc_hip.h"
#include "smc_spec.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int smc_set_error_(int code, const char* msg);   // smc_capi.hip

using namespace smc;

namespace {

constexpr int OSEG = SMC_OUTER_SEG;
constexpr int DEADK = -(1 << 30);
constexpr uint32_t OUTER_STREAM = 0xFFFFFFFEu;   // Philox stream id of the outer level (theta particles use 0 .. M-1, simulate() 0xFFFFFFFF)
constexpr uint32_t SLOT_OUTER = 34u;             // the pick numbers of resample!(smc)

struct ORec {
    double kb;        // integral, or -inf for a segment without a live entry
    uint64_t S, hi, lo;
};
static_assert(sizeof(ORec) == 32, "record layout of the C ABI");

#if defined(__x86_64__)
#define SMC_HOSTVEC __attribute__((target("avx2,fma")))
bool vec_ok() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && !getenv("SMC_HOST_SCALAR");
    return ok;
}
#else
#define SMC_HOSTVEC
bool vec_ok() { return false; }
#endif

// exp(l) = p 2^k for every entry: p [n] (0 for a dead entry), k [n] (DEADK for a dead entry).  The arithmetic is sp_exp_k /
// sp_exp_p element by element; written branch-free, with the double -> int conversion done on the bits, so that the loop
// compiles to 4-wide vector code under the AVX2 + FMA target (the scalar build of the same source gives the same bits).
#define SMC_EXP_PARTS_BODY                                                                                    \
    for (int64_t i = 0; i < n; ++i) {                                                                         \
        const double l = lw[i];                                                                               \
        const bool alive = l == l && (l < 0.0 ? -l : l) <= 7e8;                                               \
        const double ls = alive ? l : 0.0;                                                                    \
        const double kq = (ls * INV_LN2 + 0x1.8p52) - 0x1.8p52;                                               \
        double r = fma(-kq, LN2_HI, ls);                                                                      \
        r = fma(-kq, LN2_LO, r);                                                                              \
        double q = 0x1.6124613a86d09p-33;                                                                     \
        q = fma(q, r, 0x1.1eed8eff8d898p-29);                                                                 \
        q = fma(q, r, 0x1.ae64567f544e4p-26);                                                                 \
        q = fma(q, r, 0x1.27e4fb7789f5cp-22);                                                                 \
        q = fma(q, r, 0x1.71de3a556c734p-19);                                                                 \
        q = fma(q, r, 0x1.a01a01a01a01ap-16);                                                                 \
        q = fma(q, r, 0x1.a01a01a01a01ap-13);                                                                 \
        q = fma(q, r, 0x1.6c16c16c16c17p-10);                                                                 \
        q = fma(q, r, 0x1.1111111111111p-7);                                                                  \
        q = fma(q, r, 0x1.5555555555555p-5);                                                                  \
        q = fma(q, r, 0x1.5555555555555p-3);                                                                  \
        q = fma(q, r, 0.5);                                                                                   \
        q = fma(q, r, 1.0);                                                                                   \
        q = fma(q, r, 1.0);                                                                                   \
        p[i] = alive ? q : 0.0;                                                                               \
        /* (int)kq, |kq| < 2^31: the low 32 bits of the mantissa of kq + 1.5 2^52 */                          \
        const int32_t ki = (int32_t)(uint32_t)d2bits(kq + 0x1.8p52);                                          \
        k[i] = alive ? ki : DEADK;                                                                            \
    }
SMC_HOSTVEC void exp_parts_vec(const double* lw, int64_t n, double* p, int32_t* k) { SMC_EXP_PARTS_BODY }
void exp_parts_scalar(const double* lw, int64_t n, double* p, int32_t* k) { SMC_EXP_PARTS_BODY }
void exp_parts(const double* lw, int64_t n, double* p, int32_t* k) {
    if (vec_ok()) exp_parts_vec(lw, n, p, k);
    else exp_parts_scalar(lw, n, p, k);
}

// q_i of one segment (cnt <= OSEG entries) and its record.  Branch-free over a fixed width of OSEG (short segments are padded
// with dead entries), so that the inner loops become vector code: the exponent difference is clamped at -(48 + 8), where
// p 2^-8 < 1/2 rounds to the 0 that fix_weight_i returns below -(48 + 2) (as in segment_normalize of smc_kernels.h); the sum of
// squares is taken over the 25-bit halves of q (q < 2^49: three 64-bit sums of 32 x 32 -> 64 products, combined once).
#define SMC_SEGREC_BODY                                                                                        \
    double pp[OSEG];                                                                                           \
    int32_t kk[OSEG];                                                                                          \
    for (int i = 0; i < OSEG; ++i) { pp[i] = i < cnt ? p[i] : 0.0; kk[i] = i < cnt ? k[i] : DEADK; }             \
    int kbi = DEADK;                                                                                           \
    for (int i = 0; i < OSEG; ++i) kbi = kk[i] > kbi ? kk[i] : kbi;                                            \
    uint64_t q[OSEG];                                                                                          \
    for (int i = 0; i < OSEG; ++i) {                                                                           \
        const int dk = kk[i] - kbi;                                                                            \
        const int ek = (dk > -(FIX_BITS + 8) ? dk : -(FIX_BITS + 8)) + FIX_BITS;                               \
        const double sc = bits2d((uint64_t)(uint32_t)(ek + 1023) << 52);                                       \
        q[i] = d2bits(pp[i] * sc + 0x1p52) & 0x000fffffffffffffULL;                                            \
    }                                                                                                          \
    uint64_t s = 0, hh = 0, hl = 0, ll = 0;                                                                    \
    for (int i = 0; i < OSEG; ++i) {                                                                           \
        const uint64_t lo = q[i] & 0x1ffffffULL, hi = q[i] >> 25;                                              \
        s += q[i]; hh += hi * hi; hl += hi * lo; ll += lo * lo;                                                \
    }                                                                                                          \
    if (qout) for (int i = 0; i < cnt; ++i) qout[i] = kbi == DEADK ? 0 : q[i];                                 \
    ORec r{-inf(), 0, 0, 0};                                                                                   \
    if (kbi == DEADK) return r;                                                                                \
    const unsigned __int128 s2 = ((unsigned __int128)hh << 50) + ((unsigned __int128)hl << 26) + ll;           \
    r.kb = (double)kbi; r.S = s; r.hi = (uint64_t)(s2 >> 64); r.lo = (uint64_t)s2;                             \
    return r;
SMC_HOSTVEC inline ORec segment_record_vec(const double* p, const int32_t* k, int cnt, uint64_t* qout) { SMC_SEGREC_BODY }
inline ORec segment_record_scalar(const double* p, const int32_t* k, int cnt, uint64_t* qout) { SMC_SEGREC_BODY }

// records of the nseg = ceil(n / OSEG) segments of logw; qall [n] (optional): every entry's fixed-point weight
#define SMC_RECORDS_BODY(SEGREC)                                                                               \
    const int64_t nseg = (n + OSEG - 1) / OSEG;                                                                \
    for (int64_t b = 0; b < nseg; ++b) {                                                                       \
        const int64_t i0 = b * OSEG;                                                                           \
        const int cnt = (int)(n - i0 < OSEG ? n - i0 : OSEG);                                                  \
        rec[b] = SEGREC(p + i0, k + i0, cnt, qall ? qall + i0 : nullptr);                                      \
    }
SMC_HOSTVEC void records_loop_vec(const double* p, const int32_t* k, int64_t n, ORec* rec, uint64_t* qall) { SMC_RECORDS_BODY(segment_record_vec) }
void records_loop_scalar(const double* p, const int32_t* k, int64_t n, ORec* rec, uint64_t* qall) { SMC_RECORDS_BODY(segment_record_scalar) }
void records_of(const double* lw, int64_t n, ORec* rec, std::vector<double>& p, std::vector<int32_t>& k, uint64_t* qall = nullptr) {
    p.resize((size_t)n);
    k.resize((size_t)n);
    exp_parts(lw, n, p.data(), k.data());
    if (vec_ok()) records_loop_vec(p.data(), k.data(), n, rec, qall);
    else records_loop_scalar(p.data(), k.data(), n, rec, qall);
}

// The pick numbers of resample!(smc): r_j, j < m, = the 64-bit halves of draw(seed, j / 2, OUTER_STREAM, 0, SLOT_OUTER) -
// Philox4x32-10 for 8 counters at a time on hosts with AVX2 (the same integer arithmetic lane by lane: the same bits).
void pick_numbers_scalar(uint64_t seed, int64_t m, uint64_t* r) {
    for (int64_t j = 0; j < m; j += 2) {
        const u32x4 w = draw(seed, (uint32_t)(j >> 1), OUTER_STREAM, 0u, SLOT_OUTER);
        r[j] = ((uint64_t)w.v[1] << 32) | w.v[0];
        if (j + 1 < m) r[j + 1] = ((uint64_t)w.v[3] << 32) | w.v[2];
    }
}
#if defined(__x86_64__)
}  // namespace
#include <immintrin.h>
namespace {
SMC_HOSTVEC inline void mulhilo8(__m256i a, uint32_t mult, __m256i& hi, __m256i& lo) {
    const __m256i M = _mm256_set1_epi32((int)mult);
    const __m256i even = _mm256_mul_epu32(a, M);                          // lanes 0,2,4,6: 64-bit products
    const __m256i odd = _mm256_mul_epu32(_mm256_srli_epi64(a, 32), M);    // lanes 1,3,5,7
    lo = _mm256_blend_epi32(even, _mm256_slli_epi64(odd, 32), 0xAA);
    hi = _mm256_blend_epi32(_mm256_srli_epi64(even, 32), odd, 0xAA);
}
SMC_HOSTVEC void pick_numbers_vec(uint64_t seed, int64_t m, uint64_t* r) {
    const int64_t npair = (m + 1) / 2;
    alignas(32) uint32_t o0[8], o1[8], o2[8], o3[8];
    for (int64_t p0 = 0; p0 < npair; p0 += 8) {
        __m256i c0 = _mm256_add_epi32(_mm256_set1_epi32((int)(uint32_t)p0), _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7));
        __m256i c1 = _mm256_set1_epi32((int)OUTER_STREAM), c2 = _mm256_setzero_si256(), c3 = _mm256_set1_epi32((int)SLOT_OUTER);
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        for (int round = 0; round < 10; ++round) {
            __m256i h0, l0, h1, l1;
            mulhilo8(c0, 0xD2511F53u, h0, l0);
            mulhilo8(c2, 0xCD9E8D57u, h1, l1);
            const __m256i n0 = _mm256_xor_si256(_mm256_xor_si256(h1, c1), _mm256_set1_epi32((int)k0));
            const __m256i n2 = _mm256_xor_si256(_mm256_xor_si256(h0, c3), _mm256_set1_epi32((int)k1));
            c0 = n0; c1 = l1; c2 = n2; c3 = l0;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        _mm256_store_si256((__m256i*)o0, c0); _mm256_store_si256((__m256i*)o1, c1);
        _mm256_store_si256((__m256i*)o2, c2); _mm256_store_si256((__m256i*)o3, c3);
        for (int l = 0; l < 8 && p0 + l < npair; ++l) {
            const int64_t j = 2 * (p0 + l);
            r[j] = ((uint64_t)o1[l] << 32) | o0[l];
            if (j + 1 < m) r[j + 1] = ((uint64_t)o3[l] << 32) | o2[l];
        }
    }
}
#endif
void pick_numbers(uint64_t seed, int64_t m, uint64_t* r) {
#if defined(__x86_64__)
    if (vec_ok()) { pick_numbers_vec(seed, m, r); return; }
#endif
    pick_numbers_scalar(seed, m, r);
}

struct Combined {
    double K, logmu, ess;
    uint64_t Dtot, Rtot;
    int SH;
};
// the segment combine of smc_spec.h over all records of a vector of n_total entries
Combined combine(const ORec* rec, int64_t nseg, int64_t n_total) {
    Combined c{};
    c.SH = table_shift_extra(nseg * OSEG);
    double K = -inf();
    for (int64_t b = 0; b < nseg; ++b) K = rec[b].kb > K ? rec[b].kb : K;
    uint64_t D = 0, R = 0;
    for (int64_t b = 0; b < nseg; ++b) {
        const int sh = seg_shift(K, rec[b].kb, c.SH);
        D += seg_Q(rec[b].S, sh);
        R += seg_R(rec[b].hi, rec[b].lo, sh, c.SH);
    }
    c.K = K; c.Dtot = D; c.Rtot = R;
    combine_outputs(K, D, R, c.SH, n_total, c.logmu, c.ess);
    return c;
}

int fail(const std::string& m) { return smc_set_error_(SMC_EINVAL, m.c_str()); }

// normalised weights w_i = q_i 2^(-48 - (K - kb)) / (Dtot 2^(SH - 48)), the inner filter's dense weights (k_dense_weights)
void dense_weights(const uint64_t* q, int64_t n, const ORec* rec, const Combined& c, double* w) {
    const double Dd = (double)c.Dtot * pow2i(c.SH - 48);
    const int64_t nseg = (n + OSEG - 1) / OSEG;
    for (int64_t b = 0; b < nseg; ++b) {
        const int64_t i0 = b * OSEG;
        const int cnt = (int)(n - i0 < OSEG ? n - i0 : OSEG);
        const double dk = c.K - rec[b].kb;
        const double sc = (dk >= 0.0 && dk < 900.0) ? pow2i(-48 - (int)dk) : 0.0;
        for (int i = 0; i < cnt; ++i) w[i0 + i] = c.Dtot ? ((double)q[i0 + i] * sc) / Dd : 0.0;
    }
}

}  // namespace

extern "C" int smc_outer_seg(void) { return OSEG; }

// reweight(logw) -> (logmu, w, ess)
extern "C" int smc_host_reweight(const double* logw, int64_t n, double* w, double* logmu, double* ess) {
    if (!logw || n <= 0) return fail("smc_host_reweight: bad argument");
    const int64_t nseg = (n + OSEG - 1) / OSEG;
    std::vector<ORec> rec((size_t)nseg);
    std::vector<double> p;
    std::vector<int32_t> k;
    std::vector<uint64_t> q;
    if (w) q.resize((size_t)n);
    records_of(logw, n, rec.data(), p, k, w ? q.data() : nullptr);
    const Combined c = combine(rec.data(), nseg, n);
    if (w) dense_weights(q.data(), n, rec.data(), c, w);
    if (logmu) *logmu = c.logmu;
    if (ess) *ess = c.ess;
    return SMC_OK;
}

// the records of the whole segments a rank holds: n_local entries starting at a multiple of OSEG (the last segment of the
// whole vector may be short).  rec [ceil(n_local / OSEG)][4] as 8-byte words (bits of kb, S, S2hi, S2lo).
extern "C" int smc_host_outer_records(const double* logw_local, int64_t n_local, uint64_t* rec) {
    if (!logw_local || !rec || n_local <= 0) return fail("smc_host_outer_records: bad argument");
    std::vector<double> p;
    std::vector<int32_t> k;
    records_of(logw_local, n_local, reinterpret_cast<ORec*>(rec), p, k);
    return SMC_OK;
}

// (logmu, ess) of a vector of n_total entries from the records of ALL its segments, in segment order
extern "C" int smc_host_outer_combine(const uint64_t* rec, int64_t nseg, int64_t n_total, double* logmu, double* ess) {
    if (!rec || nseg <= 0 || n_total <= 0 || nseg != (n_total + OSEG - 1) / OSEG) return fail("smc_host_outer_combine: bad argument");
    const Combined c = combine(reinterpret_cast<const ORec*>(rec), nseg, n_total);
    if (logmu) *logmu = c.logmu;
    if (ess) *ess = c.ess;
    return SMC_OK;
}

// The window walk of the online sampler (smc_samplers.jl:323-338) for the entries a rank holds: the records of
// logw + lik_1, logw + lik_1 + lik_2, ... (additions in step order), rec [k][nseg_local][4].  Nothing is modified.
extern "C" int smc_host_outer_window(const double* logw_local, const double* lik /*[k][n_local]*/, int k, int64_t n_local, uint64_t* rec) {
    if (!logw_local || !lik || !rec || k < 1 || n_local <= 0) return fail("smc_host_outer_window: bad argument");
    std::vector<double> acc(logw_local, logw_local + n_local), p;
    std::vector<int32_t> kk;
    const int64_t nseg = (n_local + OSEG - 1) / OSEG;
    for (int j = 0; j < k; ++j) {
        const double* l = lik + (size_t)j * (size_t)n_local;
        for (int64_t i = 0; i < n_local; ++i) acc[(size_t)i] = acc[(size_t)i] + l[i];
        records_of(acc.data(), n_local, reinterpret_cast<ORec*>(rec) + (size_t)j * (size_t)nseg, p, kk);
    }
    return SMC_OK;
}

// ess of every step of a window from the records of ALL segments, rec [k][nseg][4]; the walk stops after the first step whose
// ESS falls below ess_min (:312 of the next call): *j_out = steps walked, ess_out [k] (the first *j_out are set)
extern "C" int smc_host_outer_walk(const uint64_t* rec, int k, int64_t nseg, int64_t n_total, double ess_min, double* ess_out, int* j_out) {
    if (!rec || !ess_out || !j_out || k < 1 || nseg <= 0 || nseg != (n_total + OSEG - 1) / OSEG) return fail("smc_host_outer_walk: bad argument");
    int j = 0;
    while (j < k) {
        const Combined c = combine(reinterpret_cast<const ORec*>(rec) + (size_t)j * (size_t)nseg, nseg, n_total);
        ess_out[j++] = c.ess;
        if (c.ess < ess_min) break;
    }
    *j_out = j;
    return SMC_OK;
}

// keep the first j steps of a window: logw .+= lik_t, logZ .+= lik_t for t = 1..j, in step order (:333-334)
extern "C" int smc_host_outer_advance(double* logw, double* logZ, const double* lik /*[k][n]*/, int j, int64_t n) {
    if (!logw || !logZ || !lik || j < 0 || n <= 0) return fail("smc_host_outer_advance: bad argument");
    for (int t = 0; t < j; ++t) {
        const double* l = lik + (size_t)t * (size_t)n;
        for (int64_t i = 0; i < n; ++i) { logw[i] = logw[i] + l[i]; logZ[i] = logZ[i] + l[i]; }
    }
    return SMC_OK;
}

// The bisection for the next tempering exponent (smc_samplers.jl:240-266), statement by statement: on return *xi_new is the
// exponent, *ess the ESS of reweight((xi_new - xi) * logZ), *resample_flag = 0 at the corner solution xi_new = 1 (:261-266),
// and logw_out [n] (optional) = (xi_new - xi) * logZ, the log-weights of that reweight.
extern "C" int smc_host_outer_temper(const double* logZ, int64_t n, double xi, double ess_min, double* xi_new, double* ess, int* resample_flag,
                                     double* logw_out) {
    if (!logZ || !xi_new || !ess || !resample_flag || n <= 0) return fail("smc_host_outer_temper: bad argument");
    const int64_t nseg = (n + OSEG - 1) / OSEG;
    std::vector<ORec> rec((size_t)nseg);
    std::vector<double> lw((size_t)n), p;
    std::vector<int32_t> k;
    auto ess_at = [&](double nx) {
        const double dx = nx - xi;
        for (int64_t i = 0; i < n; ++i) lw[(size_t)i] = dx * logZ[i];
        records_of(lw.data(), n, rec.data(), p, k);
        return combine(rec.data(), nseg, n).ess;
    };
    int flag = 1;
    double lower = xi, upper = 2.0, nx = xi, e = 0.0;
    while (upper - lower > 1e-6) {
        nx = (upper + lower) / 2.0;
        e = ess_at(nx);
        if (e == ess_min) break;
        else if (e < ess_min) upper = nx;
        else lower = nx;
    }
    if (nx >= 1.0) {
        flag = 0;
        nx = 1.0;
        e = ess_at(nx);
    }
    *xi_new = nx;
    *ess = e;
    *resample_flag = flag;
    if (logw_out) {
        const double dx = nx - xi;
        for (int64_t i = 0; i < n; ++i) logw_out[i] = dx * logZ[i];
    }
    return SMC_OK;
}

// The index draw of resample!(smc) (smc_samplers.jl:74-84: sample(1:n, Weights(w), m)): m iid draws from the weights
// exp(logw) through the inverse of their integer CDF (the inner filter's two-level table: segment sums in units of 2^K, then the
// fixed-point prefix sums of the segment), pick numbers = 64-bit Philox draws keyed by (seed, pair j/2, OUTER_STREAM, 0,
// SLOT_OUTER).  The ancestors come out in ASCENDING order (the order of a resampled population carries no information; taken
// ascending, slot m inherits from an ancestor close to m, so with theta sharded most filter copies stay on their rank).
// All weights zero: the identity.
extern "C" int smc_host_outer_resample(const double* logw, int64_t n, int64_t m, uint64_t seed, int32_t* a) {
    if (!logw || !a || n <= 0 || m < 0 || n > ((int64_t)1 << 30)) return fail("smc_host_outer_resample: bad argument");
    const int64_t nseg = (n + OSEG - 1) / OSEG;
    std::vector<ORec> rec((size_t)nseg);
    std::vector<double> p;
    std::vector<int32_t> k;
    std::vector<uint64_t> qall((size_t)n);
    records_of(logw, n, rec.data(), p, k, qall.data());
    const Combined c = combine(rec.data(), nseg, n);
    if (c.Dtot == 0) {
        for (int64_t j = 0; j < m; ++j) a[j] = (int32_t)(j < n ? j : n - 1);
        return SMC_OK;
    }
    std::vector<uint64_t> Dcum((size_t)nseg);
    std::vector<int> shs((size_t)nseg);
    uint64_t D = 0;
    for (int64_t b = 0; b < nseg; ++b) {
        shs[(size_t)b] = seg_shift(c.K, rec[(size_t)b].kb, c.SH);
        D += seg_Q(rec[(size_t)b].S, shs[(size_t)b]);
        Dcum[(size_t)b] = D;
    }
    std::vector<int32_t> cnt((size_t)n, 0);
    std::vector<uint64_t> picks((size_t)(m > 0 ? m : 1));
    pick_numbers(seed, m, picks.data());
    // first segment with Dcum > T: a guide table over equal slices of [0, Dtot) gives the segment the search starts at (a scan
    // of one or two entries then: a binary search over the segment table mispredicts at every level)
    int64_t G = 1;
    while (G < 2 * nseg) G <<= 1;
    int lg = 0;
    while ((c.Dtot >> lg) >= (uint64_t)G) ++lg;
    std::vector<int32_t> guide((size_t)G);
    {
        int64_t b = 0;
        for (int64_t g = 0; g < G; ++g) {
            const uint64_t lo = (uint64_t)g << lg;
            while (b < nseg - 1 && Dcum[(size_t)b] <= lo) ++b;
            guide[(size_t)g] = (int32_t)b;
        }
    }
    for (int64_t b = 0; b < nseg; ++b) {                                     // fixed-point weights -> their prefix sums inside the segment
        const int64_t i0 = b * OSEG, i1 = n < i0 + OSEG ? n : i0 + OSEG;
        for (int64_t i = i0 + 1; i < i1; ++i) qall[(size_t)i] += qall[(size_t)i - 1];
    }
    for (int64_t j = 0; j < m; ++j) {
        uint64_t T, lo;
        mul64wide(picks[(size_t)j], c.Dtot, T, lo);
        int64_t b = guide[(size_t)(T >> lg)];
        while (Dcum[(size_t)b] <= T) ++b;                                   // ends: T < Dtot = Dcum[nseg - 1]
        const uint64_t thr = sys_threshold(T - (b ? Dcum[(size_t)b - 1] : 0), shs[(size_t)b]);
        const int64_t i0 = b * OSEG;
        const int cn = (int)(n - i0 < OSEG ? n - i0 : OSEG);
        const uint64_t* C = qall.data() + i0;
        int i = 0;
        for (int t = 0; t < cn; ++t) i += C[t] <= thr ? 1 : 0;               // first entry whose prefix sum exceeds the threshold
        cnt[(size_t)(i0 + (i < cn ? i : cn - 1))] += 1;
    }
    int64_t o = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int32_t r = 0; r < cnt[(size_t)i]; ++r) a[o++] = (int32_t)i;
    return SMC_OK;
}

// random_walk_kernel(theta) (smc_samplers.jl:87-101): the lower Cholesky factor L [d][d] of the proposal covariance of the
// PMMH moves from the theta cloud [n][d], in a fixed order of operations (means and centred products summed over the particles
// in index order, the factorisation column by column):
//   d > 1 (:95-100): Sigma = 2.83^2 / d * cov(theta) + 1e-10 I, or 1e-2 I when norm(cov) < 1e-8 (Frobenius norm);
//   d = 1 (:87-92):  L = [[ 2.83^2 var(theta) + 1e-10 ]] or [[1e-2]] - the reference hands scale * sigma to Normal() as the
//                    STANDARD DEVIATION, so the caller squares its scales (returns *univariate = 1).
extern "C" int smc_host_rw_factor(const double* theta, int64_t n, int d, double* L, int* univariate) {
    if (!theta || !L || n < 2 || d < 1 || d > MAX_DTHETA) return fail("smc_host_rw_factor: bad argument");
    double mean[MAX_DTHETA], cov[MAX_DTHETA][MAX_DTHETA];
    for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int64_t m = 0; m < n; ++m) s = s + theta[(size_t)m * d + i];
        mean[i] = s / (double)n;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int64_t m = 0; m < n; ++m) s = s + (theta[(size_t)m * d + i] - mean[i]) * (theta[(size_t)m * d + j] - mean[j]);
            cov[i][j] = cov[j][i] = s / (double)(n - 1);
        }
    double fro = 0.0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) fro = fro + cov[i][j] * cov[i][j];
    const bool collapsed = sqrt(fro) < 1e-8;
    const double dth = 2.83 * 2.83;
    if (univariate) *univariate = d == 1 ? 1 : 0;
    if (d == 1) {
        L[0] = collapsed ? 1e-2 : dth * cov[0][0] + 1e-10;
        return SMC_OK;
    }
    double S[MAX_DTHETA][MAX_DTHETA];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) S[i][j] = collapsed ? (i == j ? 1e-2 : 0.0) : (dth / (double)d) * cov[i][j] + (i == j ? 1e-10 : 0.0);
    for (int i = 0; i < d * d; ++i) L[i] = 0.0;
    for (int j = 0; j < d; ++j) {
        double s = S[j][j];
        for (int k = 0; k < j; ++k) s = s - L[j * d + k] * L[j * d + k];
        if (!(s > 0.0)) return fail("smc_host_rw_factor: the proposal covariance is not positive definite");
        const double ljj = sqrt(s);
        L[j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            double t = S[i][j];
            for (int k = 0; k < j; ++k) t = t - L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / ljj;
        }
    }
    return SMC_OK;
}
