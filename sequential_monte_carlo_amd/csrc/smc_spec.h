// smc_spec.h -- numerical specification of the particle-filter hot path, host + device.
//
// Everything a particle ever computes is defined here with IEEE-754 binary64 +,-,*,/,sqrt,
// explicit fma() and integer arithmetic only, so that a gfx950 lane and a host core produce
// the same bits (DESIGN.md "Numerical specification").  Compile with -ffp-contract=off.
//
// Reference semantics being implemented (charlesknipp/sequential_monte_carlo @ v1):
//   rand(Normal(mu,sigma)), logpdf(Normal(mu,sigma),y)   src/particles.jl:97-98,123-124
//   model methods                                        src/state_space_models.jl:87-109,233-259
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMC_HD __host__ __device__ __forceinline__

namespace smc {

// ---- model ids (C ABI values) ---------------------------------------------------------
constexpr int MODEL_LG1D = 1;    // UnivariateLinearGaussian  ssm.jl:74-109
constexpr int MODEL_SV1D = 2;    // stochastic volatility (SURVEY A7'; obs template ssm.jl:244-247)
constexpr int MODEL_UCSV3D = 3;  // UCSV                      ssm.jl:215-263
constexpr int NPARAM = 8;        // padded row length of raw / derived parameter tables

constexpr int FIX_BITS = 48;     // q = rint(p * 2^(48 + k - kb)),  exp(logw) = p * 2^k
constexpr int MAX_SEG = 8192;
constexpr uint32_t SIM_STREAM = 0xFFFFFFFFu;
constexpr uint32_t SLOT_RESAMPLE = 0u;   // within-segment pick of child j
constexpr uint32_t SLOT_NORMAL0 = 1u;    // slots 1..d: state normals
constexpr uint32_t SLOT_OBS = 8u;        // simulate(): observation noise
constexpr uint32_t SLOT_COUNT = 9u;      // segment pick of draw i (multi-segment filters)
constexpr uint32_t SLOT_SYS = 10u;       // the one uniform of a systematic resampling step (opt-in)
constexpr uint32_t SLOT_BREAK = 16u;     // block break points: 16+2i normal, 17+2i uniform, i < 8; pair = block
constexpr uint32_t SLOT_PMMH_Z = 32u;    // PMMH proposal normals of a parameter particle: pair k holds z[2k], z[2k+1]
constexpr uint32_t SLOT_PMMH_U = 33u;    // the uniform of its accept test
constexpr int MAX_DTHETA = 8;            // parameter dimension of the samplers (SMC_MAX_DTHETA)

constexpr double HALF_LOG2PI = 0x1.d67f1c864beb5p-1;
constexpr double INV_LN2 = 0x1.71547652b82fep+0;
constexpr double LN2_HI = 0x1.62e42fee00000p-1;
constexpr double LN2_LO = 0x1.a39ef35793c76p-33;
constexpr double SQRT2 = 0x1.6a09e667f3bcdp+0;
constexpr double PIO4 = 0x1.921fb54442d18p-1;
constexpr double TWO_M53 = 0x1p-53;
constexpr double TWO_M48 = 0x1p-48;
constexpr double TWO_P48 = 0x1p+48;
constexpr double TWO_M96 = 0x1p-96;
constexpr double TWO_P64 = 0x1p+64;

template <int MODEL> struct model_dim { static constexpr int value = (MODEL == MODEL_UCSV3D) ? 3 : 1; };

SMC_HD int model_dim_rt(int id) { return id == MODEL_UCSV3D ? 3 : (id == MODEL_LG1D || id == MODEL_SV1D) ? 1 : -1; }
SMC_HD int model_nraw_rt(int id) { return id == MODEL_LG1D ? 6 : id == MODEL_SV1D ? 3 : id == MODEL_UCSV3D ? 5 : -1; }

// ---- bit casts -------------------------------------------------------------------------
SMC_HD double bits2d(uint64_t b) { return __builtin_bit_cast(double, b); }
SMC_HD uint64_t d2bits(double d) { return __builtin_bit_cast(uint64_t, d); }
SMC_HD double pow2i(int k) { return bits2d((uint64_t)(k + 1023) << 52); }  // 2^k, k in [-1022,1023]
// v * 2^k for a result in the normal range: the same bits as v * pow2i(k) (an exact scaling either way); one v_ldexp_f64
// on the device instead of building the power of two and multiplying
SMC_HD double scale2(double v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(v, k);
#else
    return v * pow2i(k);
#endif
}
SMC_HD double inf() { return bits2d(0x7ff0000000000000ULL); }

// round to nearest even: signed |v| < 2^51 ; non-negative of any size
SMC_HD double rne(double v) { return (v + 0x1.8p52) - 0x1.8p52; }
SMC_HD double rne_pos(double v) { return v < 0x1p52 ? (v + 0x1p52) - 0x1p52 : v; }

// ---- Philox4x32-10 ---------------------------------------------------------------------
struct u32x4 { uint32_t v[4]; };

SMC_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// a ^ b ^ c: one v_bitop3_b32 on gfx950 (the compiler does not form it by itself)
SMC_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

// a ^ (b & c): one v_bitop3_b32 as well
SMC_HD uint32_t xor_and(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78);   // 0xF0 ^ (0xCC & 0xAA)
#else
    return a ^ (b & c);
#endif
}

// SMC_EXP_ROUNDS / SMC_EXP_PICKF64: timing experiments of `make exp` only (scripts/dbg/exp_spec.sh: what a cheaper numerical
// specification would buy); the product and the oracle are Philox4x32-10 and 64 x 64 -> 128-bit integer picks
#ifndef SMC_EXP_ROUNDS
#define SMC_EXP_ROUNDS 10
#endif
#ifndef SMC_EXP_PICKF64
#define SMC_EXP_PICKF64 0
#endif
SMC_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < SMC_EXP_ROUNDS; ++r) {
        // full 32x32 -> 64 products (one v_mad_u64_u32 each on the device)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = xor3(h1, c1, k0), n2 = xor3(h0, c3, k1);
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{{c0, c1, c2, c3}};
}

SMC_HD u32x4 draw(uint64_t seed, uint32_t pair, uint32_t stream, uint32_t t, uint32_t slot) {
    return philox4x32_10(pair, stream, t, slot, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// fma(p, r, c) with a compile-time constant c.  On the device the constant sits in a scalar register pair filled by two
// s_mov_b32 (scalar issue port) and the instruction is the three-source v_fma_f64: the compiler's own choice, v_fmac_f64,
// accumulates INTO the constant and so first copies every coefficient into vector registers - two v_mov_b32 per
// coefficient, 29 % of Box-Muller's vector instructions.  Same IEEE fused multiply-add, same bits.
#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t BITS>
__device__ __forceinline__ double fma_const(double p, double r) {
    uint32_t lo, hi;
    double d;
    // SMC_FMAK_PINNED (the translation unit of the persistent step kernel): `volatile` keeps the constants where they are used - a
    // loop around the step would otherwise hoist all ~60 of them into scalar registers it does not have (134 spilled lanes)
#if defined(SMC_FMAK_PINNED)
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((uint32_t)BITS));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((uint32_t)(BITS >> 32)));
#else
    asm("s_mov_b32 %0, %1" : "=s"(lo) : "n"((uint32_t)BITS));
    asm("s_mov_b32 %0, %1" : "=s"(hi) : "n"((uint32_t)(BITS >> 32)));
#endif
    const double c = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
    return d;
}
#define SMC_FMAK(p, r, c) fma_const<__builtin_bit_cast(uint64_t, (double)(c))>((p), (r))
#else
#define SMC_FMAK(p, r, c) fma((p), (r), (c))
#endif

// ---- division and square root without their range handling ---------------------------------
// n / d and sqrt(x) for operands of moderate magnitude: the refinement sequences of the compiler's IEEE expansions without
// the operand scaling (v_div_scale, v_ldexp) and the special-value fix-ups around them - inactive for these operands, so the
// results are the same bits (the host side, and the oracle, use the correctly rounded library operations they reproduce).
#if defined(__HIP_DEVICE_COMPILE__)
// |d| in [2^-500, 2^500], n zero or |n / d| in [2^-500, 2^500]
__device__ __forceinline__ double div_moderate(double n, double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    const double q = n * y;
    const double r = fma(-d, q, n);
    return fma(r, y, q);
}
// x = -0.0 or x in [2^-500, 2^500]   (sqrt(-0.0) = -0.0: the refinement turns it into NaN, which the maximum drops)
__device__ __forceinline__ double sqrt_moderate_or_negzero(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    double d = fma(-g, g, x);
    h = fma(h, r, h);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return __builtin_fmax(g, -0.0);
}
#else
inline double div_moderate(double n, double d) { return n / d; }
inline double sqrt_moderate_or_negzero(double x) { return sqrt(x); }
#endif

// ---- exp / log / sincos ----------------------------------------------------------------
// exp(x) = p * 2^k, k = rint(x / ln2) returned as an integral double, p in [0.707, 1.415]; |x| <= 7e8
// the two halves of sp_exp_parts: k alone (two instructions), and p once k is known
SMC_HD double sp_exp_k(double x) { return rne(x * INV_LN2); }
SMC_HD double sp_exp_p(double x, double k);
SMC_HD double sp_exp_parts(double x, double& kout) {
    const double k = sp_exp_k(x);
    kout = k;
    return sp_exp_p(x, k);
}
SMC_HD double sp_exp_p(double x, double k) {
    double r = fma(-k, LN2_HI, x);
    r = fma(-k, LN2_LO, r);
    double p = 0x1.6124613a86d09p-33;
    p = SMC_FMAK(p, r, 0x1.1eed8eff8d898p-29);
    p = SMC_FMAK(p, r, 0x1.ae64567f544e4p-26);
    p = SMC_FMAK(p, r, 0x1.27e4fb7789f5cp-22);
    p = SMC_FMAK(p, r, 0x1.71de3a556c734p-19);
    p = SMC_FMAK(p, r, 0x1.a01a01a01a01ap-16);
    p = SMC_FMAK(p, r, 0x1.a01a01a01a01ap-13);
    p = SMC_FMAK(p, r, 0x1.6c16c16c16c17p-10);
    p = SMC_FMAK(p, r, 0x1.1111111111111p-7);
    p = SMC_FMAK(p, r, 0x1.5555555555555p-5);
    p = SMC_FMAK(p, r, 0x1.5555555555555p-3);
    p = SMC_FMAK(p, r, 0.5);
    p = SMC_FMAK(p, r, 1.0);
    p = SMC_FMAK(p, r, 1.0);
    return p;
}

SMC_HD double sp_exp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // branch-free on the device (as three nested divergent branches the calls of a thread could not overlap): the body runs on
    // whatever x is - nothing in it traps - and the two range cases are selected afterwards; a NaN passes both comparisons
    // and comes out of the arithmetic as itself
    double k;
    const double p = sp_exp_parts(x, k);
    double r = scale2(p, (int)k);    // x in (-708, 709]: a normal number
    r = x > 709.0 ? inf() : r;
    r = x <= -708.0 ? 0.0 : r;
    return r;
#else
    if (x != x) return x;
    if (!(x > -708.0)) return 0.0;
    if (x > 709.0) return inf();
    double k;
    const double p = sp_exp_parts(x, k);
    return scale2(p, (int)k);        // x > -708: the result is a normal number
#endif
}

// a log-weight takes part in the normalisation iff it is a number of sane magnitude
SMC_HD bool lw_alive(double l) { return l == l && (l < 0.0 ? -l : l) <= 7e8; }   // |k| < 2^30: differences fit int32

// q = rint(p * 2^(bits + dk)),  dk = k_i - kb <= 0 (integral doubles)
SMC_HD uint64_t fix_weight(double p, double dk, int bits) {
    if (dk < -(double)(bits + 2)) return 0;
    return (uint64_t)rne_pos(p * pow2i(bits + (int)dk));
}
// the same value for an integer exponent difference, with the rounding and the conversion done by
// ONE addition: v < 2^50, so v + 2^52 holds rint(v) in its low mantissa bits
SMC_HD uint64_t fix_weight_i(double p, int dk, int bits) {
    if (dk < -(bits + 2)) return 0;
    return d2bits(p * pow2i(bits + dk) + 0x1p52) & 0x000fffffffffffffULL;
}

// log(x) for x positive, finite and normal (no special cases to test): the body of sp_log
SMC_HD double sp_log_normal(double x, int e0) {
    const uint64_t b = d2bits(x);
    // m in (sqrt(1/2), sqrt(2)], x = m 2^e: the mantissa bits decide (m > sqrt 2 as doubles of equal exponent), and the
    // exponent field of m is written directly (0x3fe halves it)
    const uint64_t mant = b & 0x000fffffffffffffULL;
    const bool up = mant > (0x3ff6a09e667f3bcdULL & 0x000fffffffffffffULL);
    const int e = e0 + (int)((b >> 52) & 0x7ff) - 1023 + (up ? 1 : 0);
    const double m = bits2d(mant | (up ? 0x3fe0000000000000ULL : 0x3ff0000000000000ULL));
    const double f = m - 1.0;
    const double s = div_moderate(f, 2.0 + f);   // f in [-0.2929, 0.4143]: zero or of magnitude >= 2^-53
    const double z = s * s;
    double R = 0x1.642c8590b2164p-4;
    R = SMC_FMAK(R, z, 0x1.8618618618618p-4);
    R = SMC_FMAK(R, z, 0x1.af286bca1af28p-4);
    R = SMC_FMAK(R, z, 0x1.e1e1e1e1e1e1ep-4);
    R = SMC_FMAK(R, z, 0x1.1111111111111p-3);
    R = SMC_FMAK(R, z, 0x1.3b13b13b13b14p-3);
    R = SMC_FMAK(R, z, 0x1.745d1745d1746p-3);
    R = SMC_FMAK(R, z, 0x1.c71c71c71c71cp-3);
    R = SMC_FMAK(R, z, 0x1.2492492492492p-2);
    R = SMC_FMAK(R, z, 0x1.999999999999ap-2);
    R = SMC_FMAK(R, z, 0x1.5555555555555p-1);
    R = R * z;
    const double dk = (double)e;
    return dk * LN2_HI - ((s * (f - R) - dk * LN2_LO) - f);
}

SMC_HD double sp_log(double x) {
    if (x != x || x < 0.0) return bits2d(0x7ff8000000000000ULL);
    if (x == 0.0) return -inf();
    if (x == inf()) return x;
    int e = 0;
    if (x < 0x1p-1022) { x *= 0x1p54; e = -54; }
    return sp_log_normal(x, e);
}

// cos, sin of 2*pi*u for u in [0,1) a multiple of 2^-53
SMC_HD void sp_sincos2pi(double u, double& c, double& s) {
    const double a = 8.0 * u;
    const int oct = (int)a;
    // g = a - oct in the even octants, 1 - (a - oct) in the odd ones: |(oct rounded up to even) - a|, exact either way
    const double g = fabs((double)(oct + (oct & 1)) - a);
    const double y = g * PIO4;
    const double z = y * y;
    double ps = 0x1.952c77030ad4ap-49;
    ps = SMC_FMAK(ps, z, -0x1.ae7f3e733b81fp-41);
    ps = SMC_FMAK(ps, z, 0x1.6124613a86d09p-33);
    ps = SMC_FMAK(ps, z, -0x1.ae64567f544e4p-26);
    ps = SMC_FMAK(ps, z, 0x1.71de3a556c734p-19);
    ps = SMC_FMAK(ps, z, -0x1.a01a01a01a01ap-13);
    ps = SMC_FMAK(ps, z, 0x1.1111111111111p-7);
    ps = SMC_FMAK(ps, z, -0x1.5555555555555p-3);
    const double sy = fma(y * z, ps, y);
    double pc = -0x1.6827863b97d97p-53;
    pc = SMC_FMAK(pc, z, 0x1.ae7f3e733b81fp-45);
    pc = SMC_FMAK(pc, z, -0x1.93974a8c07c9dp-37);
    pc = SMC_FMAK(pc, z, 0x1.1eed8eff8d898p-29);
    pc = SMC_FMAK(pc, z, -0x1.27e4fb7789f5cp-22);
    pc = SMC_FMAK(pc, z, 0x1.a01a01a01a01ap-16);
    pc = SMC_FMAK(pc, z, -0x1.6c16c16c16c17p-10);
    pc = SMC_FMAK(pc, z, 0x1.5555555555555p-5);
    pc = SMC_FMAK(pc, z, -0.5);
    const double cy = SMC_FMAK(z, pc, 1.0);
    const bool swap = ((oct + 1) & 2) != 0;
    double cc = swap ? sy : cy;
    double ss = swap ? cy : sy;
    // cc = -cc in the octants 2..5, ss = -ss in 4..7: bit 2 of oct + 2 / of oct moved onto the sign bit
    const uint64_t cb = d2bits(cc), sb = d2bits(ss);
    const uint32_t ch = xor_and((uint32_t)(cb >> 32), ((uint32_t)oct << 29) + 0x40000000u, 0x80000000u);
    const uint32_t sh = xor_and((uint32_t)(sb >> 32), (uint32_t)oct << 29, 0x80000000u);
    c = bits2d(((uint64_t)ch << 32) | (uint32_t)cb);
    s = bits2d(((uint64_t)sh << 32) | (uint32_t)sb);
}

// four Philox words -> (z0, z1) iid N(0,1); z0 belongs to particle 2p, z1 to 2p+1
SMC_HD void box_muller(const u32x4& w, double& z0, double& z1) {
    // u1 = (n1 + 1) 2^-53, u2 = n2 2^-53 with n = (hi:lo) >> 11 = hi 2^21 + (lo >> 11): two exact terms whose sum is
    // representable, so one fma gives it exactly (instead of a 64-bit shift, a 64-bit add and a 64-bit conversion)
    const double u1 = fma((double)w.v[1], 0x1p-32, (double)((w.v[0] >> 11) + 1u) * TWO_M53);
    const double u2 = fma((double)w.v[3], 0x1p-32, (double)(w.v[2] >> 11) * TWO_M53);
    // u1 in [2^-53, 1] (positive, finite, normal): -2 log u1 is -0.0 (u1 = 1) or in [2.2e-16, 73.5]
    const double r = sqrt_moderate_or_negzero(-2.0 * sp_log_normal(u1, 0));
    double c, s;
    sp_sincos2pi(u2, c, s);
    z0 = r * c;
    z1 = r * s;
}

// ---- 64x64 -> 128 multiply ---------------------------------------------------------------
SMC_HD void mul64wide(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    hi = __umul64hi(a, b);
    lo = a * b;
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    hi = (uint64_t)(p >> 64);
    lo = (uint64_t)p;
#endif
}

// ---- break points of a multinomial resampling step (multi-segment filters) ------------------
// The n iid uniforms of resample() are generated sorted BY BLOCK (block = the seg consecutive children one
// workgroup owns): with E_i iid Exp(1) the order statistics are U_(k) = (E_1+..+E_k)/(E_1+..+E_{n+1}); a
// block of m ranks adds a Gamma(m) variate to these sums, so the largest uniform of block w is
// F_{w+1} = (g_0+..+g_w)/(g_0+..+g_{B-1}+e) and, given the break points, the other m-1 uniforms of the
// block are iid on (F_w, F_{w+1}).  The break points do not depend on the particles: a small kernel
// computes them for many steps ahead.  Everything after the Gamma variates is integer arithmetic.
SMC_HD double uniform53(const u32x4& w) {   // (0, 1], 53 bits, from words 0 and 1
    return (double)(((((uint64_t)w.v[1] << 32) | w.v[0]) >> 11) + 1) * TWO_M53;
}
// Gamma(m, 1), integer shape m >= 1, in 2^-32 fixed point (Marsaglia & Tsang 2000)
SMC_HD uint64_t gamma_fix(uint64_t seed, uint32_t w, uint32_t stream, uint32_t t, int64_t m) {
    const double d = (double)m - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double G = d;   // (all 8 attempts rejected: probability ~1e-15)
    for (uint32_t it = 0; it < 8; ++it) {
        double x, z1;
        box_muller(draw(seed, w, stream, t, SLOT_BREAK + 2 * it), x, z1);
        const double vv = 1.0 + c * x;
        if (!(vv > 0.0)) continue;
        const double v3 = vv * vv * vv, x2 = x * x;
        const double u = uniform53(draw(seed, w, stream, t, SLOT_BREAK + 2 * it + 1));
        if (u < 1.0 - 0.0331 * (x2 * x2) || sp_log(u) < 0.5 * x2 + d * ((1.0 - v3) + sp_log(v3))) {
            G = d * v3;
            break;
        }
    }
    const uint64_t g = (uint64_t)rne_pos(G * 0x1p32);
    return g ? g : 1;
}
SMC_HD uint64_t exp1_fix(uint64_t seed, uint32_t w, uint32_t stream, uint32_t t) {   // Exp(1), 2^-32 fixed point, >= 1 ulp
    const uint64_t e = (uint64_t)rne_pos(-sp_log(uniform53(draw(seed, w, stream, t, SLOT_BREAK + 1))) * 0x1p32);
    return e ? e : 1;
}
// floor(P * 2^64 / S) for P < S < 2^63 (binary long division: used once per block and step, off the hot path)
SMC_HD uint64_t div_frac64(uint64_t P, uint64_t S) {
    uint64_t rem = P, q = 0;
    for (int i = 0; i < 64; ++i) {
        rem <<= 1;
        q <<= 1;
        if (rem >= S) { rem -= S; q |= 1; }
    }
    return q;
}

// ---- systematic resampling targets (opt-in; SMC_FLAG_SYSTEMATIC) ---------------------------
// exact q = floor(D / N), r = D mod N for D < 2^63, 1 <= N < 2^31, without an integer divider:
// double-precision estimates through inv = 1/N (the first off by at most 2^12, the second by at most 1)
// followed by an exact integer correction of the remainder - the result is exact by construction.
SMC_HD void divmod_u64_u32(uint64_t D, uint32_t N, double inv, uint64_t& q, uint32_t& r) {
    if ((N & (N - 1u)) == 0u) {   // power of two: the common particle counts
        const int sft = __builtin_ctz(N);
        q = D >> sft;
        r = (uint32_t)(D & (uint64_t)(N - 1u));
        return;
    }
    uint64_t q1 = (uint64_t)((double)D * inv);
    int64_t r1 = (int64_t)(D - q1 * (uint64_t)N);          // |r1| < 2^44: exact in a double
    const int64_t q2 = (int64_t)floor((double)r1 * inv);
    q1 += (uint64_t)q2;
    r1 -= q2 * (int64_t)N;
    if (r1 < 0) { q1 -= 1; r1 += N; }
    if (r1 >= (int64_t)N) { q1 += 1; r1 -= N; }
    q = q1;
    r = (uint32_t)r1;
}
// T_j = floor((j * Dtot + v0) / n), v0 = mulhi64(u, Dtot), for the children j = j0 + k, k < 2^13:
// with Dtot = dq n + dr, v0 = q0 n + r0 and r0 + j0 dr = qa n + ra,
//     T_j = (q0 + j0 dq + qa) + k dq + floor((ra + k dr) / n),   ra + k dr < 2^44.
struct SysBase {
    uint64_t Tbase, dq;
    double inv;
    uint32_t ra, dr, n;
};
// inv = 1.0 / (double)n (a per-filter constant: the host passes it in)
SMC_HD SysBase sys_base(uint64_t Dtot, uint32_t n, double inv, uint64_t u, uint64_t j0) {
    uint64_t v0, lo;
    mul64wide(u, Dtot, v0, lo);
    SysBase sb;
    sb.n = n;
    sb.inv = inv;
    uint64_t q0, qa;
    uint32_t r0;
    divmod_u64_u32(Dtot, n, sb.inv, sb.dq, sb.dr);
    divmod_u64_u32(v0, n, sb.inv, q0, r0);
    divmod_u64_u32((uint64_t)r0 + j0 * (uint64_t)sb.dr, n, sb.inv, qa, sb.ra);
    sb.Tbase = q0 + j0 * sb.dq + qa;
    return sb;
}
SMC_HD uint64_t sys_target(const SysBase& sb, uint32_t k) {
    const uint64_t e = (uint64_t)sb.ra + (uint64_t)k * sb.dr;   // < 2^44
    uint64_t qe;
    if ((sb.n & (sb.n - 1u)) == 0u) {
        qe = e >> __builtin_ctz(sb.n);
    } else {
        qe = (uint64_t)((double)e * sb.inv);                    // floor(e / n) or one off: settle it exactly
        const int64_t rem = (int64_t)(e - qe * (uint64_t)sb.n);
        qe = rem < 0 ? qe - 1 : (rem >= (int64_t)sb.n ? qe + 1 : qe);
    }
    return sb.Tbase + (uint64_t)k * sb.dq + qe;
}
// (C >> sh) > T2  <=>  C > sys_threshold(T2, sh)
SMC_HD uint64_t sys_threshold(uint64_t T2, int sh) { return sh < 64 ? ((T2 + 1) << sh) - 1 : 0; }

SMC_HD double u128_to_double(uint64_t hi, uint64_t lo) { return (double)hi * TWO_P64 + (double)lo; }

SMC_HD int ceil_log2_i64(int64_t n) {
    int k = 0;
    while (((int64_t)1 << k) < n) ++k;
    return k;
}

// ---- models ------------------------------------------------------------------------------
// raw rows:  LG1D (A,B,Q,R,x0,sigma0)  Q,R,sigma0 VARIANCES (ssm.jl:93,102,108)
//            SV1D (mu,rho,sigma)
//            UCSV (gamma_eps,gamma_eta,x0,lse0,lsn0)  gammas STD-DEVs (ssm.jl:239-240)
// der rows:  LG1D (sQ,sR,s0,1/sR,c_obs)   SV1D (s0)
struct Params {
    double raw[NPARAM];
    double der[NPARAM];
};

SMC_HD void derive_params(int model, const double* raw, double* der) {
    for (int k = 0; k < NPARAM; ++k) der[k] = 0.0;
    if (model == MODEL_LG1D) {
        const double sR = sqrt(raw[3]);
        der[0] = sqrt(raw[2]);
        der[1] = sR;
        der[2] = sqrt(raw[5]);
        der[3] = 1.0 / sR;
        der[4] = -HALF_LOG2PI - sp_log(sR);
    } else if (model == MODEL_SV1D) {
        der[0] = raw[2] / sqrt(1.0 - raw[1] * raw[1]);
    }
}

// x = rand(initial_dist(model))            ssm.jl:105-109, 249-259
template <int MODEL>
SMC_HD void model_initial(const Params& p, const double* z, double* x) {
    if constexpr (MODEL == MODEL_LG1D) {
        x[0] = fma(p.der[2], z[0], p.raw[4]);
    } else if constexpr (MODEL == MODEL_SV1D) {
        x[0] = fma(p.der[0], z[0], p.raw[0]);
    } else {
        x[0] = fma(sp_exp(0.5 * p.raw[3]), z[0], p.raw[2]);
        x[1] = fma(p.raw[0], z[1], p.raw[3]);
        x[2] = fma(p.raw[1], z[2], p.raw[4]);
    }
}

// x = rand(transition(model, xp))          ssm.jl:87-94, 233-242
template <int MODEL>
SMC_HD void model_transition(const Params& p, const double* xp, const double* z, double* x) {
    if constexpr (MODEL == MODEL_LG1D) {
        x[0] = fma(p.der[0], z[0], p.raw[0] * xp[0]);
    } else if constexpr (MODEL == MODEL_SV1D) {
        x[0] = fma(p.raw[2], z[0], fma(p.raw[1], xp[0] - p.raw[0], p.raw[0]));
    } else {
        x[0] = fma(sp_exp(0.5 * xp[1]), z[0], xp[0]);
        x[1] = fma(p.raw[0], z[1], xp[1]);
        x[2] = fma(p.raw[1], z[2], xp[2]);
    }
}

// logpdf(observation(model, x), y)          ssm.jl:96-103, 244-247
template <int MODEL>
SMC_HD double model_logobs(const Params& p, const double* x, double y) {
    double z, c;
    if constexpr (MODEL == MODEL_LG1D) {
        z = (y - p.raw[1] * x[0]) * p.der[3];
        c = p.der[4];
    } else if constexpr (MODEL == MODEL_SV1D) {
        z = y * sp_exp(-0.5 * x[0]);
        c = fma(-0.5, x[0], -HALF_LOG2PI);
    } else {
        z = (y - x[0]) * sp_exp(-0.5 * x[2]);
        c = fma(-0.5, x[2], -HALF_LOG2PI);
    }
    return fma(-0.5 * z, z, c);
}

template <int MODEL>
SMC_HD void model_obs_moments(const Params& p, const double* x, double& mean, double& sd) {
    if constexpr (MODEL == MODEL_LG1D) {
        mean = p.raw[1] * x[0];
        sd = p.der[1];
    } else if constexpr (MODEL == MODEL_SV1D) {
        mean = 0.0;
        sd = sp_exp(0.5 * x[0]);
    } else {
        mean = x[0];
        sd = sp_exp(0.5 * x[2]);
    }
}

// ---- PMMH rejuvenation of the samplers (src/smc_samplers.jl:103-146), one parameter particle ------------------
// Random numbers are counter based like everything else: key = the move's seed, stream = GLOBAL index of the
// parameter particle (so results do not depend on how theta is sharded), t = chain position.
constexpr int PRIOR_UNIFORM = 1;      // par = (lo, hi)
constexpr int PRIOR_NORMAL = 2;       // par = (mu, sigma)
constexpr int PRIOR_TRUNCNORMAL = 3;  // par = (mu, sigma, lo, hi, log(Phi((hi-mu)/sigma) - Phi((lo-mu)/sigma)))
constexpr int PRIOR_LOGNORMAL = 4;    // par = (mu, sigma) of log x
constexpr int PRIOR_NPAR = 5;
struct PmmhSpec {
    int d;                              // parameter dimension
    int family[MAX_DTHETA];             // product_distribution([...]) component by component
    double par[MAX_DTHETA][PRIOR_NPAR];
    int nraw;                           // smc.model(theta): raw[k] = raw_from[k] >= 0 ? theta[raw_from[k]] : raw_const[k]
    int raw_from[NPARAM];
    double raw_const[NPARAM];
};
SMC_HD bool finite_d(double x) { return x == x && x != inf() && x != -inf(); }
// insupport(prior_i, x)   (smc_samplers.jl:116; Distributions' closed intervals)
SMC_HD bool prior_insupport(int fam, const double* par, double x) {
    switch (fam) {
    case PRIOR_UNIFORM: return par[0] <= x && x <= par[1];
    case PRIOR_NORMAL: return finite_d(x);
    case PRIOR_TRUNCNORMAL: return par[2] <= x && x <= par[3];
    case PRIOR_LOGNORMAL: return x > 0.0 && finite_d(x);
    }
    return false;
}
// logpdf(prior_i, x) for x in the support   (smc_samplers.jl:123; Normal: -(z^2 + log 2pi)/2 - log sigma)
SMC_HD double prior_logpdf(int fam, const double* par, double x) {
    switch (fam) {
    case PRIOR_UNIFORM: return -sp_log(par[1] - par[0]);
    case PRIOR_NORMAL: {
        const double z = (x - par[0]) / par[1];
        return -0.5 * (z * z + 2.0 * HALF_LOG2PI) - sp_log(par[1]);
    }
    case PRIOR_TRUNCNORMAL: {
        const double z = (x - par[0]) / par[1];
        return (-0.5 * (z * z + 2.0 * HALF_LOG2PI) - sp_log(par[1])) - par[4];
    }
    case PRIOR_LOGNORMAL: {
        const double lx = sp_log(x), z = (lx - par[0]) / par[1];
        return (-0.5 * (z * z + 2.0 * HALF_LOG2PI) - sp_log(par[1])) - lx;
    }
    }
    return -inf();
}
// insupport / logpdf of the product prior: components in order, sum from 0.0 left to right
SMC_HD bool pmmh_insupport(const PmmhSpec& s, const double* th) {
    bool ok = true;
    for (int i = 0; i < s.d; ++i) ok = ok && prior_insupport(s.family[i], s.par[i], th[i]);
    return ok;
}
SMC_HD double pmmh_logprior(const PmmhSpec& s, const double* th) {
    double lp = 0.0;
    for (int i = 0; i < s.d; ++i) lp = lp + prior_logpdf(s.family[i], s.par[i], th[i]);
    return lp;
}
// theta' = rand(MvNormal(theta, scale * Sigma)), Sigma = L L'  (smc_samplers.jl:99-100,114):
// theta'_i = theta_i + sq * sum_{k<=i} L[i][k] z_k, sq = sqrt(scale), the sum taken left to right
SMC_HD void pmmh_propose(const PmmhSpec& s, uint64_t seed, uint32_t stream, uint32_t c, const double* th, const double* L /*[d][d]*/,
                         double sq, double* prop) {
    double z[MAX_DTHETA + 1];
    for (int k = 0; k < s.d; k += 2) box_muller(draw(seed, (uint32_t)(k >> 1), stream, c, SLOT_PMMH_Z), z[k], z[k + 1]);
    for (int i = 0; i < s.d; ++i) {
        double a = 0.0;
        for (int k = 0; k <= i; ++k) a = a + L[i * s.d + k] * z[k];
        prop[i] = th[i] + sq * a;
    }
}
// log(rand()) of the accept test (smc_samplers.jl:129): u in (0, 1]
SMC_HD double pmmh_log_uniform(uint64_t seed, uint32_t stream, uint32_t c) {
    return sp_log(uniform53(draw(seed, 0u, stream, c, SLOT_PMMH_U)));
}
// smc.model(theta): parameter row of the model family
SMC_HD void pmmh_raw_row(const PmmhSpec& s, const double* th, double* raw /*[NPARAM]*/) {
    for (int k = 0; k < NPARAM; ++k) raw[k] = k < s.nraw ? (s.raw_from[k] >= 0 ? th[s.raw_from[k]] : s.raw_const[k]) : 0.0;
}

// ---- segment combine (integers only) ----------------------------------------------------------
// A segment record is (kb, S, S2): kb = max k_i of the segment (integral double, -inf if the segment
// has no live particle), S = sum q, S2 = sum q^2 (128 bit).  With K = max kb:
//   sh_b = (K - kb) + SH            right shift that brings segment b to the common scale 2^K
//   Q_b  = S_b  >> sh_b             entry of the segment table (uint64, total < 2^63)
//   R_b  = S2_b >> (2 (K-kb) + 49 + SH)
SMC_HD int table_shift_extra(int64_t npad) {
    const int s = ceil_log2_i64(npad) - 14;
    return s > 0 ? s : 0;
}
SMC_HD int seg_shift(double K, double kb, int SH) {
    const double dk = K - kb;   // >= 0 integral; inf or nan for a dead segment / dead filter
    int sh = (dk >= 0.0 && dk < 64.0) ? (int)dk + SH : 64;
    return sh > 64 ? 64 : sh;
}
SMC_HD uint64_t shr128(uint64_t hi, uint64_t lo, int s) {   // low 64 bits of (hi:lo) >> s, 0 < s < 128
    return s >= 64 ? (hi >> (s - 64)) : ((lo >> s) | (hi << (64 - s)));
}
SMC_HD uint64_t seg_Q(uint64_t S, int sh) { return sh < 64 ? S >> sh : 0; }
SMC_HD uint64_t seg_R(uint64_t S2hi, uint64_t S2lo, int sh, int SH) {
    const int sh2 = 2 * (sh - SH) + 49 + SH;
    return (sh < 64 && sh2 < 128) ? shr128(S2hi, S2lo, sh2) : 0;
}
SMC_HD void combine_outputs(double K, uint64_t Dtot, uint64_t Rtot, int SH, int64_t n, double& logmu, double& ess) {
    const double Dd = (double)Dtot * pow2i(SH - 48), Rd = (double)Rtot * pow2i(SH - 47);
    logmu = Dtot ? fma(K, LN2_HI, fma(K, LN2_LO, sp_log(Dd))) - sp_log((double)n) : -inf();
    ess = Rtot ? Dd * Dd / Rd : 0.0;
}

}  // namespace smc
