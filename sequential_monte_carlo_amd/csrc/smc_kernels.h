// smc_kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of the bootstrap
// particle filter hot path.  One workgroup owns one SEGMENT (SEG = 2*NP*THREADS particles)
// of one filter (theta); thread tau owns the NP particle pairs p = tau + k*THREADS, so every
// global store is a 16-byte-per-lane, fully coalesced double2 / ulonglong2.
//
//   k_init      bootstrap_filter        particles.jl:87-105   (sample x_1, weigh, normalise)
//   k_step      bootstrap_filter!       particles.jl:107-129  (resample+gather, propagate,
//                                                              weigh, normalise) - ONE launch
//   k_finalize  the (logmu, ess) return of normalize          particles.jl:10,12
//   k_resident  log_likelihood          particles.jl:132-147  (smc_resident.h)
//
// Weights never exist as doubles in memory: a segment keeps the inclusive prefix sums C of
// q_i = rint(p_i * 2^(48 + k_i - kb)) (uint64, exact, order independent; exp(logw_i) = p_i 2^k_i,
// kb = max k_i) plus its record (kb, S = sum q, S2 = sum q^2 as 128 bit).  The next launch's
// prologue turns the records of all segments of the filter into a second integer table (Dcum,
// by shifts only) in LDS.  resample() is exactly multinomial: the n iid uniforms are generated sorted BY
// BLOCK (block = the seg consecutive children of one workgroup) between break points that a small,
// particle-independent kernel (k_breaks, smc_aux_kernels.h) prepares many steps ahead - so a workgroup's
// children read the same few ancestor segments (L1/L2-resident, near-streaming) and one launch does a step:
//   k_step   builds the table, places its children between the block's break points (iid inside),
//            searches the segment table and C_b, gathers x[a], propagates, weighs, normalises.
// All searches of a thread advance level by level together (2*NP independent loads in flight).
#pragma once
#include "smc_spec.h"

namespace smc {

constexpr int WAVE = 64;
// LDS copies of a segment's prefix sums are padded by two entries per 32: a binary search probes
// index pos+s-1 with pos a multiple of 2s, so for s >= 32 EVERY lane's probe is = -1 (mod 32) and
// (8-byte entries, 64 banks) lands on one bank pair -- up to 32-way conflicts.  With the pad the
// probes of different 32-blocks fall on different banks; pairs stay 16-byte aligned.
__host__ __device__ constexpr int lds_pad(int i) { return i + ((i >> 5) << 1); }
// A search keeps the padded position pp = lds_pad(pos) next to pos: pos is a multiple of 2s when the
// level with stride s is probed, so the probe offset and the increment are compile-time constants.
__host__ __device__ constexpr int lds_probe_off(int s) { return s >= 32 ? lds_pad(s - 32) + 31 : s - 1; }   // lds_pad(pos+s-1) - pp
__host__ __device__ constexpr int lds_step_inc(int s) { return s >= 32 ? lds_pad(s) : s; }                  // lds_pad(pos+s) - pp
__host__ __device__ constexpr int lds_unpad(int pp) { return pp - 2 * (pp / 34); }                          // inverse of lds_pad
__host__ __device__ constexpr int lds_padded_len(int n) { return n + (n >> 4) + 8; }   // +8: staged neighbours start on different banks

// LDS pointers with their address space spelled out: 32-bit arithmetic (a generic pointer costs 64-bit adds and selects)
typedef __attribute__((address_space(3))) const char lds_byte;
typedef __attribute__((address_space(3))) const uint64_t lds_u64;
typedef __attribute__((address_space(3))) const double lds_f64;
__device__ __forceinline__ lds_byte* lds_ptr(const void* p) { return (lds_byte*)p; }
__device__ __forceinline__ uint64_t lds_load_u64(lds_byte* p) { return *(lds_u64*)p; }

// 16-byte global stores of a step's outputs (x, C): streaming (non-temporal) stores.  The next launch reads them through
// the Infinity Cache anyway; written this way they leave the XCD's write-back L2 during the launch instead of at its end,
// which shortens the gap between two dependent launches (measured on C2: 16.3 -> 15.5 us per step, same kernel span).
typedef unsigned long long nt_v2u64 __attribute__((ext_vector_type(2)));
// WT (the persistent step kernel, k_persist): write-through (sc1) stores - what another workgroup of the SAME launch is to read
// must leave this XCD's write-back L2 (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload, drained, then the flag)
template <bool WT = false, class T>
__device__ __forceinline__ void store_out(T* p, const T& val) {
    static_assert(sizeof(T) == 16, "16-byte stores");
    if (WT) {
        const nt_v2u64 v2 = *reinterpret_cast<const nt_v2u64*>(&val);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v2) : "memory");
    } else {
        __builtin_nontemporal_store(*reinterpret_cast<const nt_v2u64*>(&val), reinterpret_cast<nt_v2u64*>(p));
    }
}
template <bool WT, class T>
__device__ __forceinline__ void store_word(T* p, T val) {   // an 8-byte record word
    static_assert(sizeof(T) == 8, "8-byte words");
    if (WT) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = val;
}

struct FilterView {
    int64_t n;        // particles per filter (Nx)
    int64_t npad;     // nseg * seg
    int seg, nseg, nseg_p2, SH;
    int ntheta;
    uint64_t seed;
    const Params* params;    // [ntheta]
    const uint32_t* stream;  // [ntheta]
    double* x[2];            // [d][ntheta][npad]   ping-pong
    uint64_t* C[2];          // [ntheta][npad]
    double* segk[2];         // [ntheta][nseg]  segment exponent kb
    uint64_t* segS[2];
    uint64_t* segS2hi[2];
    uint64_t* segS2lo[2];
    const uint64_t* brk;     // break points F (2^-64 fixed point) of the steps [brk_t0, ..): [step][ntheta][nseg+1]
    uint32_t brk_t0;         //   (multi-segment multinomial resampling; smc_spec.h "break points")
    int32_t* anc;            // [ntheta][npad] or nullptr
    double* logZ;            // [ntheta]
    double* last_logmu;      // [ntheta]  (logmu, ess, K, D) of the most recently emitted weights
    double* last_ess;        // [ntheta]
    double* last_K;          // [ntheta]
    uint64_t* last_D;        // [ntheta]
    double* trace_logmu;     // [T][ntheta] or nullptr
    double* trace_ess;       // [T][ntheta] or nullptr
    const double* y;         // [T] on device (log_likelihood) or nullptr
    uint64_t* tabD;          // filters with more segments than a workgroup has threads: the segment table of the weights being
    int* tabsh;              // resampled, built ONCE per step by k_table ([ntheta][nseg_p2] inclusive sums / shifts) instead of by every
                             // workgroup of k_step (nseg^2 record reads and a table in LDS per workgroup); nullptr otherwise
    uint32_t host_seq;       // step API: after the three values, their writer stores this ticket in row 3 of host_out (system
                             // scope release) - the host spins on the pinned words instead of synchronising the stream; 0: no ticket
    double* host_out;        // pinned host mirror [4][ntheta] of (logZ | last_logmu | last_ess | ticket): whoever emits
                             //    these also stores them here, so the host needs no copy after its stream sync
    int systematic;          // opt-in systematic resampling (SMC_FLAG_SYSTEMATIC): selects the SYS kernels
    double inv_n;            // 1.0 / n  (systematic targets)
    int emit_now;            // single-segment step API: the launch emits (logmu, ess) of ITS OWN weights at the end
                             //    (one workgroup owns the whole filter), so no finalize launch follows
    int want_s2;             // 1: accumulate sum q^2 (ESS) in this launch; 0: its consumer never reads it
                             //    (log_likelihood discards ess, particles.jl:142: only the last step / traces need it)
    const unsigned char* skip;   // [ntheta] or nullptr: filters with skip[th] != 0 are not run by log_likelihood (PMMH proposals
                             //    outside the prior's support, smc_samplers.jl:116); their logZ reads -inf
    const int32_t* order;        // with skip, for the one-workgroup-per-filter kernel: a permutation of the filters, the
    const int32_t* n_active;     //    *n_active ones to run first - the workgroups that have work are then dealt evenly over
                             //    the CUs (speed only: measured 1.45x at 256 of 512 filters; scripts/skip_sweep.py)
    // per-step filtered summaries inside the multi-step calls (smc_set_summaries; README.md:33-61 computes quantile(x, ...) after
    // every bootstrap_filter!): weighted quantiles of state coordinate sum_comp at the levels sum_p64 (2^-64 fixed point) and
    // mean / variance of every coordinate, written per step by the kernel that owns the filter.  sum_np = sum_mom = 0: off.
    int sum_np, sum_comp, sum_mom;
    uint64_t sum_p64[8];
    double* sum_q;           // [T][ntheta][sum_np]
    double* sum_m;           // [T][2][d][ntheta]  (mean | variance)
    int abl;                 // ablation mask: always 0 in the product (profiling builds only, -DSMC_ABLATE)
    unsigned long long* dbg; // phase stamps [workgroup][8] (profiling builds only), else nullptr
};

#ifdef SMC_ABLATE
#define SMC_ABL(v, bit) (((v).abl >> (bit)) & 1)
// phase stamp k of this workgroup (100 MHz constant clock); diagnostic build only
#define SMC_STAMP(v, k)                                                                                   \
    do {                                                                                                  \
        if ((v).dbg && threadIdx.x == 0)                                                                  \
            (v).dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SMC_ABL(v, bit) 0
#define SMC_STAMP(v, k) do { } while (0)
#endif
// ---- weighted quantiles: helpers shared by the stand-alone kernels (smc_aux_kernels.h) and the per-step summaries --------
// (logZ, logmu, ess) of filter th into the pinned host mirror; with a ticket (step API) the values are released to the host
// before the ticket is: the host reads them as soon as it sees the ticket, without a stream synchronisation
template <class VIEW>
__device__ __forceinline__ void host_emit(const VIEW& v, int th, double z, double logmu, double ess) {
    if (!v.host_out) return;
    v.host_out[th] = z;
    v.host_out[(size_t)v.ntheta + th] = logmu;
    v.host_out[2 * (size_t)v.ntheta + th] = ess;
    if (v.host_seq) {
        __threadfence_system();
        __hip_atomic_store(&v.host_out[3 * (size_t)v.ntheta + th], (double)v.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
constexpr int QMAX = 8;      // quantile levels per call
__host__ __device__ inline uint64_t order_key(double x) {
    const uint64_t b = d2bits(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ inline double key_value(uint64_t k) { return bits2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k); }
__host__ __device__ inline uint64_t prob_to_u64(double p) {   // floor(p * 2^64) clamped
    if (!(p > 0.0)) return 0;
    if (p >= 1.0) return ~(uint64_t)0;
    return (uint64_t)(p * TWO_P64);
}


// Wave priority by phase of the step (0 = start .. 3 = normalisation): the EARLIER phase wins the issue slot.  Two
// workgroups share a CU; the arbiter otherwise favours the older one, which then finishes ~4 us before its neighbour
// and leaves it to run the tail of the launch alone at half occupancy.  With this rule whoever has fallen behind catches
// up and both end together (measured on C2: end-time spread of the workgroups 4.8 -> 2.1 us, 17.65 -> 16.3 us per step).
#ifndef SMC_NORMALS_EARLY
#define SMC_NORMALS_EARLY 0
#endif
#define SMC_PRIO(ph) __builtin_amdgcn_s_setprio((short)(3 - (ph)))

// threadIdx.x as the helpers below read it.  SMC_TID_OPAQUE (the translation unit of the persistent step kernel): behind an empty
// `asm volatile`, so that a loop around the step cannot hoist the lane / wave numbers and every LDS offset derived from them
// out of it and keep them in registers the step needs (k_persist: 22 spilled registers otherwise).
__device__ __forceinline__ int smc_tid() {
    int t = (int)threadIdx.x;
#if defined(SMC_TID_OPAQUE)
    asm volatile("" : "+v"(t));
#endif
    return t;
}

// ---------------------------------------------------------------------------------------------
// wave / block primitives (wave64)
// ---------------------------------------------------------------------------------------------
// DPP (data-parallel primitives) cross-lane moves: no LDS crossbar traffic, 1 VALU per dword.
// ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142 (row_mask 0xa), row_bcast:31 = 0x143 (row_mask 0xc).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// v + (v of the lane the DPP control names; 0 where there is none), 64-bit: the cross-lane read rides on the add and the
// add-with-carry themselves (two VALU instructions; moving the two halves first and adding then took five to seven).
// The leading s_nop covers the two wait states between a VALU write of a register and a DPP read of it, whatever precedes;
// the one between the add and the add-with-carry the two wait states gfx940-class hardware wants between a VALU write of VCC
// and a VALU read of it (the compiler puts the same s_nop 1 between its own v_add_co / v_addc_co pairs).
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_add_u64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#define SMC_DPP_ADD64(MOD)                                                                                  \
    asm volatile("s_nop 1\n\tv_add_co_u32_dpp %0, vcc, %0, %0 " MOD "\n\ts_nop 1\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " MOD \
                 : "+v"(lo), "+v"(hi) : : "vcc")
    static_assert(CTRL == 0x111 || CTRL == 0x112 || CTRL == 0x114 || CTRL == 0x118 || CTRL == 0x142 || CTRL == 0x143, "DPP control");
    if (CTRL == 0x111) SMC_DPP_ADD64("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x112) SMC_DPP_ADD64("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x114) SMC_DPP_ADD64("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x118) SMC_DPP_ADD64("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x142) SMC_DPP_ADD64("row_bcast:15 row_mask:0xa bank_mask:0xf");
    if (CTRL == 0x143) SMC_DPP_ADD64("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SMC_DPP_ADD64
    return ((uint64_t)hi << 32) | lo;
}
// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int /*lane*/) {
    v = dpp_add_u64<0x111>(v);
    v = dpp_add_u64<0x112>(v);
    v = dpp_add_u64<0x114>(v);
    v = dpp_add_u64<0x118>(v);
    v = dpp_add_u64<0x142>(v);
    v = dpp_add_u64<0x143>(v);
    return v;
}
// 32-bit inclusive prefix sum (the compiler folds each move into one v_add_u32_dpp)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += dpp_u32<0x111, 0xf>(0u, v);
    v += dpp_u32<0x112, 0xf>(0u, v);
    v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v);
    v += dpp_u32<0x142, 0xa>(0u, v);
    v += dpp_u32<0x143, 0xc>(0u, v);
    return v;
}
// inclusive prefix sum of values < 2^52 as two 26-bit limbs: 64 lanes x 2^26 fits 32 bits, so both
// limb scans are 32-bit DPP adds (12 VALU instead of ~36 for the 64-bit scan)
__device__ __forceinline__ uint64_t wave_incl_scan_52(uint64_t v) {
    const uint32_t lo = wave_incl_scan_u32((uint32_t)v & 0x3ffffffu);
    const uint32_t hi = wave_incl_scan_u32((uint32_t)(v >> 26));
    return ((uint64_t)hi << 26) + lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) { return readlane_u64(wave_incl_scan(v, 0), WAVE - 1); }
// Combine the NW (<= 16) per-wave totals tot[0..NW) in LDS: off = sum of the totals of the waves before
// this one, all = their sum.  Lane w reads tot[w], a DPP scan inside the first row of 16 lanes, two lane
// reads - instead of a loop over NW with selects in every thread.  `wave` must be wave-uniform.
template <int NW>
__device__ __forceinline__ void wave_totals(const uint64_t* tot, int lane, int wave, uint64_t& off, uint64_t& all) {
    static_assert(NW >= 1 && NW <= 16 && (NW & (NW - 1)) == 0, "wave count");
    uint64_t t = tot[lane & (NW - 1)];
    if (NW > 1) t = dpp_add_u64<0x111>(t);
    if (NW > 2) t = dpp_add_u64<0x112>(t);
    if (NW > 4) t = dpp_add_u64<0x114>(t);
    if (NW > 8) t = dpp_add_u64<0x118>(t);
    const int ws = __builtin_amdgcn_readfirstlane(wave);
    all = readlane_u64(t, NW - 1);
    const uint64_t prev = readlane_u64(t, ws > 0 ? ws - 1 : 0);
    off = ws > 0 ? prev : 0;
}

// 128-bit unsigned accumulator (sum of q^2).  NOTE: do not "optimise" this into 24-bit limb
// products: hipcc 7.2 folds (q & 0xFFFFFF)^2 accumulations into v_mad_u64_u32 on the UNMASKED
// register (wrong sums); the parity tests against the oracle caught it.
struct U128 {
    uint64_t lo, hi;
};
__device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
__device__ __forceinline__ U128 sq128(uint64_t q) {
    U128 r;
    r.hi = __umul64hi(q, q);
    r.lo = q * q;
    return r;
}
// the 128-bit version of dpp_add_u64: one add and three adds-with-carry, each reading its other lane itself
template <int CTRL>
__device__ __forceinline__ U128 dpp_add_u128(U128 v) {
    uint32_t w0 = (uint32_t)v.lo, w1 = (uint32_t)(v.lo >> 32), w2 = (uint32_t)v.hi, w3 = (uint32_t)(v.hi >> 32);
#define SMC_DPP_ADD128(MOD)                                                                                       \
    asm volatile("s_nop 1\n\tv_add_co_u32_dpp %0, vcc, %0, %0 " MOD "\n\ts_nop 1\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " MOD \
                 "\n\ts_nop 1\n\tv_addc_co_u32_dpp %2, vcc, %2, %2, vcc " MOD "\n\ts_nop 1\n\tv_addc_co_u32_dpp %3, vcc, %3, %3, vcc " MOD \
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : : "vcc")
    static_assert(CTRL == 0x111 || CTRL == 0x112 || CTRL == 0x114 || CTRL == 0x118 || CTRL == 0x142 || CTRL == 0x143, "DPP control");
    if (CTRL == 0x111) SMC_DPP_ADD128("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x112) SMC_DPP_ADD128("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x114) SMC_DPP_ADD128("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x118) SMC_DPP_ADD128("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (CTRL == 0x142) SMC_DPP_ADD128("row_bcast:15 row_mask:0xa bank_mask:0xf");
    if (CTRL == 0x143) SMC_DPP_ADD128("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SMC_DPP_ADD128
    return U128{((uint64_t)w1 << 32) | w0, ((uint64_t)w3 << 32) | w2};
}
// total over the wave, valid in every lane
__device__ __forceinline__ U128 wave_sum128(U128 v) {
    v = dpp_add_u128<0x111>(v);
    v = dpp_add_u128<0x112>(v);
    v = dpp_add_u128<0x114>(v);
    v = dpp_add_u128<0x118>(v);
    v = dpp_add_u128<0x142>(v);
    v = dpp_add_u128<0x143>(v);
    return U128{readlane_u64(v.lo, WAVE - 1), readlane_u64(v.hi, WAVE - 1)};
}
// max of an int over the wave, valid in every lane
__device__ __forceinline__ int wave_max_i32(int v) {
    constexpr uint32_t LOWEST = 0x80000000u;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    v = mx(v, (int)dpp_u32<0x111, 0xf>(LOWEST, (uint32_t)v));
    v = mx(v, (int)dpp_u32<0x112, 0xf>(LOWEST, (uint32_t)v));
    v = mx(v, (int)dpp_u32<0x114, 0xf>(LOWEST, (uint32_t)v));
    v = mx(v, (int)dpp_u32<0x118, 0xf>(LOWEST, (uint32_t)v));
    v = mx(v, (int)dpp_u32<0x142, 0xa>(LOWEST, (uint32_t)v));
    v = mx(v, (int)dpp_u32<0x143, 0xc>(LOWEST, (uint32_t)v));
    return __builtin_amdgcn_readlane(v, WAVE - 1);
}
// a double moved across lanes by DPP (two 32-bit moves; lanes without a source keep `old`)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const uint64_t o = d2bits(old), b = d2bits(v);
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)o, (uint32_t)b), hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(o >> 32), (uint32_t)(b >> 32));
    return bits2d(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) { return bits2d(readlane_u64(d2bits(v), l)); }
// sum / maximum of a double over the wave by the DPP scan ladder (fixed order: lane l accumulates the lanes below it), valid in
// every lane; no LDS crossbar round trips (the butterfly of __shfl_xor is six dependent ds_bpermute pairs)
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0x111, 0xf>(0.0, v);
    v += dpp_f64<0x112, 0xf>(0.0, v);
    v += dpp_f64<0x114, 0xf>(0.0, v);
    v += dpp_f64<0x118, 0xf>(0.0, v);
    v += dpp_f64<0x142, 0xa>(0.0, v);
    v += dpp_f64<0x143, 0xc>(0.0, v);
    return readlane_f64(v, WAVE - 1);
}
__device__ __forceinline__ double wave_max_f64(double v) {   // NaN-free inputs
    auto mx = [](double a, double b) { return a > b ? a : b; };
    v = mx(v, dpp_f64<0x111, 0xf>(-inf(), v));
    v = mx(v, dpp_f64<0x112, 0xf>(-inf(), v));
    v = mx(v, dpp_f64<0x114, 0xf>(-inf(), v));
    v = mx(v, dpp_f64<0x118, 0xf>(-inf(), v));
    v = mx(v, dpp_f64<0x142, 0xa>(-inf(), v));
    v = mx(v, dpp_f64<0x143, 0xc>(-inf(), v));
    return readlane_f64(v, WAVE - 1);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        const double o = __shfl_xor(v, d, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// max over the workgroup; every thread gets the result. `red` : NW doubles of LDS that nobody
// writes again before the workgroup's next barrier (ONE barrier here).
template <int THREADS>
__device__ __forceinline__ double block_max(double v, double* red) {
    constexpr int NW = THREADS / WAVE;
    const int tid_ = smc_tid(), lane = tid_ & (WAVE - 1), wave = tid_ / WAVE;
    v = wave_max(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = red[w] > r ? red[w] : r;
    return r;
}

// the same for an int (segment exponents), DPP reduction inside the wave
template <int THREADS>
__device__ __forceinline__ int block_max_i32(int v, int* red) {
    constexpr int NW = THREADS / WAVE;
    const int tid_ = smc_tid(), lane = tid_ & (WAVE - 1), wave = tid_ / WAVE;
    v = wave_max_i32(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    int r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = red[w] > r ? red[w] : r;
    return r;
}

// ---------------------------------------------------------------------------------------------
// LDS carve of the step / finalize kernels (dynamic LDS, 16-B aligned, no static LDS)
// ---------------------------------------------------------------------------------------------
struct TableLds {
    uint64_t* Dcum;  // [nseg_p2] inclusive sums of Q_b
    int* sh;         // [nseg_p2] shift of segment b (64 = unreachable)
    uint64_t* scr;   // scratch
};
__host__ __device__ inline size_t scr_words(int threads, int np) {
    return (size_t)((np + 3 > 4 ? np + 3 : 4) * (threads / WAVE) + 16);   // [red | wave totals ... | 8 window words | 8 tail words]
}
__host__ __device__ inline size_t table_lds_bytes(int nseg_p2, int threads, int np) {
    return (size_t)nseg_p2 * 16 + scr_words(threads, np) * 8;   // sh padded to 8 B per entry
}
__device__ __forceinline__ TableLds carve(char* smem, int nseg_p2) {
    TableLds t;
    t.Dcum = (uint64_t*)smem;
    t.sh = (int*)(smem + (size_t)nseg_p2 * 8);
    t.scr = (uint64_t*)(smem + (size_t)nseg_p2 * 16);
    return t;
}

// Segment-table prologue: builds Dcum / sh in LDS from the records of filter `th` in buffer
// `cur` (integer shifts only); returns Dtot.  If `emit`, thread 0 also produces (logmu, ess) of
// those weights - the return value of normalize(), particles.jl:10,12 - and adds logmu to logZ.
// this thread's record words when the table has one entry per thread (the usual case): loaded up front
// - by table_preload, as early as the caller likes - so the workgroup pays ONE global round trip
struct TablePre {
    double k1;
    uint64_t S1, hi1, lo1;
    double k2;      // RPT = 2 (two records per thread: filters with up to twice as many segments as threads): this thread's records are
    uint64_t S2;    // 2 tid (k1, S1) and 2 tid + 1 (k2, S2)
};
// (VIEW: FilterView, or the same struct read in place from the kernel-argument segment - the persistent step kernel)
template <int THREADS, class VIEW, int RPT = 1>
__device__ __forceinline__ TablePre table_preload(const VIEW& v, int cur, int th, bool emit) {
    TablePre p{-inf(), 0, 0, 0, -inf(), 0};
    const int tid = smc_tid();
    const size_t base = (size_t)th * v.nseg;
    if (RPT == 2) {   // records 2 tid and 2 tid + 1 (the table itself, when somebody needs it, is built by table_prologue from memory)
        if (2 * tid < v.nseg) { p.k1 = v.segk[cur][base + 2 * tid]; p.S1 = v.segS[cur][base + 2 * tid]; }
        if (2 * tid + 1 < v.nseg) { p.k2 = v.segk[cur][base + 2 * tid + 1]; p.S2 = v.segS[cur][base + 2 * tid + 1]; }
        return p;
    }
    if (v.nseg_p2 <= THREADS && tid < v.nseg) {
        p.k1 = v.segk[cur][base + tid];
        p.S1 = v.segS[cur][base + tid];
        if (emit) { p.hi1 = v.segS2hi[cur][base + tid]; p.lo1 = v.segS2lo[cur][base + tid]; }
    }
    return p;
}
template <int THREADS, class VIEW>
__device__ __forceinline__ uint64_t table_prologue(const VIEW& v, int cur, int th, const TableLds& L, bool emit,
                                                   bool first_emit, uint32_t t_emit, const TablePre* pre = nullptr, double* Kout = nullptr) {
    constexpr int NW = THREADS / WAVE;
    const int tid = smc_tid(), lane = tid & (WAVE - 1), wave = tid / WAVE;
    const size_t base = (size_t)th * v.nseg;
    const double* sk = v.segk[cur] + base;
    const uint64_t* sS = v.segS[cur] + base;
    double* red = (double*)L.scr;

    const bool one = v.nseg_p2 <= THREADS;
    const TablePre pl = pre ? *pre : table_preload<THREADS>(v, cur, th, emit);
    const double k1 = pl.k1;
    const uint64_t S1 = pl.S1, hi1 = pl.hi1, lo1 = pl.lo1;
    double K = k1;
    if (!one)
        for (int b = tid; b < v.nseg; b += THREADS) { const double k = sk[b]; K = k > K ? k : K; }
    {   // segment exponents are integers of magnitude < 2^30 (or -inf): reduce them as int32 (DPP max)
        constexpr int DEADK = (int)0x80000000;
        const int ki = K == -inf() ? DEADK : (int)K;
        const int km = block_max_i32<THREADS>(ki, (int*)red);
        K = km == DEADK ? -inf() : (double)km;
    }
    if (Kout) *Kout = K;

    // blocked layout: thread owns E consecutive table entries
    const int E = v.nseg_p2 >= THREADS ? v.nseg_p2 / THREADS : 1;
    uint64_t run = 0, rsum = 0;
    if (one) {
        if (tid < v.nseg_p2) {
            int sh = 64;
            if (tid < v.nseg) {
                sh = seg_shift(K, k1, v.SH);
                run = seg_Q(S1, sh);
                if (emit) rsum = seg_R(hi1, lo1, sh, v.SH);
            }
            L.Dcum[tid] = run;
            L.sh[tid] = sh;
        }
    } else if (tid * E < v.nseg_p2) {
        for (int e = 0; e < E; ++e) {
            const int b = tid * E + e;
            uint64_t Qb = 0;
            int sh = 64;
            if (b < v.nseg) {
                sh = seg_shift(K, sk[b], v.SH);
                Qb = seg_Q(sS[b], sh);
                if (emit) rsum += seg_R(v.segS2hi[cur][base + b], v.segS2lo[cur][base + b], sh, v.SH);
            }
            run += Qb;
            L.Dcum[b] = run;  // thread-local inclusive, fixed up below
            L.sh[b] = sh;
        }
    }
    const uint64_t incl = wave_incl_scan(run, lane);
    uint64_t* wt = L.scr + NW;       // [NW] wave totals of Q
    uint64_t* wr = L.scr + 2 * NW;   // [NW] wave totals of R
    if (lane == WAVE - 1) wt[wave] = incl;
    if (emit) {
        const uint64_t rw = wave_sum(rsum);
        if (lane == 0) wr[wave] = rw;
    }
    __syncthreads();
    uint64_t off, Dtot;
    wave_totals<NW>(wt, lane, wave, off, Dtot);
    const uint64_t excl = off + incl - run;
    if (tid * E < v.nseg_p2)
        for (int e = 0; e < E; ++e) {
            const uint64_t d = L.Dcum[tid * E + e] + excl;
            L.Dcum[tid * E + e] = d;
        }
    if (emit && tid == 0) {
        uint64_t Rtot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) Rtot += wr[w];
        double logmu, ess;
        combine_outputs(K, Dtot, Rtot, v.SH, v.n, logmu, ess);
        v.last_logmu[th] = logmu;
        v.last_ess[th] = ess;
        v.last_K[th] = K;
        v.last_D[th] = Dtot;
        if (v.trace_logmu) v.trace_logmu[(size_t)t_emit * v.ntheta + th] = logmu;
        if (v.trace_ess) v.trace_ess[(size_t)t_emit * v.ntheta + th] = ess;
        const double z = first_emit ? logmu : v.logZ[th] + logmu;
        v.logZ[th] = z;
        host_emit(v, th, z, logmu, ess);
    }
    __syncthreads();
    return Dtot;
}

// Window prologue: what a workgroup of k_step needs from the segment table in the usual case - the total Dtot and the
// entries Dcum[lo-1], Dcum[lo .. lo+NE-1], sh[lo .. lo+NE-1] of its speculative window - WITHOUT building the table in LDS:
// one record per thread (nseg_p2 <= THREADS), the exponent maximum, one wave scan of the Q_b, the wave totals, and the NE + 1
// values picked out of the lanes that hold them.  Two barriers (the full prologue: three, plus the table traffic).  The
// numbers are the same integers table_prologue produces.  P0 = Dcum[lo-1] (0 for lo = 0), Dc[r] = Dcum[lo+r].
template <int THREADS, int NE, class VIEW, int RPT = 1>
__device__ __forceinline__ uint64_t window_prologue(const VIEW& v, uint64_t* scr, const TablePre& pre, int lo, uint64_t& P0,
                                                    uint64_t (&Dc)[NE], int (&shw)[NE], double& Kout) {
    constexpr int NW = THREADS / WAVE;
    constexpr int DEADK = (int)0x80000000;
    const int tid = smc_tid(), lane = tid & (WAVE - 1), wave = tid / WAVE;
    // RPT = 1: thread t holds the record of segment t; RPT = 2: of segments 2 t and 2 t + 1 (nseg <= 2 THREADS)
    const bool live = RPT * tid < v.nseg, live2 = RPT == 2 && 2 * tid + 1 < v.nseg;
    int ki = (!live || pre.k1 == -inf()) ? DEADK : (int)pre.k1;
    if (RPT == 2) { const int k2 = (!live2 || pre.k2 == -inf()) ? DEADK : (int)pre.k2; ki = k2 > ki ? k2 : ki; }
    const int km = block_max_i32<THREADS>(ki, (int*)scr);            // barrier 1
    const double K = km == DEADK ? -inf() : (double)km;
    Kout = K;
    int sh = 64, sh2 = 64;
    uint64_t Q = 0, Q2 = 0;
    if (live) { sh = seg_shift(K, pre.k1, v.SH); Q = seg_Q(pre.S1, sh); }
    if (live2) { sh2 = seg_shift(K, pre.k2, v.SH); Q2 = seg_Q(pre.S2, sh2); }
    const uint64_t incl = wave_incl_scan(Q + Q2, lane);   // inclusive (inside the wave) at this thread's LAST segment
    static_assert(2 * NE + 1 <= 8, "window words");
    uint64_t* wt = scr + NW;              // [NW] wave totals
    uint64_t* slot = scr + scr_words(THREADS, 1) - 16;   // the 8 window words (the same place for every NP: 4*NW + 0..7):
                                          // [0]: inclusive sum (inside its wave) at segment lo-1; [1..NE]: Q; [NE+1 .. 2NE]: sh
    if (lane == WAVE - 1) wt[wave] = incl;
    if (RPT == 1) {
        if (tid == lo - 1) slot[0] = incl;
        const int r = tid - lo;
        if (r >= 0 && r < NE) { slot[1 + r] = Q; slot[1 + NE + r] = (uint64_t)(uint32_t)sh; }
    } else {
        if (lo > 0 && tid == (lo - 1) >> 1) slot[0] = ((lo - 1) & 1) ? incl : incl - Q2;
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int e = lo + r;
            if (tid == e >> 1) { slot[1 + r] = (e & 1) ? Q2 : Q; slot[1 + NE + r] = (uint64_t)(uint32_t)((e & 1) ? sh2 : sh); }
        }
    }
    __syncthreads();                                                 // barrier 2
    // totals of the waves: lanes 0..NW-1 of every wave scan them (DPP), Dtot and the offset of the wave holding segment lo-1
    uint64_t t = wt[lane & (NW - 1)];
    if (NW > 1) t = dpp_add_u64<0x111>(t);
    if (NW > 2) t = dpp_add_u64<0x112>(t);
    if (NW > 4) t = dpp_add_u64<0x114>(t);
    if (NW > 8) t = dpp_add_u64<0x118>(t);
    const uint64_t Dtot = readlane_u64(t, NW - 1);
    uint64_t p0 = 0;
    if (lo > 0) {                                                    // workgroup-uniform
        const int ws = __builtin_amdgcn_readfirstlane(((lo - 1) / RPT) / WAVE);
        const uint64_t before = ws > 0 ? readlane_u64(t, ws - 1) : 0;
        p0 = before + slot[0];
    }
    P0 = p0;
    uint64_t run = p0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        run += slot[1 + i];
        Dc[i] = run;
        shw[i] = (int)(uint32_t)slot[1 + NE + i];
    }
    return Dtot;
}

// ---------------------------------------------------------------------------------------------
// segment normalisation: normalize() of one segment.  lw[k][j] = log-weight of particle 2p+j of
// pair p = tau + k*THREADS (masked particles carry NaN).  Writes the inclusive sums C.
// ---------------------------------------------------------------------------------------------
struct SegRec {
    double kb;
    uint64_t S;       // valid in every thread
    uint64_t hi, lo;  // valid in thread 0 only
};

// Cout: where the segment's inclusive sums go (global buffer or LDS), 16-B aligned.
// PADDED: Cout is an LDS copy indexed through lds_pad().
template <int THREADS, int NP, bool PADDED = false, bool WT = false>
__device__ __forceinline__ SegRec segment_normalize(double (&lw)[NP][2], uint64_t* scr, uint64_t* Cout, bool want_s2) {
    constexpr int NW = THREADS / WAVE;
    const int tid = smc_tid(), lane = tid & (WAVE - 1), wave = tid / WAVE;
    // exp(logw) = p 2^k for every particle: independent of the maximum, so it overlaps the reduction
    // a dead particle (NaN, infinite or absurd log-weight) takes p = 0 and the exponent DEAD = -2^30, below every live one
    // (lw_alive: |k| < 2^30), so that k - kb never wraps: nothing after the maximum has to ask again who is alive
    constexpr int DEAD = -(1 << 30);
    double p[NP][2];
    double kq[NP][2];
    int kk[NP][2];
    int kloc = DEAD;
    // the exponents first (two instructions each): their maximum is a chain of cross-lane steps, an LDS round trip and a
    // barrier - started here, it runs under the polynomials instead of behind them
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double l = lw[k][j];
            kq[k][j] = sp_exp_k(l);
            kk[k][j] = lw_alive(l) ? (int)kq[k][j] : DEAD;
            kloc = kk[k][j] > kloc ? kk[k][j] : kloc;
        }
    }
    {
        const int wm = wave_max_i32(kloc);
        if (lane == 0) ((int*)scr)[wave] = wm;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double l = lw[k][j];
            const double pe = sp_exp_p(l, kq[k][j]);   // garbage for a dead particle: dropped here
            p[k][j] = lw_alive(l) ? pe : 0.0;
            // keep the polynomial HERE (the compiler otherwise sinks it behind the barrier into
            // divergent per-particle branches and re-materialises its constants in each of them)
            asm volatile("" : "+v"(p[k][j]));
        }
    }
    __syncthreads();
    int kbi = ((int*)scr)[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) kbi = ((int*)scr)[w] > kbi ? ((int*)scr)[w] : kbi;
    const double kb = kbi == DEAD ? -inf() : (double)kbi;

    uint64_t q[NP][2], ps[NP], incl[NP];
    U128 s2{0, 0};   // sum q^2
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // fix_weight_i(p, k - kb, FIX_BITS), branch-free: the exponent is clamped at -8, where p 2^-8 < 1/2 rounds to the
            // 0 that fix_weight_i returns below -(FIX_BITS + 2), and between the two p 2^ek <= 1.42 / 8 rounds to 0 as well
            const int dk = kk[k][j] - kbi;   // <= 0, >= -2^31
            const int ek = (dk > -(FIX_BITS + 8) ? dk : -(FIX_BITS + 8)) + FIX_BITS;
            const uint64_t qq = d2bits(scale2(p[k][j], ek) + 0x1p52) & 0x000fffffffffffffULL;
            q[k][j] = qq;
            if (want_s2) s2 = add128(s2, sq128(qq));
        }
        ps[k] = q[k][0] + q[k][1];
        incl[k] = wave_incl_scan_52(ps[k]);   // q < 1.42 * 2^48: a pair sum is below 2^50
    }
    if (want_s2) s2 = wave_sum128(s2);
    uint64_t* wtot = scr + NW;            // [NP][NW]   (scr[0..NW) is block_max's)
    uint64_t* w2 = scr + (NP + 1) * NW;   // [2][NW]
    if (lane == WAVE - 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) wtot[k * NW + wave] = incl[k];
    }
    if (lane == 0) { w2[wave] = s2.lo; w2[NW + wave] = s2.hi; }
    __syncthreads();
    uint64_t basek = 0;
    // prefix of this wave's slice in the order (k, wave) - the order of the particles.  All NP*NW totals fit one row of 16
    // lanes in the usual geometries: ONE DPP scan gives every (k, wave) offset and the segment total
    constexpr bool MERGED = NP > 1 && NP * NW <= 16;
    uint64_t tt = 0;
    if (MERGED) {
        tt = wtot[lane & (NP * NW - 1)];
        tt = dpp_add_u64<0x111>(tt);
        if (NP * NW > 2) tt = dpp_add_u64<0x112>(tt);
        if (NP * NW > 4) tt = dpp_add_u64<0x114>(tt);
        if (NP * NW > 8) tt = dpp_add_u64<0x118>(tt);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        uint64_t off, ktot;
        if (MERGED) {
            const int j = __builtin_amdgcn_readfirstlane(k * NW + wave);
            off = j > 0 ? readlane_u64(tt, j > 0 ? j - 1 : 0) : 0;     // everything before (k, wave): basek is inside
            ktot = 0;
        } else {
            wave_totals<NW>(wtot + k * NW, lane, wave, off, ktot);
        }
        const uint64_t excl = basek + off + incl[k] - ps[k];
        ulonglong2 cc;
        cc.x = excl + q[k][0];
        cc.y = cc.x + q[k][1];
        if (PADDED) *reinterpret_cast<ulonglong2*>(Cout + lds_pad(2 * (tid + k * THREADS))) = cc;
        else store_out<WT>(reinterpret_cast<ulonglong2*>(Cout + 2 * (tid + k * THREADS)), cc);
        basek += ktot;
    }
    if (MERGED) basek = readlane_u64(tt, NP * NW - 1);
    SegRec rec;
    rec.kb = kb;
    rec.S = basek;
    rec.hi = rec.lo = 0;
    if (tid == 0) {
        U128 t{0, 0};
#pragma unroll
        for (int w = 0; w < NW; ++w) t = add128(t, U128{w2[w], w2[NW + w]});
        rec.hi = t.hi;
        rec.lo = t.lo;
    }
    return rec;
}

// normalize() of one segment into the global ping-pong buffers
template <int THREADS, int NP, bool WT = false, class VIEW>
__device__ __forceinline__ SegRec segment_epilogue(const VIEW& v, int nxt, int th, int sb, double (&lw)[NP][2],
                                                   uint64_t* scr) {
    uint64_t* Cout = v.C[nxt] + (size_t)th * v.npad + (size_t)sb * v.seg;
    const SegRec rec = segment_normalize<THREADS, NP, false, WT>(lw, scr, Cout, v.want_s2 != 0);
    if (smc_tid() == 0) {
        const size_t r = (size_t)th * v.nseg + sb;
        store_word<WT>(&v.segk[nxt][r], rec.kb);
        store_word<WT>(&v.segS[nxt][r], rec.S);
        store_word<WT>(&v.segS2hi[nxt][r], rec.hi);
        store_word<WT>(&v.segS2lo[nxt][r], rec.lo);
    }
    return rec;
}

// (logmu, ess) of a single-segment filter from its own record: exactly what table_prologue's emit
// computes for a one-entry table (K = kb).  Thread 0 only.
template <class VIEW>
__device__ __forceinline__ void emit_own(const VIEW& v, int th, const SegRec& rec, uint32_t t_emit, bool first_emit) {
    const int sh = seg_shift(rec.kb, rec.kb, v.SH);
    const uint64_t D = seg_Q(rec.S, sh), R = seg_R(rec.hi, rec.lo, sh, v.SH);
    double logmu, ess;
    combine_outputs(rec.kb, D, R, v.SH, v.n, logmu, ess);
    v.last_logmu[th] = logmu;
    v.last_ess[th] = ess;
    v.last_K[th] = rec.kb;
    v.last_D[th] = D;
    if (v.trace_logmu) v.trace_logmu[(size_t)t_emit * v.ntheta + th] = logmu;
    if (v.trace_ess) v.trace_ess[(size_t)t_emit * v.ntheta + th] = ess;
    const double z = first_emit ? logmu : v.logZ[th] + logmu;
    v.logZ[th] = z;
    host_emit(v, th, z, logmu, ess);
}

// (logmu, ess = 0) of the previous step from the totals (K, Dtot) alone: what table_prologue's emit produces when the
// records carry no sum of squares (want_s2 = 0 at that step: log_likelihood without traces, particles.jl:142 discards
// ess).  Thread 0 of the emitting workgroup; no table, no extra barrier.
template <class VIEW>
__device__ __forceinline__ void emit_from_totals(const VIEW& v, int th, double K, uint64_t Dtot, bool first_emit, uint32_t t_emit) {
    double logmu, ess;
    combine_outputs(K, Dtot, 0, v.SH, v.n, logmu, ess);
    v.last_logmu[th] = logmu;
    v.last_ess[th] = ess;
    v.last_K[th] = K;
    v.last_D[th] = Dtot;
    if (v.trace_logmu) v.trace_logmu[(size_t)t_emit * v.ntheta + th] = logmu;
    if (v.trace_ess) v.trace_ess[(size_t)t_emit * v.ntheta + th] = ess;
    const double z = first_emit ? logmu : v.logZ[th] + logmu;
    v.logZ[th] = z;
    host_emit(v, th, z, logmu, ess);
}

__device__ __forceinline__ double nan_mask() { return bits2d(0x7ff8000000000000ULL); }

// XCD-aware block -> segment map (speed only, never correctness): workgroups are dealt round-robin
// over the 8 XCDs, so blocks with equal blockIdx.x % 8 share an L2.  Give each XCD a CONTIGUOUS
// range of segments: the workgroup that wrote segment b at step t-1 and the workgroups that read
// segments b-1..b+1 at step t (children are ordered by block of the weight CDF) then sit on the same XCD, and the
// staging loads / gathers hit that XCD's 4 MiB L2 instead of going out to the Infinity Cache.
__device__ __forceinline__ int logical_segment(int bid, int nseg) {
    return (nseg & 7) ? bid : (bid & 7) * (nseg >> 3) + (bid >> 3);
}

// ---------------------------------------------------------------------------------------------
// k_init : bootstrap_filter  (particles.jl:87-105)    grid (nseg, ntheta)
// ---------------------------------------------------------------------------------------------
template <int MODEL, int THREADS, int NP>
__global__ __launch_bounds__(THREADS) void k_init(FilterView v, int nxt, double y) {
    constexpr int D = model_dim<MODEL>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* scr = (uint64_t*)smem;
    const int sb = logical_segment(blockIdx.x, v.nseg), th = blockIdx.y, tid = smc_tid();
    if (v.skip && v.skip[th]) return;   // workgroup-uniform
    const Params prm = v.params[th];
    const uint32_t stream = v.stream[th];
    const int64_t seg0 = (int64_t)sb * v.seg;
    double lw[NP][2];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pl = tid + k * THREADS;            // pair within segment
        const int64_t i0 = seg0 + 2 * pl;            // first particle of the pair
        const uint32_t pg = (uint32_t)(i0 >> 1);     // pair index within the filter
        double z[D][2], xn[2][D];
#pragma unroll
        for (int c = 0; c < D; ++c) box_muller(draw(v.seed, pg, stream, 0u, SLOT_NORMAL0 + c), z[c][0], z[c][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double zz[D];
#pragma unroll
            for (int c = 0; c < D; ++c) zz[c] = z[c][j];
            model_initial<MODEL>(prm, zz, xn[j]);
            const bool valid = (i0 + j) < v.n;
            lw[k][j] = valid ? model_logobs<MODEL>(prm, xn[j], y) : nan_mask();
        }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double2 o;
            o.x = (i0 < v.n) ? xn[0][c] : 0.0;
            o.y = (i0 + 1 < v.n) ? xn[1][c] : 0.0;
            store_out(reinterpret_cast<double2*>(v.x[nxt] + ((size_t)c * v.ntheta + th) * v.npad + i0), o);
        }
        if (v.anc) {
            int2 o;
            o.x = (int)i0;
            o.y = (int)i0 + 1;
            *reinterpret_cast<int2*>(v.anc + (size_t)th * v.npad + i0) = o;
        }
    }
    const SegRec rec = segment_epilogue<THREADS, NP>(v, nxt, th, sb, lw, scr);
    if (v.emit_now && v.nseg == 1 && tid == 0) emit_own(v, th, rec, 0u, true);
}

// ancestor segments of C a workgroup stages in LDS (160 KiB per CU: keep >= 2 workgroups resident)
__host__ __device__ constexpr int nstage_for(int seg) { return seg >= 8192 ? 1 : (seg >= 4096 ? 2 : 3); }
__host__ __device__ inline size_t step_lds_bytes(int nseg_p2, int threads, int np, bool multi) {
    const size_t base = (size_t)nseg_p2 * 16 + scr_words(threads, np) * 8;
    // single-segment filters stage their one segment too (the search then probes LDS, not global memory)
    return base + (size_t)(multi ? nstage_for(2 * np * threads) : 1) * lds_padded_len(2 * np * threads) * 8;
}

// ---------------------------------------------------------------------------------------------
// k_step : bootstrap_filter!  (particles.jl:107-129)   grid (nseg, ntheta), ONE launch per step
//   prologue  segment table of the weights in buffer `cur`  (+ emit logmu/ess of step t-1)
//   a = resample(weights)          -> two-level inverse CDF, one 64-bit draw per particle
//   xp = x[a]                      -> gather from buffer `cur`
//   x[i] = rand(transition(xp[i])); logw[i] = logpdf(observation(x[i]), y)
//   normalize(logw)                -> segment epilogue into buffer `cur^1`
// MULTI = the filter has more than one segment.
// ---------------------------------------------------------------------------------------------
// SYS = opt-in systematic resampling (SMC_FLAG_SYSTEMATIC): child j takes the point T_j = floor((j Dtot + v0) / n)
// instead of the multinomial targets between the block's break points.
// PERSIST: the body as one iteration of the persistent step kernel (k_persist below): every store another workgroup of the
// same launch reads (x, C, the segment record) is a write-through store; the loads are plain - the caller has polled the
// previous step's completion flags and made an agent-scope acquire before the call.
// GTAB: the segment table comes from global memory (v.tabD, v.tabsh; k_table built it before this launch and emitted the previous
// step): filters with more segments than the workgroup has threads.  No table in LDS, no emission here.
// RPT = 2: filters with more segments than threads, but at most twice as many: the window prologue with TWO records per thread
// (no k_table launch yet; the rare full table in LDS as ever).
template <int MODEL, int THREADS, int NP, bool MULTI, bool SYS, bool PERSIST, class VIEW, bool GTAB = false, int RPT = 1>
__device__ __forceinline__ void step_body(const VIEW& v, int cur, uint32_t t, int emit_prev, double yval, char* smem) {
    static_assert(!GTAB || (MULTI && !PERSIST), "global table: multi-segment launches");
    static_assert(RPT == 1 || (RPT == 2 && MULTI && !GTAB && !PERSIST), "two records per thread: multi-segment launches");
    constexpr int D = model_dim<MODEL>::value;
    constexpr int SEG = 2 * NP * THREADS;
    constexpr int NQ = 2 * NP;   // particles per thread
    constexpr int NSTAGE = nstage_for(SEG);
    constexpr int SEGP = lds_padded_len(SEG);   // padded length of a staged segment in LDS
    int sb_ = logical_segment(blockIdx.x, v.nseg), th_ = blockIdx.y, tid_ = smc_tid();
    if (PERSIST) {   // opaque per iteration: the loop around this body must not hoist everything derived from them into registers
        asm volatile("" : "+v"(tid_));
    }
    const int sb = sb_, th = th_, tid = tid_;
    if (!PERSIST && v.skip && v.skip[th]) return;   // workgroup-uniform
    const int nxt = cur ^ 1;
    const Params prm = v.params[th];
    const uint32_t stream = v.stream[th];
    const double y = v.y ? v.y[t] : yval;
    // particle indices fit 32 bits (smc_create: n_x <= 2^31): index arithmetic in 32 bits, 64 bits only in the addresses
    const uint32_t seg0 = (uint32_t)sb * (uint32_t)SEG, n32 = (uint32_t)v.n;
    const uint64_t* Cprev = v.C[cur] + (size_t)th * v.npad;
    const double* xprev = v.x[cur];
    uint64_t* scr;
#ifdef SMC_ABLATE
    if (SMC_ABL(v, 8) && (blockIdx.x & 8)) {   // experiment: stagger half of the workgroups by ~3.4 us
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
#endif
    SMC_STAMP(v, 0);
    SMC_PRIO(0);

    // Issue-early / use-late: every load whose address is known is issued BEFORE the random-number
    // work (Philox + Box-Muller is most of this kernel's VALU), which then runs under the latency.
    // (1) the break points of this workgroup's block of sorted uniforms (multinomial, MULTI)
    uint64_t F0 = 0, F1 = 0;
    if (MULTI && !SYS) {
        const uint64_t* Fb = v.brk + ((size_t)(t - v.brk_t0) * v.ntheta + th) * ((size_t)v.nseg + 1) + sb;
        F0 = Fb[0];
        F1 = Fb[1];
    }
    const TablePre tpre = (MULTI && !GTAB) ? table_preload<THREADS, VIEW, RPT>(v, cur, th, emit_prev && sb == 0) : TablePre{-inf(), 0, 0, 0, -inf(), 0};
    // (2) the 64-bit pick numbers of this thread's children
    uint64_t rr[NQ];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (SYS) { rr[2 * k] = rr[2 * k + 1] = 0; continue; }
        const uint32_t pg = (uint32_t)((seg0 >> 1) + tid + k * THREADS);
        const u32x4 rw = SMC_ABL(v, 3) ? u32x4{{pg * 2654435761u, pg ^ t, pg * 40503u, ~pg}} : draw(v.seed, pg, stream, t, SLOT_RESAMPLE);
        rr[2 * k] = ((uint64_t)rw.v[1] << 32) | rw.v[0];
        rr[2 * k + 1] = ((uint64_t)rw.v[3] << 32) | rw.v[2];
    }
    // this thread's child indices relative to the workgroup's first child (masked children j >= n take
    // the last real child's target: they are never stored as real particles)
    const uint32_t m_blk = (seg0 + SEG < n32 ? seg0 + SEG : n32) - seg0;   // children of this block (>= 1)
    const bool ragged = seg0 + SEG > n32;   // workgroup-uniform: only the filter's last block can hold masked children
    uint32_t kk[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) kk[i] = 2 * (tid + (i >> 1) * THREADS) + (i & 1);
    if (ragged) {   // a real branch (the asm statement keeps it from being flattened into selects every block would pay)
        asm volatile("; ragged");
#pragma unroll
        for (int i = 0; i < NQ; ++i) kk[i] = (seg0 + kk[i] < n32 ? seg0 + kk[i] : n32 - 1) - seg0;
    }
    uint64_t Tg[NQ];   // MULTI: the children's targets in table units; after the segment lookup, the in-segment thresholds

    // (3) speculative staging (NSTAGE = 3): children are sorted by block of the weight CDF, so the
    //     ancestors of the children at positions [sb*SEG, (sb+1)*SEG) usually sit in segments
    //     sb-1..sb+1.  Their loads are issued NOW; the range is checked once it is known (below).
    constexpr bool SPEC = MULTI && NSTAGE == 3;
    ulonglong2 stg[NSTAGE][NP];   // staging registers: NP 16-byte pieces per thread per staged segment
    int spec_lo = 0;
    if (SPEC) {
        spec_lo = sb - 1 < 0 ? 0 : sb - 1;
        spec_lo = spec_lo + NSTAGE > v.nseg ? (v.nseg - NSTAGE < 0 ? 0 : v.nseg - NSTAGE) : spec_lo;
#pragma unroll
        for (int sg = 0; sg < NSTAGE; ++sg) {
#pragma unroll
            for (int k = 0; k < NP; ++k) stg[sg][k] = ulonglong2{0, 0};
            if (spec_lo + sg < v.nseg) {
                const ulonglong2* src = reinterpret_cast<const ulonglong2*>(Cprev + (size_t)(spec_lo + sg) * SEG);
#pragma unroll
                for (int k = 0; k < NP; ++k) stg[sg][k] = src[tid + k * THREADS];
            }
        }
    }

    ulonglong2 stg1[NP];   // single segment: the filter's whole C, staged like a multi-segment window of one
    constexpr bool EARLY1 = D == 1;   // (with three state coordinates the early copy only spills to scratch)
#pragma unroll
    for (int k = 0; k < NP; ++k)   // (every element defined on every path: the array otherwise stays a stack object - 48 B of scratch)
        stg1[k] = (!MULTI && EARLY1) ? reinterpret_cast<const ulonglong2*>(Cprev)[tid + k * THREADS] : ulonglong2{0, 0};

    // ---- the state normals of this thread's children (under the loads of the records, the break points and the staged segments) ----------------
    // (three state coordinates, two pairs per thread: twelve normals held across the table, the targets and the search push the
    //  kernel to 183 vector registers - ONE workgroup per CU; they are computed next to their use instead, below)
    constexpr bool LATE_Z = D == 3 && NP >= 2 && !SYS;
    double z[NP][D][2];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (LATE_Z) break;
        const uint32_t pg = (uint32_t)((seg0 >> 1) + tid + k * THREADS);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            if (SMC_ABL(v, 2)) { z[k][c][0] = 1e-3 * (double)(pg & 1023); z[k][c][1] = -z[k][c][0]; }
            else if (SMC_ABL(v, 6)) { const u32x4 w4 = draw(v.seed, pg, stream, t, SLOT_NORMAL0 + c); z[k][c][0] = 1e-9 * (double)w4.v[0]; z[k][c][1] = 1e-9 * (double)w4.v[2]; }
            else box_muller(draw(v.seed, pg, stream, t, SLOT_NORMAL0 + c), z[k][c][0], z[k][c][1]);
            // systematic: keep the normals HERE, under the load latencies.  multinomial: measured faster when the
            // compiler sinks them next to their use, where they fill the waits of the LDS search
            if (SYS || SMC_NORMALS_EARLY) asm volatile("" : "+v"(z[k][c][0]), "+v"(z[k][c][1]));
        }
    }


    // alive = some weight is positive; otherwise the filter collapsed and ancestors are the identity
    uint64_t alive;
    uint64_t Sseg[NQ];
    int bseg[NQ];
    uint64_t* Cst = nullptr;   // staged segments [NSTAGE][SEG] (MULTI)
    int blo = 0;
    const int tab_p2 = GTAB ? 0 : v.nseg_p2;   // table entries in LDS
    if (MULTI) {
        TableLds L = carve(smem, tab_p2);
        if (GTAB) {
            L.Dcum = v.tabD + (size_t)th * v.nseg_p2;
            L.sh = v.tabsh + (size_t)th * v.nseg_p2;
        }
        scr = L.scr;
        // systematic: the step's one uniform, drawn by ONE thread (75 VALU instructions the other waves do
        // not spend) and published through LDS across the barriers of the table prologue
        uint64_t* sysw = L.scr + scr_words(THREADS, NP) - 8;   // tail words nobody else uses
        if (SYS && tid == 0) {
            const u32x4 uw = draw(v.seed, 0u, stream, t, SLOT_SYS);
            sysw[0] = ((uint64_t)uw.v[1] << 32) | uw.v[0];
        }
        // What this workgroup needs of the segment table of the weights being resampled.  Usual case: only its speculative
        // window (window_prologue: no table in LDS).  The workgroup that emits (logmu, ess) of the previous step, filters with
        // more segments than threads, and windows that turn out too narrow build the whole table (table_prologue).
        // emit_prev = 2: the records of the previous step carry no sum of squares, (logmu, ess = 0) follow from the totals every
        // workgroup computes anyway - the emitting workgroup then is no slower than the others (the full table made it, and the
        // workgroup sharing its CU, end 0.75 us after everybody else: the whole launch waited for them)
        const bool emitter = emit_prev && sb == 0;
        const bool emit_totals = emitter && emit_prev == 2;
        const int s_hi = spec_lo + NSTAGE - 1 < v.nseg - 1 ? spec_lo + NSTAGE - 1 : v.nseg - 1;
        const bool use_fast = SPEC && (GTAB || ((!emitter || emit_totals) && v.nseg_p2 <= RPT * THREADS));      // workgroup-uniform
        uint64_t P0 = 0, Dc[NSTAGE];
        int shw[NSTAGE];
        if (GTAB) {   // the table exists: the total and, for the speculative window, its entries (nseg > THREADS >= NSTAGE)
            alive = L.Dcum[v.nseg - 1];
            if (SPEC) {
                P0 = spec_lo ? L.Dcum[spec_lo - 1] : 0;
#pragma unroll
                for (int i = 0; i < NSTAGE; ++i) { Dc[i] = L.Dcum[spec_lo + i]; shw[i] = L.sh[spec_lo + i]; }
            }
            if (SYS) __syncthreads();   // (the step's uniform in LDS: the prologues' barriers otherwise)
        } else if (use_fast) {
            double Kw;
            alive = window_prologue<THREADS, NSTAGE, VIEW, RPT>(v, L.scr, tpre, spec_lo, P0, Dc, shw, Kw);
            if (emit_totals && tid == 0) emit_from_totals(v, th, Kw, alive, t == 1u, t - 1u);
        } else {
            alive = table_prologue<THREADS>(v, cur, th, L, emitter, t == 1u, t - 1u, RPT == 1 ? &tpre : nullptr);
        }
        SMC_STAMP(v, 1);
        SMC_PRIO(1);
        // targets of the first and last child of the block, in table units.  multinomial: the block's n
        // uniforms lie between its break points; systematic: T_j = floor((j Dtot + v0) / n)
        SysBase sbase{};
        uint64_t Tfirst, Tlast, lo_ = 0;
        if (SYS) {
            sbase = sys_base(alive, n32, v.inv_n, sysw[0], (uint64_t)seg0);
            Tfirst = sys_target(sbase, 0u);
            Tlast = sys_target(sbase, m_blk - 1);
        } else {
            mul64wide(F0, alive, Tfirst, lo_);
            mul64wide(F1, alive, Tlast, lo_);
        }
        // targets lie in [Tfirst, Tlast]: the workgroup's ancestors are the segments [b_lo, b_hi] of these two.
        // Usual case: both inside the speculative window.
        bool fast = false;
        if (use_fast) {
            fast = (spec_lo == 0 || P0 <= Tfirst) && Tlast < Dc[s_hi - spec_lo];
            if (!fast && !GTAB) alive = table_prologue<THREADS>(v, cur, th, L, false, false, 0u, RPT == 1 ? &tpre : nullptr);   // rare: very uneven weights
        }
        int b_lo = 0, b_hi = 0;
        if (fast || (SPEC && !use_fast && (spec_lo == 0 || L.Dcum[spec_lo - 1] <= Tfirst) && Tlast < L.Dcum[s_hi])) {
            b_lo = spec_lo;      // a superset of the true range is as good: children search inside it
            b_hi = s_hi;
        } else {
            for (int s = v.nseg_p2 >> 1; s >= 1; s >>= 1) {
                b_lo += (L.Dcum[b_lo + s - 1] <= Tfirst) ? s : 0;
                b_hi += (L.Dcum[b_hi + s - 1] <= Tlast) ? s : 0;
            }
            b_lo = b_lo < v.nseg ? b_lo : v.nseg - 1;
            b_hi = b_hi < v.nseg ? b_hi : v.nseg - 1;
            b_hi = b_hi < b_lo ? b_lo : b_hi;
        }
        int w0 = 1;
        while (w0 < b_hi - b_lo + 1) w0 <<= 1;
        Cst = (uint64_t*)(smem + (size_t)tab_p2 * 16 + scr_words(THREADS, NP) * 8);
        // the staged window: the speculative one if it covers the range, else (re)load from b_lo -
        // 16 B per lane, coalesced; the data lands in LDS after the normals have been computed
        const bool spec_ok = SPEC && b_lo >= spec_lo && b_hi < spec_lo + NSTAGE;
        int st_lo = spec_lo;
        if (!spec_ok) {   // workgroup-uniform
            st_lo = b_lo;
            const int nst = (b_hi - b_lo + 1) < NSTAGE ? (b_hi - b_lo + 1) : NSTAGE;
#pragma unroll
            for (int sg = 0; sg < NSTAGE; ++sg) {
#pragma unroll
                for (int k = 0; k < NP; ++k) stg[sg][k] = ulonglong2{0, 0};
                if (sg < nst) {   // only the segments of the range cost memory traffic
                    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(Cprev + (size_t)(b_lo + sg) * SEG);
#pragma unroll
                    for (int k = 0; k < NP; ++k) stg[sg][k] = src[tid + k * THREADS];
                }
            }
        }
        blo = st_lo;
        int pos[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (SYS) {
                Tg[i] = sys_target(sbase, kk[i]);
            } else {   // the block's largest uniform is its break point; the others are iid below it
                uint64_t pick;
#if SMC_EXP_PICKF64
                pick = (uint64_t)(fma((double)(uint32_t)(rr[i] >> 32), 0x1p-32, (double)((uint32_t)rr[i] >> 11) * 0x1p-53) * (double)(Tlast - Tfirst));
#else
                mul64wide(rr[i], Tlast - Tfirst, pick, lo_);
#endif
                Tg[i] = kk[i] == m_blk - 1 ? Tlast : Tfirst + pick;
            }
            pos[i] = b_lo;
        }
        if (fast) {   // the window's three entries sit in registers: two comparisons place a child
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                int r = 0;
                uint64_t base = P0;
                int shc = shw[0];
#pragma unroll
                for (int k = 0; k + 1 < NSTAGE; ++k) {
                    const bool up = Dc[k] <= Tg[i];
                    r += up ? 1 : 0;
                    base = up ? Dc[k] : base;
                    shc = up ? shw[k + 1] : shc;
                }
                bseg[i] = spec_lo + r;
                Sseg[i] = 0;
                // (C >> sh) > T - base  <=>  C > sys_threshold(T - base, sh); the chosen segment holds mass (its table entry is
                // positive, or the child would have passed it), so its shift is below 64 and the formula needs no guard
                Tg[i] = ((Tg[i] - base + 1) << shc) - 1;
            }
        } else {
            for (int s = w0 >> 1; s >= 1; s >>= 1) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int c = pos[i] + s;
                    pos[i] = (c <= b_hi && L.Dcum[c - 1] <= Tg[i]) ? c : pos[i];
                }
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                bseg[i] = pos[i];
                Sseg[i] = 0;
                const uint64_t base = bseg[i] ? L.Dcum[bseg[i] - 1] : 0;
                Tg[i] = sys_threshold(Tg[i] - base, L.sh[bseg[i]]);   // (C >> sh) > T - base  <=>  C > threshold
            }
        }
    } else {
        const TableLds L = carve(smem, v.nseg_p2);
        scr = L.scr;
        if (emit_prev && sb == 0) table_prologue<THREADS>(v, cur, th, L, true, t == 1u, t - 1u);
        alive = v.segS[cur][(size_t)th * v.nseg];
#pragma unroll
        for (int i = 0; i < NQ; ++i) { bseg[i] = 0; Sseg[i] = alive; }
        if (SYS) {   // one segment: K = kb, shift = SH, table = (S >> SH)
            const int sh = seg_shift(0.0, 0.0, v.SH);
            const uint64_t Dtot = seg_Q(alive, sh);
            const u32x4 uw = draw(v.seed, 0u, stream, t, SLOT_SYS);
            const SysBase sbase = sys_base(Dtot, n32, v.inv_n, ((uint64_t)uw.v[1] << 32) | uw.v[0], 0u);
#pragma unroll
            for (int i = 0; i < NQ; ++i) Tg[i] = sys_threshold(sys_target(sbase, kk[i]), sh);
            alive = Dtot;
        }
    }

    SMC_STAMP(v, 2);
    SMC_STAMP(v, 3);
    // ---- a = resample(weights), level 2: iid pick inside the child's segment -------------------
    uint64_t T2[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        uint64_t lo;
        if (SYS || MULTI) T2[i] = Tg[i];
        else mul64wide(rr[i], Sseg[i], T2[i], lo);
    }
    if (MULTI) {
#pragma unroll
        for (int sg = 0; sg < NSTAGE; ++sg)
#pragma unroll
            for (int k = 0; k < NP; ++k)
                *reinterpret_cast<ulonglong2*>(Cst + sg * SEGP + lds_pad(2 * (tid + k * THREADS))) = stg[sg][k];
        __syncthreads();   // staged segments visible
    } else {
        Cst = (uint64_t*)(smem + (size_t)v.nseg_p2 * 16 + scr_words(THREADS, NP) * 8);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const ulonglong2 c2 = EARLY1 ? stg1[k] : reinterpret_cast<const ulonglong2*>(Cprev)[tid + k * THREADS];
            *reinterpret_cast<ulonglong2*>(Cst + lds_pad(2 * (tid + k * THREADS))) = c2;
        }
        __syncthreads();
    }
    SMC_STAMP(v, 4);
    SMC_PRIO(2);
    int pos[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) pos[i] = 0;
    if (SMC_ABL(v, 0)) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) pos[i] = (int)(T2[i] & (SEG - 1));
    } else if (MULTI) {
        // branch-free search in the staged copy (LDS); lanes whose segment is not staged keep a
        // harmless in-range index and redo the search in global memory below
        bool far = false;
        // pb carries the padded position as an LDS byte pointer: a probe is then one ds_read_b64
        // with an immediate offset, no per-level address arithmetic
        lds_byte* pb[NQ];
        lds_byte* pb0[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int r = bseg[i] - blo;
            far |= r >= NSTAGE;
            pb0[i] = lds_ptr(Cst + (r < NSTAGE ? r : 0) * SEGP);
            pb[i] = pb0[i];
        }
#pragma unroll
        for (int s = SEG >> 1; s >= 1; s >>= 1) {
            uint64_t val[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) val[i] = lds_load_u64(pb[i] + 8 * lds_probe_off(s));
#pragma unroll
            for (int i = 0; i < NQ; ++i) pb[i] += (val[i] <= T2[i]) ? 8 * lds_step_inc(s) : 0;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) pos[i] = lds_unpad((int)(pb[i] - pb0[i]) >> 3);
        if (__builtin_amdgcn_ballot_w64(far)) {   // rare: very uneven weights spread a workgroup over many segments
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (bseg[i] - blo >= NSTAGE) {
                    const uint64_t* Cb = Cprev + (size_t)bseg[i] * SEG;
                    int pp = 0;
                    for (int s = SEG >> 1; s >= 1; s >>= 1) pp += (Cb[pp + s - 1] <= T2[i]) ? s : 0;
                    pos[i] = pp;
                }
            }
        }
    } else {
        lds_byte* pb[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) pb[i] = lds_ptr(Cst);
#pragma unroll
        for (int s = SEG >> 1; s >= 1; s >>= 1) {
            uint64_t val[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) val[i] = lds_load_u64(pb[i] + 8 * lds_probe_off(s));
#pragma unroll
            for (int i = 0; i < NQ; ++i) pb[i] += (val[i] <= T2[i]) ? 8 * lds_step_inc(s) : 0;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) pos[i] = lds_unpad((int)(pb[i] - lds_ptr(Cst)) >> 3);
    }
    SMC_STAMP(v, 5);
    uint32_t anc[NQ];
    double xp[NQ][D];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint32_t own = seg0 + 2 * (tid + (i >> 1) * THREADS) + (i & 1);
        uint32_t a = (uint32_t)bseg[i] * (uint32_t)SEG + (uint32_t)pos[i];
        if (!alive || ragged) {   // workgroup-uniform and rare
            asm volatile("; collapsed or ragged");
            a = alive ? a : own;                // collapsed filter: identity
            a = a < n32 ? a : n32 - 1;          // only masked children (j >= n) can land there
        }
        anc[i] = a;
#pragma unroll
        for (int c = 0; c < D; ++c) xp[i][c] = SMC_ABL(v, 1) ? 0.25 * (double)(a & 7) : xprev[((size_t)c * v.ntheta + th) * v.npad + a];
    }

    // ---- x[i] = rand(transition(xp[i])); logw[i] = logpdf(observation(x[i]), y) ---------------
    double lw[NP][2];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const uint32_t i0 = seg0 + 2 * (tid + k * THREADS);
        double xn[2][D];
        if (LATE_Z) {
            const uint32_t pg = (uint32_t)((seg0 >> 1) + tid + k * THREADS);
#pragma unroll
            for (int c = 0; c < D; ++c) box_muller(draw(v.seed, pg, stream, t, SLOT_NORMAL0 + c), z[k][c][0], z[k][c][1]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double zz[D];
#pragma unroll
            for (int c = 0; c < D; ++c) zz[c] = z[k][c][j];
            model_transition<MODEL>(prm, xp[2 * k + j], zz, xn[j]);
            lw[k][j] = model_logobs<MODEL>(prm, xn[j], y);
        }
        if (ragged) {   // masked children: NaN weight, zero state
            asm volatile("; ragged");
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (i0 + j >= n32) {
                    lw[k][j] = nan_mask();
#pragma unroll
                    for (int c = 0; c < D; ++c) xn[j][c] = 0.0;
                }
        }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double2 o;
            o.x = xn[0][c];
            o.y = xn[1][c];
            store_out<PERSIST>(reinterpret_cast<double2*>(v.x[nxt] + ((size_t)c * v.ntheta + th) * v.npad + i0), o);
        }
        if (v.anc) {
            int2 o;
            o.x = (int)anc[2 * k];
            o.y = (int)anc[2 * k + 1];
            *reinterpret_cast<int2*>(v.anc + (size_t)th * v.npad + i0) = o;
        }
    }
    SMC_STAMP(v, 6);
    SMC_PRIO(3);
    if (SMC_ABL(v, 4)) {   // keep lw alive, skip the normalisation
        double acc = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) acc += lw[k][0] + lw[k][1];
        if (acc == 1.2345) v.logZ[th] = acc;
        return;
    }
    const SegRec rec = segment_epilogue<THREADS, NP, PERSIST>(v, nxt, th, sb, lw, scr);
    if (!MULTI && v.emit_now && tid == 0) emit_own(v, th, rec, t, false);
    SMC_STAMP(v, 7);
}

template <int MODEL, int THREADS, int NP, bool MULTI, bool SYS = false, bool GTAB = false, int RPT = 1>
__global__ __launch_bounds__(THREADS) void k_step(FilterView v, int cur, uint32_t t, int emit_prev, double yval) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    step_body<MODEL, THREADS, NP, MULTI, SYS, false, FilterView, GTAB, RPT>(v, cur, t, emit_prev, yval, smem);
}

// ---------------------------------------------------------------------------------------------
// k_persist : the steps [t0, t1) of log_likelihood's loop (particles.jl:141-144) for a multi-segment filter in ONE launch -
// OPT-IN (SMC_PERSIST=1), measured SLOWER than one launch per step on MI355X ("persistent step kernel"):
// kept as the measured form of that negative result.  grid (nseg, ntheta), every workgroup resident (the host checks the
// occupancy; every spin is bounded).  Between two steps: the storing waves drain their write-through stores, a barrier, ONE
// lane publishes the workgroup's completion flag (tag = next step); thread i of every workgroup of the filter polls the flag
// of segment i (relaxed sc1 loads, s_sleep), one lane's agent-scope acquire, a barrier - then the step's plain loads.
// ---------------------------------------------------------------------------------------------
struct PersistCtl {
    unsigned* flags[2];   // [ntheta * nseg] per buffer: t + 1 once step t of that segment is complete and visible
    int* err;             // pinned host word: != 0 when a spin expired (not every workgroup resident)
};
// (second launch bound: as many waves per SIMD as two workgroups per CU need - the loop otherwise hoists its invariants into 225 registers)
template <int MODEL, int THREADS, int NP>
__global__ __launch_bounds__(THREADS, (THREADS >= 512 ? 4 : (THREADS >= 256 ? 4 : 2))) void k_persist(FilterView v_arg, int cur, uint32_t t0, uint32_t t1, PersistCtl pc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    // The view is read IN PLACE from the kernel-argument segment at every step, through a pointer the compiler cannot see through:
    // everything derived from it (Philox key schedule, base addresses, flags) is then loaded and computed where the step uses it, as
    // in k_step - hoisted out of the loop it does not fit the scalar registers (110 spilled lanes, 45 spilled vector registers).
    typedef __attribute__((address_space(4))) const char karg_byte;
    karg_byte* ka = (karg_byte*)__builtin_amdgcn_kernarg_segment_ptr();   // v_arg is the first argument: offset 0
    // a word of the LDS scratch nobody else uses (the last tail word: the systematic kernels' only): "a spin of this workgroup expired"
    int* expired_flag = (int*)((uint64_t*)(smem + (size_t)v_arg.nseg_p2 * 16) + scr_words(THREADS, NP) - 1);
    if (tid == 0) *expired_flag = 0;
    // diagnostic builds (FilterView::dbg, allocated by the profiling build's smc_create only): where a workgroup's time goes -
    // accumulated in four more of those LDS words (wait, body, publish, last stamp), not in registers the step needs
    unsigned long long* acc = (unsigned long long*)((uint64_t*)(smem + (size_t)v_arg.nseg_p2 * 16) + scr_words(THREADS, NP) - 5);
    if (v_arg.dbg && tid == 0) { acc[0] = acc[1] = acc[2] = 0; acc[3] = __builtin_amdgcn_s_memrealtime(); }
    for (uint32_t t = t0; t < t1; ++t) {
        asm volatile("" : "+s"(ka));
        typedef __attribute__((address_space(4))) const FilterView karg_view;
        karg_view& v = *(karg_view*)ka;
        const int sb = logical_segment(blockIdx.x, v.nseg), th = blockIdx.y;
        if (t > t0) {
            if (tid < v.nseg) {
                const unsigned* f = pc.flags[cur] + (size_t)th * v.nseg + tid;
                const unsigned long long start = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t) {
                    if (__builtin_amdgcn_s_memrealtime() - start > 10000000ull) { *expired_flag = 1; break; }   // 100 ms: give up
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (*expired_flag) {   // workgroup-uniform (read behind the barrier): this workgroup leaves; the others' spins expire in turn
                if (tid == 0) *pc.err = 1;
                return;
            }
        }
        if (v.dbg && tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); acc[0] += now - acc[3]; acc[3] = now; }
        step_body<MODEL, THREADS, NP, true, false, true>(v, cur, t, 2, 0.0, smem);
        if (v.dbg && tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); acc[1] += now - acc[3]; acc[3] = now; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) __hip_atomic_store(pc.flags[cur ^ 1] + (size_t)th * v.nseg + sb, t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v.dbg && tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); acc[2] += now - acc[3]; acc[3] = now; }
        cur ^= 1;
    }
    if (v_arg.dbg && tid == 0) {   // [steps, wait (poll + acquire + barrier), step body, drain + barrier + publish] in 10 ns units; marker in word 7
        unsigned long long* d = v_arg.dbg + ((size_t)gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;   // second half of the buffer
        d[0] = t1 - t0; d[1] = acc[0]; d[2] = acc[1]; d[3] = acc[2]; d[7] = 0x5045525349535421ull;
    }
}

// ---------------------------------------------------------------------------------------------
// k_finalize : (logmu, ess) of the weights currently in buffer `cur`.   grid (ntheta)
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_finalize(FilterView v, int cur, int first_emit, uint32_t t_emit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (v.skip && v.skip[blockIdx.x]) {   // a filter that was not run: logZ = -inf
        if (threadIdx.x == 0) {
            v.logZ[blockIdx.x] = -inf();
            if (v.host_out) v.host_out[blockIdx.x] = -inf();
        }
        return;
    }
    const TableLds L = carve(smem, v.nseg_p2);
    table_prologue<THREADS>(v, cur, blockIdx.x, L, true, first_emit != 0, t_emit);
}

// ---------------------------------------------------------------------------------------------
// k_table : the segment table of the weights in buffer `cur` into global memory (v.tabD, v.tabsh), ONCE per step, for filters
// with more segments than a workgroup of k_step has threads (k_step<..., GTAB> then reads its window and the total instead of
// every workgroup rebuilding the table from all records: nseg^2 record reads per step and 16 B of LDS per segment in every
// workgroup - the reason filters beyond 2^20 particles ran at half the rate).  The same integers table_prologue puts in LDS.
// emit: 0 nothing; 1 (logmu, ess) of these weights from the full records; 2 (logmu, ess = 0) from the totals (the records carry
// no sum of squares).  grid (ntheta)
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_table(FilterView v, int cur, int emit, int first_emit, uint32_t t_emit) {
    constexpr int NW = THREADS / WAVE;
    constexpr int DEADK = (int)0x80000000;
    __shared__ int redk[NW];
    __shared__ uint64_t wt[NW], wr[NW];
    const int th = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    if (v.skip && v.skip[th]) return;
    const size_t base = (size_t)th * v.nseg;
    const double* sk = v.segk[cur] + base;
    const uint64_t* sS = v.segS[cur] + base;
    uint64_t* Dcum = v.tabD + (size_t)th * v.nseg_p2;
    int* shv = v.tabsh + (size_t)th * v.nseg_p2;
    // the exponent maximum: a coalesced sweep over the records
    int ki = DEADK;
    for (int b = tid; b < v.nseg; b += THREADS) {
        const double k = sk[b];
        const int kb = k == -inf() ? DEADK : (int)k;
        ki = kb > ki ? kb : ki;
    }
    ki = wave_max_i32(ki);
    if (lane == 0) redk[wave] = ki;
    __syncthreads();
    int km = redk[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) km = redk[w] > km ? redk[w] : km;
    const double K = km == DEADK ? -inf() : (double)km;
    // wave w owns the 64 E consecutive entries from w 64 E on, lane l the entries l, l + 64, ... of them: every load and store of a
    // wave is one contiguous run of records; E short scans with a carried total; the inclusive sums stay in registers until
    // the offsets of the waves are known (stores only: nothing of the table is read back)
    constexpr int EMAX = 16;   // nseg_p2 <= 16384
    const int E = v.nseg_p2 >= THREADS ? v.nseg_p2 / THREADS : 1, w0 = wave * E * WAVE;
    uint64_t inc[EMAX], carry = 0, rsum = 0;
    int she[EMAX];
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        inc[i] = 0; she[i] = 64;
        if (i < E) {   // wave-uniform
            const int b = w0 + i * WAVE + lane;
            uint64_t Q = 0;
            if (b < v.nseg) {
                she[i] = seg_shift(K, sk[b], v.SH);
                Q = seg_Q(sS[b], she[i]);
                if (emit == 1) rsum += seg_R(v.segS2hi[cur][base + b], v.segS2lo[cur][base + b], she[i], v.SH);
            }
            inc[i] = wave_incl_scan(Q, lane) + carry;
            carry = readlane_u64(inc[i], WAVE - 1);
        }
    }
    if (lane == 0) wt[wave] = carry;
    if (emit == 1) {
        const uint64_t rw = wave_sum(rsum);
        if (lane == 0) wr[wave] = rw;
    }
    __syncthreads();
    uint64_t off = 0, Dtot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { off += w < wave ? wt[w] : 0; Dtot += wt[w]; }
#pragma unroll
    for (int i = 0; i < EMAX; ++i)
        if (i < E) {
            const int b = w0 + i * WAVE + lane;
            if (b < v.nseg_p2) { Dcum[b] = off + inc[i]; shv[b] = she[i]; }
        }
    if (emit == 1 && tid == 0) {   // (logmu, ess) of these weights: what table_prologue's emitting thread does
        uint64_t Rtot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) Rtot += wr[w];
        double logmu, ess;
        combine_outputs(K, Dtot, Rtot, v.SH, v.n, logmu, ess);
        v.last_logmu[th] = logmu;
        v.last_ess[th] = ess;
        v.last_K[th] = K;
        v.last_D[th] = Dtot;
        if (v.trace_logmu) v.trace_logmu[(size_t)t_emit * v.ntheta + th] = logmu;
        if (v.trace_ess) v.trace_ess[(size_t)t_emit * v.ntheta + th] = ess;
        const double z = first_emit ? logmu : v.logZ[th] + logmu;
        v.logZ[th] = z;
        host_emit(v, th, z, logmu, ess);
    }
    if (emit == 2 && tid == 0) emit_from_totals(v, th, K, Dtot, first_emit != 0, t_emit);
}

}  // namespace smc
