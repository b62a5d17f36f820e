// Diagnostic: issue rate of the VALU instructions the filters are made of, on gfx950.
// Every test runs ITER x 8 independent chains of one instruction per lane, 8 waves per SIMD on every CU, and prints the
// time per wave-instruction per SIMD relative to v_xor_b32 (= one full-rate slot).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate scripts/dbg/valu_rate.hip && /tmp/valu_rate
// Measured on an MI355X of this pool (one slot = 1.076 ns per wave-instruction per SIMD, i.e. 2.5 cycles at 2.3-2.4 GHz):
//   1.0   v_add_u32 v_sub_u32 v_xor/and/or_b32 v_bitop3_b32 v_mov_b32 v_lshrrev_b32 v_fma/mul/add_f32
//   1.6-1.9  v_fma_f64 1.88 v_mul_f64 1.85 v_add_f64 1.7 v_mad_u64_u32 1.72 v_mul_lo_u32 1.8 v_mul_hi_u32 1.67 v_cndmask_b32 1.63
//         v_cmp_* 1.67 v_lshlrev_b32 1.6 v_lshl_add_u32/u64 1.7 v_lshl/lshrrev_b64 1.65 v_add3_u32 v_max/min_i32 v_cvt_* v_ldexp_f64
//         v_mov_b64 v_readlane_b32 and every DPP form (v_mov_b32_dpp, v_add_u32_dpp) 1.65
//   3.2   v_exp_f32        6.3   v_rcp_f64 v_rsq_f64 v_sqrt_f64 v_trig_preop_f64
//   (v_cndmask_b32 with a VCC nobody has written in the kernel shows 8.8: an artefact of this test, 1.6 after a v_cmp)
// scripts/dbg/isa_cost.py weights an ISA listing with this table: the loop of k_resident (LG) came to 978 slots per wave and
// step = 4.2 us at four waves per SIMD, the measured time per step - the batched kernel is bound by VALU issue, nothing else.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096

#define CHAIN32(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {                   \
        uint32_t r[8];                                                                                       \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;                                                  \
        for (int it = 0; it < ITER; ++it) {                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b));   \
        }                                                                                                    \
        uint32_t s = 0;                                                                                      \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                               \
        if (s == 0x12345) out[0] = s;                                                                        \
    }

#define CHAIN64(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {                   \
        double r[8];                                                                                         \
        const double da = 1.0 + 1e-9 * a, db = 1e-9 * b;                                                     \
        for (int i = 0; i < 8; ++i) r[i] = 1.0 + 1e-3 * (threadIdx.x + i);                                   \
        for (int it = 0; it < ITER; ++it) {                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(da), "v"(db)); \
        }                                                                                                    \
        double s = 0;                                                                                        \
        for (int i = 0; i < 8; ++i) s += r[i];                                                               \
        if (s == 0.12345) out[0] = 1;                                                                        \
    }

#define CHAINMAD(NAME, ASM)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t a, uint32_t b) {                   \
        uint64_t r[8];                                                                                       \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;                                                  \
        for (int it = 0; it < ITER; ++it) {                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b));   \
        }                                                                                                    \
        uint64_t s = 0;                                                                                      \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                               \
        if (s == 0x12345) out[0] = (uint32_t)s;                                                              \
    }

CHAIN32(t_xor, "v_xor_b32 %0, %0, %1")
CHAIN32(t_add, "v_add_u32 %0, %0, %1")
CHAIN32(t_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
CHAIN32(t_mul_lo, "v_mul_lo_u32 %0, %0, %1")
CHAIN32(t_mul_hi, "v_mul_hi_u32 %0, %0, %1")
CHAIN32(t_mul_u24, "v_mul_u32_u24 %0, %0, %1")
CHAIN32(t_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
CHAIN32(t_lshl, "v_lshlrev_b32 %0, 3, %0")
CHAIN32(t_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
CHAIN32(t_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
CHAIN32(t_perm, "v_perm_b32 %0, %0, %1, %2")
CHAIN32(t_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
CHAIN32(t_fma32, "v_fma_f32 %0, %0, %1, %2")
CHAINMAD(t_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
CHAINMAD(t_lshl64, "v_lshlrev_b64 %0, 3, %0")
CHAINMAD(t_pk_fma32, "v_pk_fma_f32 %0, %0, %0, %0")
CHAIN64(t_fma64, "v_fma_f64 %0, %0, %1, %2")
CHAIN64(t_mul64, "v_mul_f64 %0, %0, %1")
CHAIN64(t_add64, "v_add_f64 %0, %0, %2")
CHAIN64(t_ldexp64, "v_ldexp_f64 %0, %0, 1")
CHAIN64(t_rcp64, "v_rcp_f64 %0, %0")
CHAIN64(t_rsq64, "v_rsq_f64 %0, %0")
CHAIN64(t_sqrt64, "v_sqrt_f64 %0, %0")
CHAIN64(t_fract64, "v_fract_f64 %0, %0")
CHAIN64(t_rndne64, "v_rndne_f64 %0, %0")
CHAIN64(t_max64, "v_max_f64 %0, %0, %1")
CHAIN64(t_divfix64, "v_div_fixup_f64 %0, %0, %1, %2")
CHAIN64(t_divfmas64, "v_div_fmas_f64 %0, %0, %1, %2")
CHAIN64(t_cmp64, "v_cmp_lt_f64 vcc, %0, %1")
CHAIN64(t_mov64, "v_mov_b64 %0, %1")
CHAIN64(t_fmac64, "v_fmac_f64 %0, %1, %2")
CHAIN64(t_cmp64_s, "v_cmp_lt_f64 s[20:21], %0, %1")
CHAIN64(t_cmpu64, "v_cmp_lt_u64 vcc, %0, %1")
CHAIN64(t_class64, "v_cmp_class_f64 vcc, %0, 3")
CHAIN64(t_trig_preop, "v_trig_preop_f64 %0, %0, 1")
CHAIN64(t_frexp_mant, "v_frexp_mant_f64 %0, %0")
CHAIN32(t_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
CHAIN32(t_cndmask_c01, "v_cndmask_b32_e64 %0, 0, 1, vcc")
CHAIN32(t_cndmask_other, "v_cndmask_b32 %0, %1, %2, vcc")
CHAIN32(t_cmp32, "v_cmp_lt_u32 vcc, %0, %1")
CHAIN32(t_cmp32_s, "v_cmp_lt_u32 s[20:21], %0, %1")
CHAIN32(t_cmp_cnd, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %2, vcc")
CHAIN32(t_mov32, "v_mov_b32 %0, %1")
CHAIN32(t_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CHAIN32(t_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CHAIN32(t_add3, "v_add3_u32 %0, %0, %1, %2")
CHAIN32(t_and, "v_and_b32 %0, %0, %1")
CHAIN32(t_or, "v_or_b32 %0, %0, %1")
CHAIN32(t_lshr, "v_lshrrev_b32 %0, 3, %0")
CHAIN32(t_sub, "v_sub_u32 %0, %0, %1")
CHAIN32(t_max_i32, "v_max_i32 %0, %0, %1")
CHAIN32(t_min_u32, "v_min_u32 %0, %0, %1")
CHAIN32(t_add_co, "v_add_co_u32 %0, vcc, %0, %1")
CHAIN32(t_bfe, "v_bfe_u32 %0, %0, 3, 8")
CHAIN32(t_and_or, "v_and_or_b32 %0, %0, %1, %2")
CHAIN32(t_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
CHAIN32(t_xad, "v_xad_u32 %0, %0, %1, %2")
CHAIN32(t_readlane, "v_readlane_b32 s20, %0, 3")
CHAIN32(t_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
CHAIN32(t_mul_f32, "v_mul_f32 %0, %0, %1")
CHAIN32(t_add_f32, "v_add_f32 %0, %0, %1")
CHAIN32(t_exp_f32, "v_exp_f32 %0, %0")
CHAIN32(t_mul_hi_i32, "v_mul_hi_i32 %0, %0, %1")
CHAINMAD(t_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
CHAINMAD(t_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
CHAINMAD(t_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
CHAINMAD(t_lshr64, "v_lshrrev_b64 %0, 3, %0")
CHAINMAD(t_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %2, %0")

template <class K> static double run(K k, uint32_t* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // 256 CUs x 8 workgroups of 256 threads = 8 waves per SIMD
    k<<<256 * 8, 256>>>(out, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 4; ++i) k<<<256 * 8, 256>>>(out, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 4;
}

int main() {
    uint32_t* out; hipMalloc(&out, 64);
    const double base = run(t_xor, out);
    // waves per SIMD = 8, instructions per wave = ITER*8
    const double ns_per = base * 1e6 / (8.0 * ITER * 8);
    printf("v_xor_b32: %.3f ms  = %.3f ns per wave-instruction per SIMD (4 cycles at %.2f GHz)\n", base, ns_per, 4.0 / ns_per);
#define T(NAME) printf("%-16s %.2fx\n", #NAME, run(NAME, out) / base);
    T(t_add) T(t_bitop3) T(t_mul_lo) T(t_mul_hi) T(t_mul_u24) T(t_mad_u24) T(t_lshl) T(t_cndmask) T(t_lshl_add) T(t_perm) T(t_alignbit) T(t_fma32)
    T(t_mad_u64_u32) T(t_lshl64) T(t_pk_fma32) T(t_fma64) T(t_mul64) T(t_add64) T(t_ldexp64) T(t_rcp64) T(t_rsq64) T(t_sqrt64) T(t_fract64) T(t_rndne64)
    T(t_max64) T(t_divfix64) T(t_divfmas64) T(t_cmp64) T(t_mov64)
    T(t_fmac64) T(t_cmp64_s) T(t_cmpu64) T(t_class64) T(t_trig_preop) T(t_frexp_mant)
    T(t_cndmask_e64) T(t_cndmask_c01) T(t_cndmask_other) T(t_cmp32) T(t_cmp32_s) T(t_cmp_cnd) T(t_mov32) T(t_mov_dpp) T(t_add_dpp) T(t_add3) T(t_and) T(t_or)
    T(t_lshr) T(t_sub) T(t_max_i32) T(t_min_u32) T(t_add_co) T(t_bfe) T(t_and_or) T(t_lshl_or) T(t_xad) T(t_readlane) T(t_cvt_f32_u32) T(t_mul_f32) T(t_add_f32) T(t_exp_f32) T(t_mul_hi_i32)
    T(t_add_u64) T(t_cvt_f64_u32) T(t_cvt_f64_i32) T(t_lshr64) T(t_mad_i64_i32)
    return 0;
}
