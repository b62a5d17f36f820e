// Diagnostic only (never shipped): relative issue cost of the VALU instructions the step kernels use.
// One wave per SIMD (256-thread workgroups, 256 of them), 8 independent chains per instruction.
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rates scripts/dbg/valu_rates.hip ; run: /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// core clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, from workgroup 0
#define CLK_BEGIN const uint64_t c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define CLK_END                                                                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                                         \
        ((uint64_t*)out)[256] = __builtin_amdgcn_s_memtime() - c0_;                                    \
        ((uint64_t*)out)[257] = __builtin_amdgcn_s_memrealtime() - r0_;                                \
    }
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL32(NAME, ASM)                                                                  \
    __global__ void NAME(uint32_t* out, int iters, uint32_t seed) {                          \
        CLK_BEGIN                                                                            \
        uint32_t a[8], b = seed | 1u, c = seed * 3u + threadIdx.x;                            \
        for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i + seed;                            \
        for (int it = 0; it < iters; ++it) {                                                 \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
            }                                                                                \
        }                                                                                    \
        uint32_t s = 0;                                                                      \
        for (int i = 0; i < 8; ++i) s ^= a[i];                                                \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                           \
        CLK_END                                                                              \
    }
#define KERNEL64(NAME, ASM)                                                                  \
    __global__ void NAME(uint32_t* out, int iters, uint32_t seed) {                          \
        CLK_BEGIN                                                                            \
        uint64_t a[8], b = ((uint64_t)seed << 20) | 0x3ff0000000000001ull, c = 0x3ff0000000000003ull + threadIdx.x; \
        uint32_t b32 = seed | 1u, c32 = seed * 3u + threadIdx.x;                              \
        for (int i = 0; i < 8; ++i) a[i] = 0x3ff0000000000000ull + threadIdx.x + i + seed;   \
        for (int it = 0; it < iters; ++it) {                                                 \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "v"(b32), "v"(c32) : "vcc"); \
            }                                                                                \
        }                                                                                    \
        uint64_t s = 0;                                                                      \
        for (int i = 0; i < 8; ++i) s ^= a[i];                                                \
        if (s == 0x12345678u) out[threadIdx.x] = (uint32_t)s;                                 \
        CLK_END                                                                              \
    }
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_cnd, "v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cmp32, "v_cmp_gt_u32 vcc, %0, %1")
KERNEL32(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cnd_zero, "v_cndmask_b32_e64 %0, 0, %1, vcc")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %3, %4, %0")
KERNEL64(k_lshl_add64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(k_fma64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add64, "v_add_f64 %0, %0, %1")
KERNEL64(k_ldexp64, "v_ldexp_f64 %0, %0, %3")
KERNEL64(k_cmp64, "v_cmp_gt_u64 vcc, %0, %1")
KERNEL64(k_rcp64, "v_rcp_f64 %0, %0")
KERNEL64(k_rsq64, "v_rsq_f64 %0, %0")
KERNEL64(k_sqrt64, "v_sqrt_f64 %0, %0")
KERNEL64(k_cvt64, "v_cvt_f64_u32 %0, %3")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_mov64, "v_mov_b64 %0, %1")
KERNEL64(k_pkfma32, "v_pk_fma_f32 %0, %0, %1, %2")
typedef void (*kern_t)(uint32_t*, int, uint32_t);
static double g_mhz = 0;
static double run(kern_t k, uint32_t* out, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipLaunchKernelGGL(k, grid, block, 0, 0, out, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, grid, block, 0, 0, out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t clk[2];
    hipMemcpy(clk, (uint64_t*)out + 256, 16, hipMemcpyDeviceToHost);
    g_mhz = (double)clk[0] / (double)clk[1] * 100.0;
    return (double)ms * 1e6 / ((double)iters * 32.0 * waves_per_simd);   // ns per wave-instruction per SIMD
}
int main() {
    uint32_t* out;
    hipMalloc(&out, 8192);
    struct { const char* n; kern_t k; } ks[] = {
        {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_mul_lo_u32", k_mul_lo},
        {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24}, {"v_cndmask_b32", k_cndmask}, {"cmp+cndmask pair", k_cmp_cnd}, {"v_cmp_gt_u32", k_cmp32}, {"cndmask sgpr mask", k_cnd_sgpr}, {"cndmask 0,v,vcc", k_cnd_zero},
        {"v_mov_b32_dpp", k_mov_dpp}, {"v_add_u32_dpp", k_add_dpp}, {"v_lshl_add_u32", k_lshl_add},
        {"v_alignbit_b32", k_alignbit}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshl_add64},
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_ldexp_f64", k_ldexp64},
        {"v_cmp_gt_u64", k_cmp64}, {"v_rcp_f64", k_rcp64}, {"v_rsq_f64", k_rsq64}, {"v_sqrt_f64", k_sqrt64},
        {"v_cvt_f64_u32", k_cvt64}, {"v_lshrrev_b64", k_lshr64}, {"v_mov_b64", k_mov64}, {"v_pk_fma_f32", k_pkfma32}};
    double base1 = 0, base4 = 0;
    for (auto& e : ks) {
        const double t1 = run(e.k, out, 1), t4 = run(e.k, out, 4);
        if (base1 == 0) { base1 = t1; base4 = t4; }
        printf("%-16s  1 wave/SIMD %.3f ns (x%.2f)   4 waves/SIMD %.3f ns (x%.2f) = %.2f cycles at %.0f MHz\n", e.n, t1, t1 / base1, t4,
               t4 / base4, t4 * g_mhz * 1e-3, g_mhz);
    }
    return 0;
}
