// Diagnostic (never shipped): the period of back-to-back dependent launches of a kernel that does (almost) nothing, with
// the launch shape of k_step at C2 (512 workgroups x 512 threads, 61 KB dynamic LDS, ~300 B of kernel arguments).
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/launch_gap scripts/dbg/launch_gap.hip ; run: /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double a[36]; };
__global__ void k_empty(Big b, double* out) {
    extern __shared__ char smem[];
    if (b.a[0] == 123.456 && threadIdx.x == 0) out[blockIdx.x] = b.a[1] + smem[0];
}
__global__ void k_touch(Big b, double* out, int n) {   // writes 16 B per thread like the step kernel's outputs
    extern __shared__ char smem[];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 v{b.a[2] + i, b.a[3]};
    for (int r = 0; r < n; ++r) __builtin_nontemporal_store(v.x + r, &out[(r * (size_t)gridDim.x * blockDim.x + i) * 2]);
}
int main() {
    double* d; hipMalloc(&d, 64 << 20);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b{}; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 4; ++variant) {
        const int lds = variant == 1 ? 0 : 61 * 1024;
        const int N = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) {
                if (variant < 2) hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), lds, s, b, d);
                else hipLaunchKernelGGL(k_touch, dim3(512), dim3(512), lds, s, b, d, variant == 2 ? 1 : 8);
            }
            hipEventRecord(e1, s); hipStreamSynchronize(s);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("variant %d (%s, lds %d): %.2f us per launch\n", variant, variant < 2 ? "empty" : (variant == 2 ? "4 MB streamed stores" : "32 MB streamed stores"), lds, ms / N * 1e3);
    }
    return 0;
}
