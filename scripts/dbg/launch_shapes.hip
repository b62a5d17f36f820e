// Diagnostic (never shipped): the period of back-to-back dependent launches of an empty kernel as a function of the launch
// shape (workgroups x threads, dynamic LDS, VGPRs do not matter for an empty kernel).
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/launch_shapes scripts/dbg/launch_shapes.hip ; run: /tmp/launch_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double a[36]; };
__global__ void k_empty(Big b, double* out) {
    extern __shared__ char smem[];
    if (b.a[0] == 123.456 && threadIdx.x == 0) out[blockIdx.x] = b.a[1] + smem[0];
}
int main() {
    double* d; (void)hipMalloc(&d, 64 << 20);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b{}; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int shapes[][3] = {{512, 512, 61}, {512, 512, 0}, {512, 256, 61}, {256, 1024, 61}, {256, 512, 61}, {1024, 256, 30}, {1024, 512, 30},
                             {2048, 256, 16}, {512, 64, 61}, {64, 512, 61}, {1, 64, 0}, {4096, 256, 16}};
    for (auto& sh : shapes) {
        const int N = 2000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), sh[2] * 1024, s, b, d);
            (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%5d workgroups x %4d threads, %2d KB LDS (%5d waves): %.2f us per launch\n", sh[0], sh[1], sh[2], sh[0] * sh[1] / 64, ms / N * 1e3);
    }
    return 0;
}
