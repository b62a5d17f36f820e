// Diagnostic (never shipped): can the dependent-launch floor be hidden?  A chain of "steps", each a launch of 512 workgroups x
// 512 threads with 61 KB of LDS whose workgroups do A microseconds of independent arithmetic, then need the PREVIOUS step to be
// complete, then do B microseconds of dependent arithmetic.
//   mode 0: one stream, the dependency is the stream order (what log_likelihood does today)
//   mode 2: ONE stream, every launch flagged hipExtAnyOrderLaunch (AQL barrier bit off: the command processor may start a
//           launch before the previous one has completed - in queue order, so the older launch always gets its slots first),
//           dependency by the same completion counter
//   mode 1: two streams alternating, no events: the dependency is a completion counter the workgroups of the previous step
//           bump (release) and the workgroups of this step wait for (bounded spin, acquire) AFTER their independent part -
//           the workgroups of step t+1 move into the slots that step t's finishing workgroups free
// Prints the period per step.  A launch must fit the chip in one go (512 workgroups = 2 per CU), or mode 1 can starve.
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/overlap_probe scripts/dbg/overlap_probe.hip ; run: /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ double burn(double x, int iters) {   // ~ iters * 8 dependent f64 fma per thread
    for (int i = 0; i < iters; ++i) {
        x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9);
        x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9);
    }
    return x;
}
__global__ void __launch_bounds__(512) k_stepish(int mode, int t, int nwg, int itA, int itB, unsigned* done, double* out, int* timeouts) {
    extern __shared__ char smem[];
    double x = burn(1.0 + threadIdx.x * 1e-6 + smem[0] * 0.0, itA);   // independent part
    if (mode >= 1 && t > 0) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&done[t - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > 20000) { atomicAdd(timeouts, 1); break; }   // bounded (~3 ms): never hang the GPU
            }
        }
        __syncthreads();
    }
    x = burn(x, itB);   // dependent part
    if (x == 0.12345) out[blockIdx.x] = x;
    if (mode >= 1) {
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&done[t], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    const int nwg = 512, T = 200;
    unsigned* done; double* out; int* timeouts;
    (void)hipMalloc(&done, T * 4); (void)hipMalloc(&out, nwg * 8); (void)hipMalloc(&timeouts, 4);
    hipStream_t s[2]; (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    (void)hipFuncSetAttribute((const void*)k_stepish, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int itA : {60, 120}) for (int itB : {120, 180}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f; int to = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(done, 0, T * 4); (void)hipMemset(timeouts, 0, 4); (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, s[0]);
                for (int t = 0; t < T; ++t) {
                    if (mode == 2)
                        hipExtLaunchKernelGGL(k_stepish, dim3(nwg), dim3(512), 61 * 1024, s[0], nullptr, nullptr, hipExtAnyOrderLaunch, mode, t, nwg, itA, itB, done, out, timeouts);
                    else
                    hipLaunchKernelGGL(k_stepish, dim3(nwg), dim3(512), 61 * 1024, s[mode == 1 ? (t & 1) : 0], mode, t, nwg, itA, itB, done, out, timeouts);
                    if (mode == 1 && t == 0) {   // step 1 must not be dispatched before step 0 (afterwards the stream order of t-1 -> t+1 sees to it)
                        (void)hipEventRecord(e1, s[0]); (void)hipStreamWaitEvent(s[1], e1, 0);
                    }
                }
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e1, s[0]); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
                (void)hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
            }
            printf("A=%3d B=%3d iterations, mode %d (%s): %.2f us per step%s\n", itA, itB, mode, mode == 0 ? "one stream" : (mode == 1 ? "two streams + completion counter" : "one stream, any-order launches + completion counter"), best / T * 1e3,
                   to ? "  [SPIN TIMEOUTS]" : "");
            fflush(stdout);
        }
    }
    return 0;
}
