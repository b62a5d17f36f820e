// Diagnostic (never shipped): what does a per-step all-to-all dependency cost INSIDE one persistent launch, against one launch per
// step?  Synthetic steps of k_step's shape: 512 workgroups x 512 threads, 61 KB of LDS (two per CU), per step and workgroup
//   A  independent arithmetic (the next step's Philox / Box-Muller work),
//   -- needs EVERY workgroup's 16-byte record of the previous step (K, S of its segment) --
//   B  dependent arithmetic, which reads 3 x 16 KB of the previous step's payload of its neighbours (staged C) and writes 32 KB (x, C).
// mode 0: one launch per step on one stream (what log_likelihood does today)
// mode 1: ONE launch; payload by sc1 (write-through) 16-byte stores, every storing wave drains (vmcnt(0)), barrier, ONE lane publishes the
//         workgroup's record as a 16-byte {tag, value} granule (sc1 store); consumers poll the records (sc1 loads), one lane's agent
//         acquire, plain payload loads.  Options (bit mask):
//           1  only the lanes whose record has not arrived yet poll again (exec-masked reload)
//           2  longer sleep between polls (s_sleep 8 instead of 2)
//           4  arrival counters (16 shards, agent-scope atomic add after the record store) polled by ONE lane per workgroup; the records
//              are then loaded once
//           8  half of A moved between the payload stores and their drain (covers the write-through latency)
//          16  payload loads sc1 as well, no acquire
// Every payload word carries (step, workgroup, index): a stale or torn read is COUNTED (errs), a spin that expires too (timeouts).
// Stamps (s_memrealtime, 10 ns): mean microseconds per step a workgroup spends in A | poll | acquire+barrier | payload reads | B | stores
// issued | drained + barrier | published.
// build: hipcc -O2 --offload-arch=gfx950 -o scripts/dbg/bin/persist_probe scripts/dbg/persist_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int NWG = 512, TH = 512, SEGW = 2048;   // payload words (8 B) per workgroup and buffer half (x | C: 2 x 16 KB)
constexpr int NSHARD = 16;
typedef unsigned long long u64;
typedef u64 v2u64 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double burn(double x, int iters) {
    for (int i = 0; i < iters; ++i) {
        x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9);
        x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9);
    }
    return x;
}
__device__ __forceinline__ v2u64 load16_sc1(const u64* p) {
    v2u64 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store16_sc1(u64* p, v2u64 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void psleep(int opt) { if (opt & 2) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ u64 word_of(unsigned t, unsigned wg, unsigned i) { return ((u64)t << 40) | ((u64)wg << 20) | i; }

struct Args {
    u64* rec[2];     // [NWG][2] records: {tag = step, value}
    u64* pay[2];     // [NWG][2 * SEGW] payload
    unsigned* cnt;   // [NSHARD * 32] arrival counters, one per 128-byte line
    unsigned* dflag[2];   // [NWG * 32] per-workgroup DATA flags (opt 32): payload of step t drained; one per 128-byte line
    int* errs;       // [2]: stale payload words, spin timeouts
    double* sink;
    u64* stamps;     // [NWG][8] summed intervals (10 ns)
    int itA, itB, mode, opt, T;
};
#define STAMP(k) do { if (persistent && tid == 0) { const u64 now_ = __builtin_amdgcn_s_memrealtime(); acc[k] += now_ - last; last = now_; } } while (0)

// one step of workgroup wg; cur = buffer the previous step wrote
__device__ __forceinline__ void step_body(const Args& a, int wg, int t, int cur, bool persistent, char* smem, bool& dead, u64 (&acc)[8], u64& last,
                                          double& carry) {
    const int tid = threadIdx.x;
    const bool splitA = persistent && (a.opt & 8);
    double x = splitA ? carry : burn(1.0 + tid * 1e-6, a.itA);          // independent part (or its second half, done behind the last stores)
    if (splitA) x = burn(x, a.itA - a.itA / 2);
    STAMP(0);
    u64 K = 0;
    if (persistent && t > 0) {
        const u64* r = a.rec[cur] + 2 * (size_t)tid;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        if (a.opt & 4) {
            if (tid == 0) {   // ONE lane polls the arrival counters: all NWG workgroups have published step t
                for (; !dead;) {
                    unsigned s = 0;
                    for (int k = 0; k < NSHARD; ++k) s += __hip_atomic_load(a.cnt + 32 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (s >= (unsigned)NWG * (unsigned)t) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { atomicAdd(&a.errs[1], 1); dead = true; break; }
                    psleep(a.opt);
                }
            }
            __syncthreads();
            const v2u64 v = load16_sc1(r);
            K = ((v.x >> 32) == (u64)t && (unsigned)v.x == (unsigned)tid) ? v.y : ~0ull;
        } else if (a.opt & 1) {
            bool ok = dead;
            while (!ok) {   // divergent: only the lanes still waiting load again
                const v2u64 v = load16_sc1(r);
                ok = (v.x >> 32) == (u64)t && (unsigned)v.x == (unsigned)tid;
                K = v.y;
                if (!ok) {
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { atomicAdd(&a.errs[1], 1); dead = true; break; }
                    psleep(a.opt);
                }
            }
        } else {
            bool ok = false;
            for (; !dead;) {
                const v2u64 v = load16_sc1(r);
                ok = (v.x >> 32) == (u64)t && (unsigned)v.x == (unsigned)tid;
                K = v.y;
                if (__all(ok)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { if ((tid & 63) == 0) atomicAdd(&a.errs[1], 1); dead = true; break; }   // 20 ms, once
                psleep(a.opt);
            }
        }
        STAMP(1);
        if (!(a.opt & (16 | 32))) {
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        STAMP(2);
    } else if (t > 0) {
        K = a.rec[cur][2 * (size_t)tid + 1];
    }
    if (t > 0 && !dead && K != (u64)(t * 1000 + tid)) { if (tid == 0) atomicAdd(&a.errs[0], 1); }
    const bool early = persistent && (a.opt & 32);
    double xb = 0.0;
    if (early) {   // the part of B that needs the records only (segment table, targets), then the neighbours' payload flags
        xb = burn(x + smem[tid & 63] * 0.0, a.itB / 4);
        if (t > 0) {
            if (tid < 3) {
                const int src = (wg + tid - 1 + NWG) % NWG;
                const u64 t0 = __builtin_amdgcn_s_memrealtime();
                while (!dead && __hip_atomic_load(a.dflag[cur] + 32 * src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)t) {
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { atomicAdd(&a.errs[1], 1); dead = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (!(a.opt & 16)) { if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } }
            __syncthreads();
        }
    }
    // dependent part: read the 3 neighbouring payload halves of the previous step (staging), check every word
    int bad = 0;
    if (t > 0) {
        for (int nb = -1; nb <= 1; ++nb) {
            const int src = (wg + nb + NWG) % NWG;
            const u64* p = a.pay[cur] + (size_t)src * 2 * SEGW + SEGW;   // its "C" half
            for (int k = 0; k < 2; ++k) {
                const int i = 2 * (tid + k * TH);
                v2u64 v;
                if (persistent && (a.opt & 16)) v = load16_sc1(p + i);
                else v = *reinterpret_cast<const v2u64*>(p + i);
                bad += v.x != word_of(t, src, SEGW + i) || v.y != word_of(t, src, SEGW + i + 1);
            }
        }
        // a few random 8-byte gathers from the "x" half of the neighbours
        for (int g = 0; g < 4; ++g) {
            const unsigned h = (tid * 2654435761u + g * 40503u + t * 97u);
            const int src = (wg + (int)(h % 3) - 1 + NWG) % NWG, i = (h >> 8) % SEGW;
            const u64* p = a.pay[cur] + (size_t)src * 2 * SEGW + i;
            u64 v;
            if (persistent && (a.opt & 16)) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = *p;
            bad += v != word_of(t, src, i);
        }
    }
    if (bad && !dead) atomicAdd(&a.errs[0], bad);
    STAMP(3);
    x = early ? burn(xb, a.itB - a.itB / 4) : burn(x + smem[tid & 63] * 0.0, a.itB);
    if (x == 0.12345) a.sink[wg] = x;
    STAMP(4);
    if (early && tid == 0) {   // the record goes out BEFORE the payload: everybody's segment table needs it, only the neighbours need the payload
        v2u64 v;
        v.x = ((u64)(t + 1) << 32) | (unsigned)wg;
        v.y = (u64)((t + 1) * 1000 + wg);
        store16_sc1(a.rec[cur ^ 1] + 2 * (size_t)wg, v);
    }
    // write this step's payload (x | C halves: 2 x 2 x 16 B per thread) and publish the record
    const int nxt = cur ^ 1;
    u64* q = a.pay[nxt] + (size_t)wg * 2 * SEGW;
    for (int k = 0; k < 4; ++k) {
        const int i = 2 * (tid + k * TH);
        v2u64 v;
        v.x = word_of(t + 1, wg, i);
        v.y = word_of(t + 1, wg, i + 1);
        if (persistent) store16_sc1(q + i, v);
        else *reinterpret_cast<v2u64*>(q + i) = v;
    }
    STAMP(5);
    if (persistent) {
        if (splitA) carry = burn(1.0 + tid * 1e-6 + x * 1e-30, a.itA / 2);   // first half of the NEXT step's independent part, under the stores' latency
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(6);
        if (early) {
            if (tid == 0) __hip_atomic_store(a.dflag[nxt] + 32 * wg, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (tid == 0) {
            v2u64 v;
            v.x = ((u64)(t + 1) << 32) | (unsigned)wg;
            v.y = (u64)((t + 1) * 1000 + wg);
            store16_sc1(a.rec[nxt] + 2 * (size_t)wg, v);
            if (a.opt & 4) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.cnt + 32 * (wg % NSHARD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        STAMP(7);
    } else if (tid == 0) {
        a.rec[nxt][2 * (size_t)wg] = ((u64)(t + 1) << 32) | (unsigned)wg;
        a.rec[nxt][2 * (size_t)wg + 1] = (u64)((t + 1) * 1000 + wg);
    }
}

__global__ void __launch_bounds__(TH) k_one(Args a, int t, int cur) {
    extern __shared__ char smem[];
    bool dead = false;
    u64 acc[8] = {0}, last = 0;
    double carry = 0;
    step_body(a, blockIdx.x, t, cur, false, smem, dead, acc, last, carry);
}
__global__ void __launch_bounds__(TH) k_persist(Args a) {
    extern __shared__ char smem[];
    int cur = 0;
    bool dead = false;
    u64 acc[8] = {0}, last = __builtin_amdgcn_s_memrealtime();
    double carry = burn(1.0 + threadIdx.x * 1e-6, a.itA / 2);
    for (int t = 0; t < a.T; ++t) { step_body(a, blockIdx.x, t, cur, true, smem, dead, acc, last, carry); cur ^= 1; }
    if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) a.stamps[blockIdx.x * 8 + k] = acc[k];
}

int main(int argc, char** argv) {
    Args a{};
    const int T = 400;
    for (int b = 0; b < 2; ++b) {
        (void)hipMalloc(&a.rec[b], NWG * 16);
        (void)hipMalloc(&a.pay[b], (size_t)NWG * 2 * SEGW * 8);
    }
    (void)hipMalloc(&a.errs, 8); (void)hipMalloc(&a.sink, NWG * 8); (void)hipMalloc(&a.cnt, NSHARD * 128); (void)hipMalloc(&a.stamps, NWG * 64); for (int b = 0; b < 2; ++b) (void)hipMalloc(&a.dflag[b], NWG * 128);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_persist, TH, 61 * 1024);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    printf("occupancy API: %d workgroups per CU x %d CUs = %d (grid %d)\n", nb, prop.multiProcessorCount, nb * prop.multiProcessorCount, NWG);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int AB[][2] = {{0, 0}, {60, 120}, {100, 100}, {100, 140}};
    const int OPTS[] = {-1, 0, 16, 32, 40, 48, 56, 33, 41};
    for (auto& ab : AB) {
        for (int opt : OPTS) {
            const int mode = opt < 0 ? 0 : 1;
            a.itA = ab[0]; a.itB = ab[1]; a.mode = mode; a.opt = opt < 0 ? 0 : opt; a.T = T;
            float best = 1e9f; int errs[2] = {0, 0};
            std::vector<u64> st(NWG * 8);
            for (int rep = 0; rep < 3; ++rep) {
                for (int b = 0; b < 2; ++b) { (void)hipMemsetAsync(a.rec[b], 0, NWG * 16, s); (void)hipMemsetAsync(a.pay[b], 0xff, (size_t)NWG * 2 * SEGW * 8, s); }
                (void)hipMemsetAsync(a.errs, 0, 8, s); (void)hipMemsetAsync(a.cnt, 0, NSHARD * 128, s); for (int b = 0; b < 2; ++b) (void)hipMemsetAsync(a.dflag[b], 0, NWG * 128, s);
                (void)hipStreamSynchronize(s);
                (void)hipEventRecord(e0, s);
                if (mode == 0) { int cur = 0; for (int t = 0; t < T; ++t) { hipLaunchKernelGGL(k_one, dim3(NWG), dim3(TH), 61 * 1024, s, a, t, cur); cur ^= 1; } }
                else hipLaunchKernelGGL(k_persist, dim3(NWG), dim3(TH), 61 * 1024, s, a);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; if (mode) (void)hipMemcpy(st.data(), a.stamps, NWG * 64, hipMemcpyDeviceToHost); }
                int e[2]; (void)hipMemcpy(e, a.errs, 8, hipMemcpyDeviceToHost);
                errs[0] += e[0]; errs[1] += e[1];
            }
            printf("A=%3d B=%3d  %s opt %2d: %6.2f us per step   stale/torn %d, timeouts %d", ab[0], ab[1], mode ? "persistent" : "launches  ", opt, best / T * 1e3, errs[0], errs[1]);
            if (mode) {
                printf("   [A|poll|acq|reads|B|stores|drain|publish] =");
                for (int k = 0; k < 8; ++k) { double sum = 0; for (int w = 0; w < NWG; ++w) sum += (double)st[w * 8 + k]; printf(" %.2f", sum / NWG / T * 0.01); }
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
