#include "../../sequential_monte_carlo_amd/csrc/smc_kernels.h"
#include <cstdio>
#include <vector>
using namespace smc;
__global__ __launch_bounds__(256) void kdbg(const double* lwin, uint64_t* C, uint64_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* scr = (uint64_t*)smem;
    double lw[2][2];
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 2; ++j) lw[k][j] = lwin[2 * (threadIdx.x + k * 256) + j];
    SegRec r = segment_normalize<256, 2>(lw, scr, C);
    if (threadIdx.x == 0) { out[0] = d2bits(r.m); out[1] = r.S; out[2] = r.hi; out[3] = r.lo;
        for (int i = 0; i < 16; ++i) out[4 + i] = scr[2 * 4 + i]; }
}
int main() {
    std::vector<double> lw(1024);
    FILE* f = fopen("scripts/dbg/lw.bin", "rb"); if (!f) f = fopen("lw.bin", "rb");
    fread(lw.data(), 8, 1024, f); fclose(f);
    double* d; uint64_t *C, *o; hipMalloc(&d, 8192); hipMalloc(&C, 8192); hipMalloc(&o, 256);
    hipMemcpy(d, lw.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kdbg, dim3(1), dim3(256), scr_words(256, 2) * 8, 0, d, C, o);
    uint64_t h[20]; hipMemcpy(h, o, 160, hipMemcpyDeviceToHost);
    printf("S=%llx hi=%llx lo=%llx\n", (unsigned long long)h[1], (unsigned long long)h[2], (unsigned long long)h[3]);
    for (int i = 0; i < 12; ++i) printf("w2[%d]=%llx\n", i, (unsigned long long)h[4 + i]);
    // host reference from C
    std::vector<uint64_t> hc(1024); hipMemcpy(hc.data(), C, 8192, hipMemcpyDeviceToHost);
    unsigned __int128 s2 = 0; uint64_t A=0,B=0,Cc=0;
    uint64_t aw[4]={0},bw[4]={0},cw[4]={0};
    for (int i = 0; i < 1024; ++i) { uint64_t q = hc[i] - (i ? hc[i-1] : 0); s2 += (unsigned __int128)q*q; uint64_t qh=q>>24, ql=q&0xffffff; A+=qh*qh;B+=qh*ql;Cc+=ql*ql;
        int pair=i/2; int tid=pair%256; int w=tid/64; aw[w]+=qh*qh; bw[w]+=qh*ql; cw[w]+=ql*ql; }
    printf("host hi=%llx lo=%llx A=%llx B=%llx C=%llx\n", (unsigned long long)(s2>>64), (unsigned long long)s2, (unsigned long long)A,(unsigned long long)B,(unsigned long long)Cc);
    for (int w=0;w<4;++w) printf("host wave %d a=%llx b=%llx c=%llx\n", w,(unsigned long long)aw[w],(unsigned long long)bw[w],(unsigned long long)cw[w]);
    return 0;
}
