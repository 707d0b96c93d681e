// Do two kernels from two streams share a CU on gfx950?  Kernel A stands in for the window filter
// (512 threads, ~157 KB LDS, VA VGPRs, thousands of short workgroups), kernel B for the accumulate
// kernel (256 threads, no LDS, VB VGPRs, a few hundred long workgroups).  Every workgroup records
// where it ran (XCC, SE, CU) and when (wall clock), and the host prints how the two interleave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };

__device__ __forceinline__ float spin(int iters, float x) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    }
    return x;
}

#ifndef VA
#define VA 168
#endif
#ifndef VB
#define VB 168
#endif
#define STR2(x) #x
#define STR(x) STR2(x)

__global__ __launch_bounds__(512) void kernA(Rec *rec, float *sink, int iters) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    asm volatile("v_mov_b32 v" STR(VA) ", 0" ::: "v" STR(VA));
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float x = spin(iters, lds[(threadIdx.x + 1) & 511]);
    if (x == 123.f) sink[0] = x;
    if (threadIdx.x == 0) {
        Rec r;
        r.hw = __builtin_amdgcn_s_getreg(63492);
        r.xcc = __builtin_amdgcn_s_getreg(63508);
        r.t0 = t0; r.t1 = wall_clock64();
        rec[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void kernB(Rec *rec, float *sink, int iters) {
    const unsigned long long t0 = wall_clock64();
    asm volatile("v_mov_b32 v" STR(VB) ", 0" ::: "v" STR(VB));
    float x = spin(iters, (float)threadIdx.x);
    if (x == 123.f) sink[0] = x;
    if (threadIdx.x == 0) {
        Rec r;
        r.hw = __builtin_amdgcn_s_getreg(63492);
        r.xcc = __builtin_amdgcn_s_getreg(63508);
        r.t0 = t0; r.t1 = wall_clock64();
        rec[blockIdx.x] = r;
    }
}

static int cu_key(const Rec &r) { return ((r.xcc & 15) << 8) | (((r.hw >> 13) & 7) << 5) | (((r.hw >> 12) & 1) << 4) | ((r.hw >> 8) & 15); }

int main(int argc, char **argv) {
    const int gridA = argc > 1 ? atoi(argv[1]) : 3240, gridB = argc > 2 ? atoi(argv[2]) : 256;
    const int itersA = argc > 3 ? atoi(argv[3]) : 600, itersB = argc > 4 ? atoi(argv[4]) : 30000;
    const size_t ldsA = argc > 5 ? (size_t)atoi(argv[5]) : 160608;
    const int order = argc > 6 ? atoi(argv[6]) : 0;  // 0: A first, 1: B first
    Rec *ra, *rb; float *sink;
    CHK(hipMalloc(&ra, sizeof(Rec) * gridA)); CHK(hipMalloc(&rb, sizeof(Rec) * gridB)); CHK(hipMalloc(&sink, 4));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernA), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t sa, sb; CHK(hipStreamCreate(&sa)); CHK(hipStreamCreate(&sb));
    hipEvent_t e0, e1, e2; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2));
    float tA = 0, tB = 0, tAB = 0;
    for (int rep = 0; rep < 2; rep++) {  // alone
        CHK(hipEventRecord(e0, sa)); hipLaunchKernelGGL(kernA, dim3(gridA), dim3(512), ldsA, sa, ra, sink, itersA); CHK(hipEventRecord(e1, sa));
        CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&tA, e0, e1));
        CHK(hipEventRecord(e0, sb)); hipLaunchKernelGGL(kernB, dim3(gridB), dim3(256), 0, sb, rb, sink, itersB); CHK(hipEventRecord(e1, sb));
        CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&tB, e0, e1));
    }
    CHK(hipEventRecord(e0, 0)); CHK(hipStreamWaitEvent(sa, e0, 0)); CHK(hipStreamWaitEvent(sb, e0, 0));
    if (order == 0) {
        hipLaunchKernelGGL(kernA, dim3(gridA), dim3(512), ldsA, sa, ra, sink, itersA);
        hipLaunchKernelGGL(kernB, dim3(gridB), dim3(256), 0, sb, rb, sink, itersB);
    } else {
        hipLaunchKernelGGL(kernB, dim3(gridB), dim3(256), 0, sb, rb, sink, itersB);
        hipLaunchKernelGGL(kernA, dim3(gridA), dim3(512), ldsA, sa, ra, sink, itersA);
    }
    CHK(hipEventRecord(e1, sa)); CHK(hipEventRecord(e2, sb));
    CHK(hipStreamWaitEvent(0, e1, 0)); CHK(hipStreamWaitEvent(0, e2, 0));
    hipEvent_t e3; CHK(hipEventCreate(&e3)); CHK(hipEventRecord(e3, 0)); CHK(hipDeviceSynchronize());
    CHK(hipEventElapsedTime(&tAB, e0, e3));
    std::vector<Rec> ha(gridA), hb(gridB);
    CHK(hipMemcpy(ha.data(), ra, sizeof(Rec) * gridA, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hb.data(), rb, sizeof(Rec) * gridB, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
    for (auto &r : ha) { a0 = std::min(a0, r.t0); a1 = std::max(a1, r.t1); }
    for (auto &r : hb) { b0 = std::min(b0, r.t0); b1 = std::max(b1, r.t1); }
    tmin = std::min(a0, b0);
    std::map<int, int> nb;  // B workgroups per CU
    for (auto &r : hb) nb[cu_key(r)]++;
    std::map<int, int> hist;
    for (auto &kv : nb) hist[kv.second]++;
    // A workgroups that ran while a B workgroup was resident on the same CU
    long long co = 0;
    for (auto &r : ha) {
        const int k = cu_key(r);
        for (auto &q : hb) if (cu_key(q) == k && q.t0 < r.t1 && r.t0 < q.t1) { co++; break; }
    }
    printf("VA=%d VB=%d gridA=%d gridB=%d ldsA=%zu order=%s | A alone %.3f ms, B alone %.3f ms, both %.3f ms\n", VA, VB, gridA, gridB, ldsA,
           order ? "B-first" : "A-first", tA, tB, tAB);
    printf("  wall clock (100 MHz ticks from first start): A [%llu, %llu]  B [%llu, %llu]\n", a0 - tmin, a1 - tmin, b0 - tmin, b1 - tmin);
    printf("  CUs used by B: %zu; B workgroups per CU histogram:", nb.size());
    for (auto &kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\n  A workgroups that shared their CU with a resident B workgroup: %lld of %d\n", co, gridA);
    return 0;
}
