// Streaming-read bandwidth of the accumulate kernel's access pattern through LDS-DMA (global_load_lds_dwordx4) instead
// of loads into registers: every wave walks S planes (stride = plane bytes) of its 64 lanes x 4 pixels x C floats
// (1 KiB / 3 KiB contiguous per plane), D planes in flight in a wave-private LDS ring, each lane reading its own 16 B x C
// back from LDS.  MI355X_MICROARCH.md gives 6.4 (default policy) / 6.5 - 6.8 TB/s (nt) for LDS-DMA streams against
// 6.0 - 6.3 for register loads.  hipcc -O3 --offload-arch=gfx950 hbm_read_ldsdma.hip -o hbm_read_ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// C floats4 per lane and plane; D planes in flight; NT: non-temporal (aux = 2)
template <int C, int D, bool NT>
__global__ __launch_bounds__(256) void rd(const float *__restrict__ src, float *out, long long n_px, int S) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *ring = lds + wave * (D * C * 256);          // D slots of C x 1 KiB
    const long long n_groups = n_px / 4, n_elems = n_px * C;
    const long long n_wave_groups = n_groups / 64;     // whole waves only (n_px multiple of 256 in this benchmark)
    float acc = 0.f;
    for (long long wg = (long long)blockIdx.x * 4 + wave; wg < n_wave_groups; wg += (long long)gridDim.x * 4) {
        const float *base = src + wg * 64 * 4 * C;     // the wave's 64 x 4 x C floats of plane 0
        auto issue = [&](int s) {
            const float *p = base + (long long)s * n_elems + lane * 4;
            float *slot = ring + (s % D) * (C * 256);
#pragma unroll
            for (int k = 0; k < C; k++)
                __builtin_amdgcn_global_load_lds(p + k * 256, (__attribute__((address_space(3))) void *)(slot + k * 256), 16, 0, NT ? 2 : 0);
        };
#pragma unroll
        for (int s = 0; s < D; s++) if (s < S) issue(s);
        for (int s = 0; s < S; s++) {
            // the oldest plane has landed when at most (D - 1) planes' transfers are outstanding
            if (s + D <= S) wait_vm<C * (D - 1)>(); else wait_vm<0>();
            const float *slot = ring + (s % D) * (C * 256);
            vfloat4 v[C];
#pragma unroll
            for (int k = 0; k < C; k++) v[k] = *reinterpret_cast<const vfloat4 *>(slot + lane * 4 * C + 4 * k);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot has been read: it may be refilled
            if (s + D < S) issue(s + D);
#pragma unroll
            for (int k = 0; k < C; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int C, int D, bool NT>
void run(const float *src, float *out, long long n_px, int S, int grid) {
    const size_t lds = (size_t)4 * D * C * 1024;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rd<C, D, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rd<C, D, NT>), dim3(grid), dim3(256), lds, 0, src, out, n_px, S);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((rd<C, D, NT>), dim3(grid), dim3(256), lds, 0, src, out, n_px, S);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double bytes = (double)n_px * C * 4 * S;
    printf("C=%d D=%2d nt=%d grid=%5d lds/WG=%3zu KB: %.3f ms  %.0f GB/s\n", C, D, (int)NT, grid, lds / 1024, ms, bytes / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long n_px = 1920LL * 1080;
    const int S = 256;
    float *src, *out;
    CHK(hipMalloc(&src, (size_t)n_px * 3 * 4 * S));
    CHK(hipMalloc(&out, 64));
    CHK(hipMemset(src, 0, (size_t)n_px * 3 * 4 * S));
    for (int grid : {512, 1024, 2048}) {
        run<1, 8, true>(src, out, n_px, S, grid);
        run<1, 16, true>(src, out, n_px, S, grid);
        run<1, 16, false>(src, out, n_px, S, grid);
        run<3, 3, true>(src, out, n_px, S, grid);
        run<3, 5, true>(src, out, n_px, S, grid);
        run<3, 5, false>(src, out, n_px, S, grid);
        run<3, 8, true>(src, out, n_px, S, grid);
    }
    return 0;
}
