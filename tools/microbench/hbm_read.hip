// Streaming-read bandwidth of the accumulate kernel's access pattern on gfx950: each lane walks S
// planes (stride = plane bytes) reading C float4 per plane, like accumulate_kernel, with a trivial
// reduction instead of the moment update.  Variants: non-temporal or plain loads, unroll depth.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <int C, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void rd(const float *__restrict__ src, float *out, long long n_px, int S) {
    const long long n_groups = n_px / 4, n_elems = n_px * C;
    float acc = 0.f;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n_groups; g += (long long)gridDim.x * 256) {
        const float *sp = src + g * 4 * C;
#pragma unroll UNROLL
        for (int s = 0; s < S; s++, sp += n_elems) {
#pragma unroll
            for (int k = 0; k < C; k++) {
                vfloat4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const vfloat4 *>(sp + 4 * k))
                               : *reinterpret_cast<const vfloat4 *>(sp + 4 * k);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int C, bool NT, int UNROLL>
void run(const float *src, float *out, long long n_px, int S, int grid) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rd<C, NT, UNROLL>), dim3(grid), dim3(256), 0, 0, src, out, n_px, S);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((rd<C, NT, UNROLL>), dim3(grid), dim3(256), 0, 0, src, out, n_px, S);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("C=%d nt=%d unroll=%d grid=%5d: %.3f ms  %.0f GB/s\n", C, (int)NT, UNROLL, grid, ms,
           (double)n_px * C * 4 * S / ms / 1e6);
}

int main() {
    const long long n_px = 1920 * 1080;
    const int S = 256;
    float *src, *out;
    CHK(hipMalloc(&src, (size_t)n_px * 3 * 4 * S));  // 6.4 GB
    CHK(hipMalloc(&out, 4));
    CHK(hipMemset(src, 0, (size_t)n_px * 3 * 4 * S));
    for (int grid : {2048, 8192}) {
        run<3, true, 2>(src, out, n_px, S, grid);
        run<3, true, 4>(src, out, n_px, S, grid);
        run<3, true, 8>(src, out, n_px, S, grid);
        run<3, false, 4>(src, out, n_px, S, grid);
        run<1, true, 4>(src, out, n_px * 3, S, grid);
        run<1, true, 8>(src, out, n_px * 3, S, grid);
    }
    return 0;
}
