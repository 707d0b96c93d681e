// Can a kernel pull page-locked host memory over PCIe as fast as the copy engine does (56 GB/s on this pool)?  The
// Upload / Denoise / Download bracket of the reference is bound by 76 B/px of copies in, issued as 42 pieces behind one
// queue with ~9.5 us between pieces; a pre-pass that reads its four statistics images straight from the
// host would take 24 of the pieces off the queue.  hipcc -O3 --offload-arch=gfx950 pcie_read.hip -o pcie_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void pull(const vfloat4 *__restrict__ host, vfloat4 *__restrict__ dev, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const vfloat4 v = NT ? __builtin_nontemporal_load(host + i) : host[i];
        dev[i] = v;
    }
}

// stand-in for the window filter: one 512-thread workgroup per CU holding 150 KB of LDS and the whole register file (launch
// bounds force 256 VGPRs per lane), busy for ~0.2 ms
__global__ __launch_bounds__(512, 2) void hog(float *out, int iters) {
    extern __shared__ float lds[];
    float acc[96];
#pragma unroll
    for (int i = 0; i < 96; i++) acc[i] = threadIdx.x * 1e-3f + i;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");   // the whole register file, like the window filter (250 VGPRs, two waves per SIMD)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 96; i++) acc[i] = __builtin_fmaf(acc[i], 1.0001f, 0.5f);
        lds[threadIdx.x] = acc[it & 63];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 96; i++) s += acc[i];
    if (s == 12345.f) out[0] = s + lds[0];
}

int main() {
    const size_t bytes = (size_t)1920 * 1080 * 40;   // the four statistics images of a 1080p film: 82.9 MB
    const size_t n4 = bytes / 16;
    void *h, *h2;
    CHK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CHK(hipHostMalloc(&h2, bytes, hipHostMallocDefault));
    memset(h, 1, bytes);
    memset(h2, 2, bytes);
    void *hd = nullptr;
    CHK(hipHostGetDevicePointer(&hd, h, 0));
    vfloat4 *d, *d2;
    CHK(hipMalloc(&d, bytes));
    CHK(hipMalloc(&d2, bytes));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto timed = [&](const char *what, auto fn, double moved) {
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, s1));
            fn();
            CHK(hipStreamSynchronize(s2));
            CHK(hipEventRecord(e1, s1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-64s %.3f ms  %.1f GB/s\n", what, best, moved / best / 1e6);
        fflush(stdout);
    };
    timed("copy engine, one hipMemcpyAsync of 82.9 MB", [&] { CHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1)); }, (double)bytes);
    for (int grid : {16, 64, 256, 1024}) {
        char what[128];
        snprintf(what, sizeof what, "kernel pull, %4d workgroups, plain loads", grid);
        timed(what, [&] { hipLaunchKernelGGL(pull<false>, dim3(grid), dim3(256), 0, s1, (const vfloat4 *)hd, d, n4); }, (double)bytes);
        snprintf(what, sizeof what, "kernel pull, %4d workgroups, non-temporal loads", grid);
        timed(what, [&] { hipLaunchKernelGGL(pull<true>, dim3(grid), dim3(256), 0, s1, (const vfloat4 *)hd, d, n4); }, (double)bytes);
    }
    timed("copy engine (82.9 MB) beside a kernel pull (82.9 MB, 64 workgroups)", [&] {
        CHK(hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, s2));
        hipLaunchKernelGGL(pull<true>, dim3(64), dim3(256), 0, s1, (const vfloat4 *)hd, d, n4);
    }, 2.0 * bytes);
    timed("two copy-engine transfers on two streams (2 x 82.9 MB)", [&] {
        CHK(hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, s2));
        CHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1));
    }, 2.0 * bytes);
    // the 1080p upload as the band pipeline issues it: 7 images (12, 4, 12, 12, 12, 12, 12 B/px) x 6 bands = 42 pieces
    {
        const size_t px = (size_t)1920 * 1080;
        const int bpp[7] = {12, 4, 12, 12, 12, 12, 12};
        char *hs[7], *ds[7];
        for (int i = 0; i < 7; i++) {
            CHK(hipHostMalloc((void **)&hs[i], px * bpp[i], hipHostMallocDefault));
            memset(hs[i], i + 1, px * bpp[i]);
            CHK(hipMalloc((void **)&ds[i], px * bpp[i]));
        }
        const int queue2[7] = {0, 1, 0, 1, 0, 1, 1};   // 40 B/px on queue 0, 36 on queue 1
        auto pieces = [&](int mode) {   // 0: one copy-engine queue; 1: two; 2: copy engine + pulling kernel
            for (int k = 0; k < 6; k++)
                for (int i = 0; i < 7; i++) {
                    const size_t row = (size_t)1920 * bpp[i], y0 = 180 * k, rows = 180;
                    char *dst = ds[i] + y0 * row;
                    const char *src = hs[i] + y0 * row;
                    const bool second = mode > 0 && queue2[i];
                    if (second && mode == 2) {
                        void *m = nullptr;
                        CHK(hipHostGetDevicePointer(&m, (void *)src, 0));
                        hipLaunchKernelGGL(pull<false>, dim3(64), dim3(256), 0, s2, (const vfloat4 *)m, (vfloat4 *)dst, rows * row / 16);
                    } else {
                        CHK(hipMemcpyAsync(dst, src, rows * row, hipMemcpyHostToDevice, second ? s2 : s1));
                    }
                }
        };
        const double total = (double)px * 76;
        timed("42 pieces, one copy-engine queue", [&] { pieces(0); }, total);
        timed("42 pieces, two copy-engine queues", [&] { pieces(1); }, total);
        timed("42 pieces, copy-engine queue + pulling kernels", [&] { pieces(2); }, total);
        // the same beside a stream of filter-like kernels (345 workgroups: one round of every CU and a partial one, like a band)
        hipStream_t s3;
        CHK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
        CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float *hout;
        CHK(hipMalloc(&hout, 64));
        {
            hipEvent_t h0, h1; CHK(hipEventCreate(&h0)); CHK(hipEventCreate(&h1));
            hipLaunchKernelGGL(hog, dim3(345), dim3(512), 150 * 1024, s3, hout, 1200);
            CHK(hipEventRecord(h0, s3));
            hipLaunchKernelGGL(hog, dim3(345), dim3(512), 150 * 1024, s3, hout, 1200);
            CHK(hipEventRecord(h1, s3)); CHK(hipEventSynchronize(h1));
            float ms; CHK(hipEventElapsedTime(&ms, h0, h1));
            printf("one filter-like kernel (345 workgroups of 512 threads, 150 KB LDS): %.3f ms\n", ms);
        }
        for (int mode = 0; mode < 3; mode++) {
            char what[128];
            snprintf(what, sizeof what, "42 pieces, mode %d, beside 10 filter-like kernels", mode);
            timed(what, [&] { for (int r = 0; r < 10; r++) hipLaunchKernelGGL(hog, dim3(345), dim3(512), 150 * 1024, s3, hout, 1200); pieces(mode); }, total);
            CHK(hipStreamSynchronize(s3));
        }
    }
    return 0;
}
