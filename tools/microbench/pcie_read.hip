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
    return 0;
}
