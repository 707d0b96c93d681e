// Round 5: would the accumulation gain from writing its moments in chip-wide PHASES?  HISTORY.md 4.1: the state's stores cost
// about four times what read bytes cost when they trickle into the saturated read stream.  This program moves the bytes of
// accumulate_kernel (per thread and pass: 448 B of state read, S x 176 B of samples streamed, 448 B of state written; a
// persistent grid of 2 workgroups per CU) three ways:
//   free      every workgroup stores when its pass is done (what the kernel does)
//   barrier1  a soft grid barrier BEFORE the stores (every workgroup has finished reading when the first one writes)
//   barrier2  ... and another one behind them (nobody reads again before everybody has written)
// and, free-running, with the state of a wave (and its sample rows) in contiguous blocks instead of one plane per channel.
// The barrier is bounded (a workgroup goes on after 40 us whatever the counter says): it cannot hang.
// hipcc -O3 --offload-arch=gfx950 phase_probe.hip -o phase_probe && ./phase_probe [width height]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));

constexpr int kStateV = 28;    // float4 per thread of state (4 pixels x 112 B)
constexpr int kRowV = 11;      // float4 per thread and sample (4 pixels x 44 B)

__device__ __forceinline__ void soft_barrier(unsigned *counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target &&
               __builtin_amdgcn_s_memrealtime() - t0 < 4000ull)
            __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
}

// thread t of the grid owns thread-items t, t + T, t + 2 T, ... (item = 4 pixels); float4 k of sample s of item i is at
// samples[(s * kRowV + k) * n_items + i] (every load of a wave is 1 KiB in a row)
// LAYOUT 0: every float4 index k a plane of its own (the product's images).  1: the state of a wave's 64 items in one 28-KiB block
// ([wave block][k][lane]).  2: ... and a wave's sample rows in one block per sample ([s][wave block][k][lane]).
template <int MODE, int LAYOUT = 0>
__global__ __launch_bounds__(256, 2) void phases(const vfloat4 *samples, vfloat4 *state, size_t n_items, int S, unsigned *counter) {
    const size_t T = (size_t)gridDim.x * blockDim.x;
    const size_t n_pass = (n_items + T - 1) / T;
    unsigned phase = 0;
    for (size_t pass = 0; pass < n_pass; pass++) {
        const size_t i = pass * T + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool on = i < n_items;
        vfloat4 st[kStateV];
        if (on) {
#pragma unroll
            for (int k = 0; k < kStateV; k++) st[k] = LAYOUT >= 1 ? state[(i >> 6) * (kStateV * 64) + k * 64 + (i & 63)] : state[(size_t)k * n_items + i];
            for (int s = 0; s < S; s++) {
                const vfloat4 *row = LAYOUT >= 2 ? samples + (size_t)s * kRowV * n_items + (i >> 6) * (kRowV * 64) + (i & 63) : samples + (size_t)s * kRowV * n_items + i;
                const size_t kstride = LAYOUT >= 2 ? 64 : n_items;
#pragma unroll
                for (int k = 0; k < kRowV; k++) st[k] += __builtin_nontemporal_load(row + (size_t)k * kstride);
            }
        }
        if (MODE >= 1) soft_barrier(counter, ++phase * gridDim.x);
        if (on) {
#pragma unroll
            for (int k = 0; k < kStateV; k++) (LAYOUT >= 1 ? state[(i >> 6) * (kStateV * 64) + k * 64 + (i & 63)] : state[(size_t)k * n_items + i]) = st[k];
        }
        if (MODE >= 2) soft_barrier(counter, ++phase * gridDim.x);
    }
}

int main(int argc, char **argv) {
    const int W = argc > 2 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160;
    const size_t n_items = (size_t)W * H / 4;
    int dev = 0, cus = 0;
    CHK(hipGetDevice(&dev));
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = 2 * cus;
    const int S_max = 64;
    vfloat4 *samples, *state;
    unsigned *counter;
    CHK(hipMalloc(&samples, (size_t)S_max * n_items * kRowV * 16));
    CHK(hipMalloc(&state, n_items * kStateV * 16));
    CHK(hipMalloc(&counter, 4));
    CHK(hipMemset(samples, 0, (size_t)S_max * n_items * kRowV * 16));
    CHK(hipMemset(state, 0, n_items * kStateV * 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%dx%d, %d workgroups, %zu items, %zu passes per thread\n", W, H, grid, n_items, (n_items + (size_t)grid * 256 - 1) / ((size_t)grid * 256));
    for (int S : {4, 16, 64}) {
        const double bytes = (double)n_items * (2.0 * kStateV * 16 + (double)S * kRowV * 16);
        printf("S = %2d (%.2f GB):", S, bytes / 1e9);
        for (int mode = 0; mode < 5; mode++) {
            std::vector<float> ms;
            for (int rep = 0; rep < 6; rep++) {
                CHK(hipMemset(counter, 0, 4));
                CHK(hipEventRecord(e0, nullptr));
                if (mode == 0) hipLaunchKernelGGL(phases<0>, dim3(grid), dim3(256), 0, nullptr, samples, state, n_items, S, counter);
                else if (mode == 1) hipLaunchKernelGGL(phases<1>, dim3(grid), dim3(256), 0, nullptr, samples, state, n_items, S, counter);
                else if (mode == 2) hipLaunchKernelGGL(phases<2>, dim3(grid), dim3(256), 0, nullptr, samples, state, n_items, S, counter);
                else if (mode == 3) hipLaunchKernelGGL((phases<0, 1>), dim3(grid), dim3(256), 0, nullptr, samples, state, n_items, S, counter);
                else hipLaunchKernelGGL((phases<0, 2>), dim3(grid), dim3(256), 0, nullptr, samples, state, n_items, S, counter);
                CHK(hipEventRecord(e1, nullptr));
                CHK(hipEventSynchronize(e1));
                float t = 0.f;
                CHK(hipEventElapsedTime(&t, e0, e1));
                if (rep) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("   %s %.3f ms %.2f TB/s", mode == 0 ? "free" : mode == 1 ? "barrier1" : mode == 2 ? "barrier2" : mode == 3 ? "state-blocked" : "all-blocked", ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e9);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
