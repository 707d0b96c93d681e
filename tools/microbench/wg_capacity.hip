// Round 5: how many workgroups of the pair-symmetric filter's shape (512 threads, ~151 KB of LDS) does the chip run AT ONCE?
// tools/experiments/strip_scan2.py: 180 tiles take one tile's time, 240 take two.  Every workgroup here spins for a fixed time
// (s_memrealtime) and touches its LDS; the launch time steps up where the grid stops fitting in one round.
// hipcc -O3 --offload-arch=gfx950 wg_capacity.hip -o wg_capacity && ./wg_capacity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// SCRATCH > 0: every lane keeps that many floats in private (scratch) memory, like a kernel that spills
// VREGS: the highest vector register the kernel claims (an empty asm statement that names it), like a kernel of that many VGPRs
template <int THREADS, int SCRATCH = 0, int VREGS = 0>
__global__ __launch_bounds__(THREADS) void spin(unsigned long long ticks, int lds_floats, float *sink, unsigned *where) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (VREGS == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if constexpr (VREGS == 248) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    if constexpr (VREGS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    [[maybe_unused]] volatile float priv[SCRATCH > 0 ? SCRATCH : 1];
    if constexpr (SCRATCH > 0) {
        for (int i = 0; i < SCRATCH; i++) priv[i] = (float)(i + threadIdx.x);
    }
    for (int i = threadIdx.x; i < lds_floats; i += THREADS) lds[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0 && where) {
        unsigned xcc = 0, hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        where[2 * blockIdx.x] = xcc | ((unsigned)(t0 & 0xFFFFFFull) << 8);      // + when the workgroup started (10 ns units, 24 bits)
        where[2 * blockIdx.x + 1] = hw;
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 7) % lds_floats] == -1.f) sink[0] = 1.f;
    if constexpr (SCRATCH > 0) {
        if (priv[(threadIdx.x + (int)ticks) % SCRATCH] == -1.f) sink[1] = 1.f;
    }
}

template <int THREADS, int SCRATCH = 0, int VREGS = 0>
float run(int grid, size_t lds_bytes, float *sink, unsigned *where) {
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spin<THREADS, SCRATCH, VREGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((spin<THREADS, SCRATCH, VREGS>), dim3(grid), dim3(THREADS), lds_bytes, nullptr, 5000ull, (int)(lds_bytes / 4), sink, where);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        float t = 0.f;
        CHK(hipEventElapsedTime(&t, e0, e1));
        if (rep && t < best) best = t;
    }
    CHK(hipEventDestroy(e0));
    CHK(hipEventDestroy(e1));
    return best;
}

int main() {
    int dev = 0, cus = 0;
    CHK(hipGetDevice(&dev));
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float *sink;
    unsigned *where;
    CHK(hipMalloc(&sink, 64));
    CHK(hipMalloc(&where, 2 * 4096 * sizeof(unsigned)));
    printf("%d CUs; every workgroup spins 50 us\n", cus);
    for (size_t kb : {151, 96, 80, 64, 32}) {
        for (int threads : {512, 256}) {
            printf("LDS %3zu KB, %d threads:", kb, threads);
            int last_one_round = 0;
            for (int grid = 64; grid <= 1100; grid += 8) {
                const float ms = threads == 512 ? run<512>(grid, kb * 1024, sink, where) : run<256>(grid, kb * 1024, sink, where);
                if (ms < 0.085f) last_one_round = grid;
            }
            printf("  largest grid that ends within one spin: %d workgroups\n", last_one_round);
            fflush(stdout);
        }
    }
    // the same shape with private memory per lane (a kernel that spills): 10 floats = 40 B, the shipped r = 20 build's; 75 = 300 B
    for (int scratch : {10, 75}) {
        printf("LDS 151 KB, 512 threads, %3d B of scratch per lane:", 4 * scratch);
        int last_one_round = 0;
        for (int grid = 64; grid <= 300; grid += 4) {
            const float ms = scratch == 10 ? run<512, 10>(grid, 151 * 1024, sink, where) : run<512, 75>(grid, 151 * 1024, sink, where);
            if (ms < 0.085f) last_one_round = grid;
        }
        printf("  largest grid that ends within one spin: %d workgroups\n", last_one_round);
        fflush(stdout);
    }
    // ... and with the filter's register count: 8 waves x 256 VGPRs = every register of a CU
    for (int vregs : {128, 248, 256}) {
        printf("LDS 151 KB, 512 threads, %3d VGPRs:", vregs);
        int last_one_round = 0;
        for (int grid = 64; grid <= 300; grid += 4) {
            const float ms = vregs == 128 ? run<512, 0, 128>(grid, 151 * 1024, sink, where) : vregs == 248 ? run<512, 0, 248>(grid, 151 * 1024, sink, where) : run<512, 0, 256>(grid, 151 * 1024, sink, where);
            if (ms < 0.085f) last_one_round = grid;
        }
        printf("  largest grid that ends within one spin: %d workgroups\n", last_one_round);
        fflush(stdout);
    }
    // everything at once (the filter's footprint), and WHEN every workgroup starts
    for (int grid : {200, 208, 232, 255, 256}) {
        CHK(hipMemset(where, 0xff, 2 * 4096 * sizeof(unsigned)));
        const float ms = run<512, 10, 256>(grid, 151 * 1024, sink, where);
        std::vector<unsigned> w(2 * grid);
        CHK(hipMemcpy(w.data(), where, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        unsigned t_min = ~0u;
        for (int b = 0; b < grid; b++) t_min = std::min(t_min, w[2 * b] >> 8);
        int late = 0;
        for (int b = 0; b < grid; b++) late += ((w[2 * b] >> 8) - t_min) > 2000u ? 1 : 0;
        printf("151 KB + 256 VGPRs + 48 B scratch, %d workgroups: %.3f ms, %d start more than 20 us after the first\n", grid, ms, late);
    }
    // where do 256 workgroups of the filter's shape go?
    CHK(hipMemset(where, 0xff, 2 * 4096 * sizeof(unsigned)));
    run<512>(256, 151 * 1024, sink, where);
    std::vector<unsigned> h(2 * 256);
    CHK(hipMemcpy(h.data(), where, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    for (int b = 0; b < 256; b++) per_xcc[h[2 * b] & 15]++;
    for (int b = 0; b < 256; b++) h[2 * b] &= 15u;
    printf("256 workgroups (151 KB, 512 threads) by XCC_ID:");
    for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
    printf("\nfirst 24 (xcc, hw_id):");
    for (int b = 0; b < 24; b++) printf(" (%u,%08x)", h[2 * b], h[2 * b + 1]);
    printf("\n");
    return 0;
}
