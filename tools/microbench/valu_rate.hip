// VALU issue-rate microbenchmark for gfx950: how many cycles does a wave64 v_fma_f32 /
// v_pk_fma_f32 / v_exp_f32 / v_cmp+v_cndmask cost per SIMD at 1, 2, 4, 8 waves per SIMD?
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;  // independent accumulators

template <int MODE>
__global__ void kern(float *out, float a, float b) {
    float acc[UNROLL];
    v2f acc2[UNROLL / 2];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < UNROLL / 2; i++) acc2[i] = v2f{acc[2 * i], acc[2 * i + 1]};
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < UNROLL / 2; i++) acc2[i] = __builtin_elementwise_fma(acc2[i], v2f{a, a}, v2f{b, b});
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) acc[i] = __builtin_amdgcn_exp2f(acc[i]);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) acc[i] = (acc[i] <= a) ? acc[i] + b : 0.5f;  // cmp + add + cndmask
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) acc[i] = acc[i] - a;  // v_sub
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += acc[i];
#pragma unroll
    for (int i = 0; i < UNROLL / 2; i++) s += acc2[i].x + acc2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int ops_per_iter, int waves_per_simd) {
    const int threads = 64 * 4 * waves_per_simd;  // one block per CU, 4 SIMDs
    const int blocks = 256;
    float *out;
    CHK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    if (threads <= 1024) {
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        const double wave_instr_per_simd = (double)ITERS * ops_per_iter * waves_per_simd;
        const double ns_per_instr = ms * 1e6 / wave_instr_per_simd;
        printf("%-14s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name,
               waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4);
    }
    CHK(hipFree(out));
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", UNROLL, w);
        run<1>("v_pk_fma_f32", UNROLL / 2, w);
        run<2>("v_exp_f32", UNROLL, w);
        run<3>("cmp+add+cnd", UNROLL, w);
        run<4>("v_sub_f32", UNROLL, w);
    }
    return 0;
}
