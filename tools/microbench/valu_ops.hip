// Per-instruction VALU cost on gfx950 (ns per wave64 instruction per SIMD) for the op mix of the
// window filter's inner loop.  Inline asm so the compiler cannot fold anything.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 2048;
constexpr int U = 16;

#define BODY(ASM)                                                              \
    for (int it = 0; it < ITERS; it++) {                                       \
        _Pragma("unroll") for (int i = 0; i < U; i++) { ASM; }                 \
    }

template <int MODE>
__global__ void kern(float *out, float a, float b) {
    float x[U], y[U];
#pragma unroll
    for (int i = 0; i < U; i++) { x[i] = threadIdx.x * 1e-3f + i; y[i] = a + i; }
    if (MODE == 0) BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(b)))
    if (MODE == 1) BODY(asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" :: "v"(x[i]), "v"(y[i]) : "vcc"))
    if (MODE == 2) BODY(asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" :: "v"(x[i]), "v"(y[i]) : "s20", "s21"))
    if (MODE == 3) BODY(asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : ))
    if (MODE == 4) BODY(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(y[i])))
    if (MODE == 5) BODY(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(b)))
    if (MODE == 6) BODY(asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x[i])))
    if (MODE == 7) BODY(asm volatile("v_cmp_le_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc"))
    if (MODE == 8) BODY(asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])))
    if (MODE == 9) BODY(asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x[i]) : "v"(y[i]), "v"(b)))
    if (MODE == 10) BODY(asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])))
    if (MODE == 11) BODY(asm volatile("v_readlane_b32 s22, %0, 5" :: "v"(x[i]) : "s22"))
    if (MODE == 12) BODY(asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1\n\tv_cmp_le_f32_e64 s[22:23], %0, %2\n\ts_and_b64 s[20:21], s[20:21], s[22:23]\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(y[i]), "v"(b) : "s20", "s21", "s22", "s23"))
    if (MODE == 13) BODY(asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])))
    if (MODE == 14) BODY(asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])))
    if (MODE == 15) BODY(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(b)))
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f px[U], py[U];
#pragma unroll
    for (int i = 0; i < U; i++) { px[i] = v2f{x[i], x[i] + 1.f}; py[i] = v2f{y[i], y[i] * 0.5f}; }
    if (MODE == 16) BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(px[i]) : "v"(py[i])))
    if (MODE == 17) BODY(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(px[i]) : "v"(py[i])))
    if (MODE == 18) BODY(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(px[i]) : "v"(py[i])))
    if (MODE == 19) BODY(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %3, %3" : "+v"(px[i]), "+v"(x[i]) : "v"(py[i]), "v"(y[i])))
    float s = 0;
#pragma unroll
    for (int i = 0; i < U; i++) s += x[i] + px[i].x + px[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int instr_per_body) {
    for (int w : {1, 2, 4}) {
        const int threads = 64 * 4 * w, blocks = 256;
        float *out;
        CHK(hipMalloc(&out, (size_t)blocks * threads * 4));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        const double bodies = (double)ITERS * U * w;
        printf("%-34s waves/SIMD=%d  %.3f ns per body (%d instr)\n", name, w, ms * 1e6 / bodies, instr_per_body);
        CHK(hipFree(out));
    }
}

int main() {
    run<0>("v_fma_f32", 1);
    run<13>("v_sub_f32_e32", 1);
    run<8>("v_mul_f32_e32", 1);
    run<9>("v_fma_f32 clamp", 1);
    run<1>("v_cmp_le_f32_e32 vcc", 1);
    run<2>("v_cmp_le_f32_e64 sgpr", 1);
    run<3>("v_cndmask_b32_e32 vcc", 1);
    run<4>("v_cndmask_b32_e64 sgpr", 1);
    run<7>("cmp_e32 + cndmask_e32", 2);
    run<12>("2cmp_e64+s_and+nop+cndmask_e64", 5);
    run<5>("v_max3_f32", 1);
    run<15>("v_med3_f32", 1);
    run<14>("v_min_f32_e32", 1);
    run<10>("v_and_b32", 1);
    run<6>("v_exp_f32", 1);
    run<11>("v_readlane_b32", 1);
    run<16>("v_pk_fma_f32", 1);
    run<17>("v_pk_add_f32", 1);
    run<18>("v_pk_mul_f32", 1);
    run<19>("v_pk_fma_f32 + v_fma_f32", 2);
    return 0;
}
