// What does it cost to hand a per-lane partial sum to another lane's pixel on gfx950?
// Candidates for the q-side scatter of the pair-symmetric window filter:
//   mode 0  v_fma_f32 only (the VALU background: FMAS fmas per block)
//   mode 1  background + 16 ds_add_f32 (no return), consecutive lanes -> consecutive dwords
//   mode 2  background + 16 ds_add_f32, lane stride 4 dwords (2-way bank conflict)
//   mode 3  background + 16 v_add_f32 with a DPP wave_shl:1 source (register rotation)
//   mode 4  16 ds_add_f32 alone (LDS atomic issue rate)
//   mode 5  16 ds_read_b128 alone (for scale)
//   mode 6  background + 16 ds_read_b32 + 16 v_add + 16 ds_write_b32 (read-modify-write instead of the atomic)
// Reports ns per block per wave (8 waves per CU = 2 per SIMD, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 4096;
constexpr int FMAS = 256;  // background VALU instructions per block

template <int MODE>
__global__ __launch_bounds__(512) void kern(float *out, float a, float b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 0.f;
    __syncthreads();
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // each wave scatters into its own 2048-dword row
    const unsigned base = (wave * 2048 + (MODE == 2 ? 4 * lane : lane)) * 4;
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 3 || MODE == 6) {
#pragma unroll
            for (int r = 0; r < FMAS / 16; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
        if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(base), "v"(x[i]), "n"(i * (MODE == 2 ? 16 : 256)) : "memory");
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(a));
        }
        if (MODE == 5) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f t[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i]) : "v"(base * 4u), "n"((g * 4 + i) * 1024 % 32768) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; i++) x[g * 4 + i] += t[i].x;
            }
        }
        if (MODE == 6) {
            float t[16];
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[i]) : "v"(base), "n"(i * 256) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) {
                t[i] += x[i];
                asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(t[i]), "n"(i * 256) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float s = lds[threadIdx.x] + lds[threadIdx.x + 2048];
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name) {
    const int threads = 512, blocks = 256;
    float *out;
    CHK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 128 * 1024, 0, out, 1.0001f, 1e-9f);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
    }
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f ns per block per wave-pair (SIMD)\n", name, ms, ms * 1e6 / ITERS);
    CHK(hipFree(out));
}

int main() {
    run<0>("fma x256");
    run<1>("fma x256 + 16 ds_add_f32 (conflict-free)");
    run<2>("fma x256 + 16 ds_add_f32 (lane stride 4)");
    run<3>("fma x256 + 16 v_add_f32_dpp wave_shl:1");
    run<6>("fma x256 + 16 (ds_read, add, ds_write)");
    run<4>("16 ds_add_f32 alone");
    run<5>("16 ds_read_b128 alone");
    return 0;
}
