// Does a physical allocation mapped at a virtual address another one has just left reach the shaders?  (statmc_placement.hip: the
// reference slot's trade, holes of the range filled again after statmc_placement_trim, window slots used again.)
// hipcc --offload-arch=gfx950 -O2 -o vmm_remap tools/microbench/vmm_remap.hip && ./vmm_remap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void poke(unsigned *a, unsigned v) { a[0] = v; }
__global__ void peek(const unsigned *a, unsigned *out) { out[0] = a[0]; }
static unsigned *g_out;
static unsigned rd(const void *p) {
    unsigned h = 0;
    hipLaunchKernelGGL(peek, dim3(1), dim3(1), 0, 0, (const unsigned *)p, g_out);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, g_out, 4, hipMemcpyDeviceToHost);
    return h;
}
static void wr(void *p, unsigned v) {
    hipLaunchKernelGGL(poke, dim3(1), dim3(1), 0, 0, (unsigned *)p, v);
    (void)hipDeviceSynchronize();
}
static const size_t G = 1ull << 30;
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static int map_at(char *at, hipMemGenericAllocationHandle_t h) {
    CK(hipMemMap(at, G, 0, h, 0));
    CK(hipMemSetAccess(at, G, &acc, 1));
    return 0;
}
int main(int argc, char **argv) {
    const int scen = argc > 1 ? atoi(argv[1]) : 0;
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    char *raw = nullptr;
    CK(hipMemAddressReserve((void **)&raw, 9 * G, G, nullptr, 0));
    char *base = (char *)(((uintptr_t)raw + G - 1) / G * G);
    CK(hipMalloc(&g_out, 64));
    hipMemGenericAllocationHandle_t h0, h1;
    CK(hipMemCreate(&h0, G, &prop, 0));
    CK(hipMemCreate(&h1, G, &prop, 0));
    char *A = base, *W0 = base + 4 * G, *W1 = base + 5 * G;   // W0 / W1: witnesses -- the two allocations at addresses never used before
    if (map_at(A, h0)) return 1;
    wr(A + G - 64, 0xA0u);
    CK(hipMemUnmap(A, G));
    if (scen == 1) CK(hipDeviceSynchronize());
    if (scen == 2) { CK(hipMemRelease(h0)); }
    if (scen == 3) {   // the address range itself is given back and reserved again
        CK(hipMemAddressFree(raw, 9 * G));
        char *raw2 = nullptr;
        CK(hipMemAddressReserve((void **)&raw2, 9 * G, G, nullptr, 0));
        printf("range again at %p (was %p)\n", (void *)raw2, (void *)raw);
        raw = raw2;
        base = (char *)(((uintptr_t)raw + G - 1) / G * G);
        A = base; W0 = base + 4 * G; W1 = base + 5 * G;
    }
    if (scen == 4) {   // a kernel touches nothing of A between unmap and map, but a big other launch may flush translation caches
        void *big = nullptr;
        CK(hipMalloc(&big, 2 * G));
        CK(hipMemset(big, 1, 2 * G));
        CK(hipDeviceSynchronize());
        CK(hipFree(big));
    }
    if (scen == 5) {   // a small allocation made and freed
        void *small = nullptr;
        CK(hipMalloc(&small, 2 << 20));
        CK(hipFree(small));
    }
    if (scen == 6) {   // other memory touched, nothing freed
        void *big = nullptr;
        CK(hipMalloc(&big, 2 * G));
        CK(hipMemset(big, 1, 2 * G));
        CK(hipDeviceSynchronize());
    }
    if (scen == 7) {   // the range given back and asked for again AT THE SAME ADDRESS
        CK(hipMemAddressFree(raw, 9 * G));
        char *raw2 = nullptr;
        CK(hipMemAddressReserve((void **)&raw2, 9 * G, G, raw, 0));
        printf("range again at %p (was %p)\n", (void *)raw2, (void *)raw);
        raw = raw2;
        base = (char *)(((uintptr_t)raw + G - 1) / G * G);
        A = base; W0 = base + 4 * G; W1 = base + 5 * G;
    }
    if (scen == 8) {   // a kernel launch and a wait between unmap and map (no other memory traffic)
        wr(g_out + 8, 1u);
    }
    if (map_at(A, h1)) return 1;
    wr(A + G - 64, 0xB1u);          // lands in h1 if the new mapping is what the shaders see, in h0 otherwise
    if (scen != 2 && map_at(W0, h0)) return 1;
    if (map_at(W1, h1)) return 1;
    printf("scenario %d: A (now h1) reads %#x; h1 at a fresh address reads %#x (0xb1 = the write through A arrived there)", scen, rd(A + G - 64), rd(W1 + G - 64));
    if (scen != 2) printf("; h0 at a fresh address reads %#x (0xa0 = untouched)", rd(W0 + G - 64));
    printf("\n");
    return 0;
}
