// scratch: which hipMemMap / hipMemSetAccess shapes does this runtime accept?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define T(x) do { hipError_t e = (x); printf("%-90s %s\n", #x, hipGetErrorString(e)); (void)hipGetLastError(); } while (0)
int main() {
    const size_t GiB = 1ull << 30, MiB = 1ull << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    char *raw = nullptr;
    T(hipMemAddressReserve((void **)&raw, 40 * GiB, GiB, nullptr, 0));
    char *va = (char *)(((uintptr_t)raw + GiB - 1) / GiB * GiB);
    int slot = 0;
    for (size_t tail : {2 * MiB, 4 * MiB, 8 * MiB, 32 * MiB, 128 * MiB}) {
        printf("---- tail %zu MiB\n", tail / MiB);
        hipMemGenericAllocationHandle_t ht, hb;
        T(hipMemCreate(&ht, tail, &prop, 0));
        T(hipMemCreate(&hb, GiB - tail, &prop, 0));
        char *s = va + (size_t)(slot++) * GiB;
        T(hipMemMap(s + GiB - tail, tail, 0, ht, 0));
        const bool tail_ok = hipMemSetAccess(s + GiB - tail, tail, &ad, 1) == hipSuccess;
        (void)hipGetLastError();
        printf("tail access: %s\n", tail_ok ? "ok" : "REFUSED");
        T(hipMemMap(s, GiB - tail, 0, hb, 0));
        T(hipMemSetAccess(s, GiB - tail, &ad, 1));
        T(hipMemset(s, 0, tail_ok ? GiB : GiB - tail));
        T(hipDeviceSynchronize());
        T(hipMemUnmap(s, GiB - tail));
        T(hipMemMap(s, GiB - tail, 0, hb, 0));
        T(hipMemSetAccess(s, GiB - tail, &ad, 1));
    }
    printf("---- many 32 MiB tails first, bodies afterwards\n");
    hipMemGenericAllocationHandle_t tails[8], hb;
    for (int j = 0; j < 8; j++) {
        hipMemCreate(&tails[j], 32 * MiB, &prop, 0);
        char *s = va + (size_t)(slot + j) * GiB;
        hipError_t e1 = hipMemMap(s + GiB - 32 * MiB, 32 * MiB, 0, tails[j], 0);
        hipError_t e2 = hipMemSetAccess(s + GiB - 32 * MiB, 32 * MiB, &ad, 1);
        printf("tail %d: map %s, access %s\n", j, hipGetErrorString(e1), hipGetErrorString(e2));
    }
    T(hipMemCreate(&hb, GiB - 32 * MiB, &prop, 0));
    for (int j = 0; j < 8; j++) {
        char *s = va + (size_t)(slot + j) * GiB;
        hipError_t e1 = hipMemMap(s, GiB - 32 * MiB, 0, hb, 0);
        hipError_t e2 = hipMemSetAccess(s, GiB - 32 * MiB, &ad, 1);
        hipError_t e3 = e2 == hipSuccess ? hipMemset(s, 0, GiB - 32 * MiB) : hipErrorUnknown;
        hipError_t e4 = hipDeviceSynchronize();
        hipError_t e5 = hipMemUnmap(s, GiB - 32 * MiB);
        printf("body at slot %d: map %s, access %s, memset %s, sync %s, unmap %s\n", j, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3), hipGetErrorString(e4), hipGetErrorString(e5));
        (void)hipGetLastError();
    }
    return 0;
}
