// Round 5: which physical 1-GiB pieces of the card interfere with each other when one is READ as a stream and the other is
// WRITTEN beside it?  (tools/experiments/acc_pool.py fastslow: the accumulation runs 12 - 17 % faster when the sample arenas and
// the read-modify-write state lie in regions of different "class" -- whatever a class is physically; same class or not is
// what this program maps.)
// N physical allocations of 1 GiB (hipMemCreate) are mapped side by side into one reserved range.  probe(x, r) streams piece x
// with non-temporal 16-byte loads while every 16th step read-modify-writes a float4 of piece r (64 MiB of r per probe: ~6 %
// of the bytes); time per probe, best of 3.  Row k of the output: reference piece r_k (r_0 = piece 0, r_1 = the first piece
// that ran fast against r_0, r_2 = the first piece fast against both, ...), one character per piece: '=' slow against r_k
// (same class), '.' fast.
// Modes (second argument): 0 the class rows above; 1 the same with a copy as the probe (no clean levels); 2 pieces probed through ONE
// re-used address window, then side by side (+ a third argument: one piece at many addresses); 4 two pieces, addresses only;
// 5 (DO NOT RUN: hipMemSetAccess refuses 2-MiB mappings next to a larger one on ROCm 7.2, and touching such a range faults the GPU); 6 ONE physical GiB mapped at every slot; 7 the decider: one
// physical GiB mapped at 140 addresses, each mapping made after a fresh physical allocation, against those 140 allocations mapped
// in one burst -- the class follows the physical memory.
// hipcc -O3 --offload-arch=gfx950 rank_probe.hip -o rank_probe && ./rank_probe [pieces] [mode]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr size_t GiB = 1ull << 30;

#ifndef RMW_EVERY
#define RMW_EVERY 4
#endif
__global__ __launch_bounds__(256) void probe(const vfloat4 *x, vfloat4 *r, size_t n4, size_t r4, float *out) {
    vfloat4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t k = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride, k++) {
        acc += __builtin_nontemporal_load(x + i);
        if ((k & (RMW_EVERY - 1)) == 0) {
            const size_t j = (i / RMW_EVERY) % r4;
            vfloat4 v = r[j];
            v.x += 1.f;
            r[j] = v;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// copy: every float4 of x is written to r (as much written as read: the largest possible share of writes)
__global__ __launch_bounds__(256) void probe_copy(const vfloat4 *x, vfloat4 *r, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) r[i] = __builtin_nontemporal_load(x + i);
}

__global__ void fill(vfloat4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = vfloat4{1.f, 2.f, 3.f, 4.f};
}

int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 256;
    const bool copy_mode = argc > 2 && atoi(argv[2]) == 1;
    int dev = 0;
    CHK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t free_b, total_b;
    CHK(hipMemGetInfo(&free_b, &total_b));
    if ((size_t)n * GiB > free_b - 4 * GiB) n = (int)((free_b - 4 * GiB) / GiB);
    printf("granularity %zu B, free %.1f GiB, pieces %d\n", gran, free_b / (double)GiB, n);
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    if (mode == 7) {
        // Page tables or data?  Slot i gets a mapping of the ONE physical GiB D, made right after a fresh physical GiB F_i was
        // created (so that the mappings' page-table pages are placed at different moments of the allocator's life, the data being
        // the same everywhere); afterwards F_i is mapped at slot N + i (own data, page tables placed in one burst).
        const int slots = n / 2;
        char *raw = nullptr;
        CHK(hipMemAddressReserve((void **)&raw, (size_t)(2 * slots + 3) * GiB, GiB, nullptr, 0));
        char *va = (char *)(((uintptr_t)raw + GiB - 1) / GiB * GiB);
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        hipMemGenericAllocationHandle_t hd, ht;
        CHK(hipMemCreate(&ht, GiB, &prop, 0));
        char *target = va + (size_t)(2 * slots + 1) * GiB;
        CHK(hipMemMap(target, GiB, 0, ht, 0));
        CHK(hipMemSetAccess(target, GiB, &ad, 1));
        CHK(hipMemset(target, 0, 64ull << 20));
        CHK(hipMemCreate(&hd, GiB, &prop, 0));
        float *out2;
        CHK(hipMalloc(&out2, 64));
        hipEvent_t a0, a1;
        CHK(hipEventCreate(&a0));
        CHK(hipEventCreate(&a1));
        auto probe2 = [&](const char *x) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHK(hipEventRecord(a0));
                hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)x, (vfloat4 *)target, GiB / 16, (64ull << 20) / 16, out2);
                CHK(hipEventRecord(a1));
                CHK(hipEventSynchronize(a1));
                float ms;
                CHK(hipEventElapsedTime(&ms, a0, a1));
                if (rep > 0 && ms < best) best = ms;
            }
            return best;
        };
        std::vector<hipMemGenericAllocationHandle_t> f(slots);
        for (int i = 0; i < slots; i++) {
            CHK(hipMemCreate(&f[i], GiB, &prop, 0));
            CHK(hipMemMap(va + (size_t)i * GiB, GiB, 0, hd, 0));
            CHK(hipMemSetAccess(va + (size_t)i * GiB, GiB, &ad, 1));
        }
        std::vector<float> pd(slots), pf(slots);
        for (int i = 0; i < slots; i++) pd[i] = probe2(va + (size_t)i * GiB);
        for (int i = 0; i < slots; i++) {
            CHK(hipMemMap(va + (size_t)(slots + i) * GiB, GiB, 0, f[i], 0));
            CHK(hipMemSetAccess(va + (size_t)(slots + i) * GiB, GiB, &ad, 1));
        }
        for (int i = 0; i < slots; i++) pf[i] = probe2(va + (size_t)(slots + i) * GiB);
        printf("slot i = the same physical GiB D, mapped right after physical GiB F_i was created:\n  ");
        for (int i = 0; i < slots; i++) putchar(pd[i] > 0.185f ? '=' : '.');
        printf("\nslot N + i = F_i itself, all mapped afterwards in one go:\n  ");
        for (int i = 0; i < slots; i++) putchar(pf[i] > 0.185f ? '=' : '.');
        printf("\n");
        return 0;
    }
    if (mode == 6) {
        // ONE physical GiB (D) mapped at many slots at once: does every mapping have a class of its own?  Does the class of a slot
        // survive exchanging D for a fresh physical GiB?  Does it change when another mapping is made between unmap and map?
        const int slots = n, spare = 16;
        char *raw = nullptr;
        CHK(hipMemAddressReserve((void **)&raw, (size_t)(slots + spare + 3) * GiB, GiB, nullptr, 0));
        char *va = (char *)(((uintptr_t)raw + GiB - 1) / GiB * GiB);
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        hipMemGenericAllocationHandle_t hd, ht;
        CHK(hipMemCreate(&hd, GiB, &prop, 0));
        CHK(hipMemCreate(&ht, GiB, &prop, 0));
        char *target = va + (size_t)(slots + spare + 1) * GiB;
        CHK(hipMemMap(target, GiB, 0, ht, 0));
        CHK(hipMemSetAccess(target, GiB, &ad, 1));
        CHK(hipMemset(target, 0, 64ull << 20));
        float *out2;
        CHK(hipMalloc(&out2, 64));
        hipEvent_t a0, a1;
        CHK(hipEventCreate(&a0));
        CHK(hipEventCreate(&a1));
        auto probe2 = [&](const char *x) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHK(hipEventRecord(a0));
                hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)x, (vfloat4 *)target, GiB / 16, (64ull << 20) / 16, out2);
                CHK(hipEventRecord(a1));
                CHK(hipEventSynchronize(a1));
                float ms;
                CHK(hipEventElapsedTime(&ms, a0, a1));
                if (rep > 0 && ms < best) best = ms;
            }
            return best;
        };
        std::vector<float> p1(slots), p1b(slots), p2(slots, 0.f), p3(slots, 0.f);
        for (int j = 0; j < slots; j++) {
            hipError_t e = hipMemMap(va + (size_t)j * GiB, GiB, 0, hd, 0);
            if (e != hipSuccess) { printf("mapping the same allocation a second time: %s (slot %d)\n", hipGetErrorString(e), j); return 0; }
            CHK(hipMemSetAccess(va + (size_t)j * GiB, GiB, &ad, 1));
        }
        for (int j = 0; j < slots; j++) p1[j] = probe2(va + (size_t)j * GiB);
        for (int j = 0; j < slots; j++) p1b[j] = probe2(va + (size_t)j * GiB);
        auto show = [&](const char *what, const std::vector<float> &t) {
            printf("%s\n  ", what);
            for (int j = 0; j < slots; j++) putchar(t[j] == 0.f ? ' ' : t[j] > 0.185f ? '=' : '.');
            putchar('\n');
            fflush(stdout);
        };
        show("one physical GiB mapped at every slot:", p1);
        show("the same again:", p1b);
        // pick 8 slots of each class for the plain exchange (A) and 8 of each for the exchange with a mapping in between (B)
        std::vector<int> same, apart;
        for (int j = 0; j < slots; j++) (p1[j] > 0.185f ? same : apart).push_back(j);
        printf("%zu slots of the target's class, %zu apart\n", same.size(), apart.size());
        std::vector<int> setA, setB;
        for (int k = 0; k < 8; k++) {
            if (k < (int)same.size()) setA.push_back(same[k]);
            if (k < (int)apart.size()) setA.push_back(apart[k]);
            if (k + 8 < (int)same.size()) setB.push_back(same[k + 8]);
            if (k + 8 < (int)apart.size()) setB.push_back(apart[k + 8]);
        }
        for (int j : setA) {
            hipMemGenericAllocationHandle_t f;
            CHK(hipMemCreate(&f, GiB, &prop, 0));
            CHK(hipMemUnmap(va + (size_t)j * GiB, GiB));
            CHK(hipMemMap(va + (size_t)j * GiB, GiB, 0, f, 0));
            CHK(hipMemSetAccess(va + (size_t)j * GiB, GiB, &ad, 1));
            p2[j] = probe2(va + (size_t)j * GiB);
        }
        show("A: D exchanged for a fresh physical GiB (unmap, map):", p2);
        std::vector<float> p4(slots, 0.f);
        int k = 0;
        for (int j : setB) {
            if (k >= spare) break;
            hipMemGenericAllocationHandle_t f;
            CHK(hipMemCreate(&f, GiB, &prop, 0));
            CHK(hipMemUnmap(va + (size_t)j * GiB, GiB));
            CHK(hipMemMap(va + (size_t)(slots + k) * GiB, GiB, 0, hd, 0));
            CHK(hipMemSetAccess(va + (size_t)(slots + k) * GiB, GiB, &ad, 1));
            CHK(hipMemMap(va + (size_t)j * GiB, GiB, 0, f, 0));
            CHK(hipMemSetAccess(va + (size_t)j * GiB, GiB, &ad, 1));
            p3[j] = probe2(va + (size_t)j * GiB);
            p4[j] = probe2(va + (size_t)(slots + k) * GiB);
            k++;
        }
        show("B: the same with ANOTHER mapping made between unmap and map:", p3);
        show("   ... and the class of that other mapping (shown under the slot):", p4);
        return 0;
    }
    if (mode == 5 && !(argc > 4 && atoi(argv[4]) == 1)) {
        printf("mode 5 is kept for the record only: hipMemSetAccess refuses the 2-MiB tail mappings it needs (ROCm 7.2), and touching a range\n"
               "whose access was not set faults the GPU.  (A fifth argument of 1 runs it anyway.)\n");
        return 0;
    }
    if (mode == 5) {
        // Is a slot's class a property of its PAGE-TABLE page?  GiB-aligned slots; every slot keeps a 2-MiB handle mapped at its
        // tail (which keeps the slot's page-directory page alive); ONE probe piece of 1022 MiB visits the slots in turn.
        const size_t MiB2 = 2ull << 20, body = GiB - MiB2;
        const int slots = n;
        char *raw = nullptr;
        CHK(hipMemAddressReserve((void **)&raw, (size_t)(slots + 1) * GiB, GiB, nullptr, 0));
        const size_t shift = argc > 3 ? (size_t)atoi(argv[3]) * (1ull << 20) : 0;      // MiB off the GiB grid
        char *va = (char *)(((uintptr_t)raw + GiB - 1) / GiB * GiB) + shift;
        if (va + (size_t)slots * GiB > raw + (size_t)(slots + 1) * GiB) va -= GiB;
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        std::vector<hipMemGenericAllocationHandle_t> tail(slots);
        hipMemGenericAllocationHandle_t hr, hc;
        CHK(hipMemCreate(&hr, body, &prop, 0));
        CHK(hipMemCreate(&hc, body, &prop, 0));
        float *out2;
        CHK(hipMalloc(&out2, 64));
        hipEvent_t a0, a1;
        CHK(hipEventCreate(&a0));
        CHK(hipEventCreate(&a1));
        auto probe2 = [&](const char *x, char *r) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                CHK(hipEventRecord(a0));
                hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)x, (vfloat4 *)r, body / 16, (64ull << 20) / 16, out2);
                CHK(hipEventRecord(a1));
                CHK(hipEventSynchronize(a1));
                float ms;
                CHK(hipEventElapsedTime(&ms, a0, a1));
                if (rep > 0 && ms < best) best = ms;
            }
            return best;
        };
        auto map_tails = [&](bool reverse) {
            for (int k = 0; k < slots; k++) {
                const int j = reverse ? slots - 1 - k : k;
                CHK(hipMemMap(va + (size_t)j * GiB + body, MiB2, 0, tail[j], 0));   // (access is set with the slot's body: a 2-MiB range alone is refused)
            }
        };
        for (int j = 0; j < slots; j++) CHK(hipMemCreate(&tail[j], MiB2, &prop, 0));
        map_tails(false);
        CHK(hipMemMap(va, body, 0, hr, 0));
        CHK(hipMemSetAccess(va, body, &ad, 1));
        { hipError_t e = hipMemSetAccess(va, GiB, &ad, 1); if (e != hipSuccess) { printf("(slot-wide SetAccess: %s)\n", hipGetErrorString(e)); (void)hipGetLastError(); } }
        printf("slots at %p (reserved at %p), %d slots\n", (void *)va, (void *)raw, slots);
        int n_slot_wide_failed = 0;
        auto pass = [&](const char *what) {
            printf("%s\n  #", what);
            for (int j = 1; j < slots; j++) {
                char *c = va + (size_t)j * GiB;
                CHK(hipMemMap(c, body, 0, hc, 0));
                CHK(hipMemSetAccess(c, body, &ad, 1));
                if (hipMemSetAccess(c, GiB, &ad, 1) != hipSuccess) { (void)hipGetLastError(); n_slot_wide_failed++; }
                const float ms = probe2(c, va);
                CHK(hipMemUnmap(c, body));
                putchar(ms > 0.184f ? '=' : '.');
            }
            printf("  (slot-wide SetAccess refused %d times)\n", n_slot_wide_failed);
            fflush(stdout);
        };
        pass("pass 1 (tails mapped in slot order)");
        pass("pass 2 (the same again)");
        for (int j = 0; j < slots; j++) CHK(hipMemUnmap(va + (size_t)j * GiB + body, MiB2));
        CHK(hipMemUnmap(va, body));
        map_tails(true);
        CHK(hipMemMap(va, body, 0, hr, 0));
        CHK(hipMemSetAccess(va, body, &ad, 1));
        { hipError_t e = hipMemSetAccess(va, GiB, &ad, 1); if (e != hipSuccess) { printf("(slot-wide SetAccess: %s)\n", hipGetErrorString(e)); (void)hipGetLastError(); } }
        pass("pass 3 (every mapping removed, tails mapped again in reverse order)");
        pass("pass 4 (the same again)");
        return 0;
    }
    if (mode == 4) {
        // VIRTUAL addresses only: two physical pieces; the reference piece at one of a few addresses, the candidate piece at
        // V0 + j GiB for every j; then a fine sweep (32 MiB steps) across the first transition
        const int span = n;                       // GiB of address space swept
        char *va = nullptr;
        CHK(hipMemAddressReserve((void **)&va, (size_t)(span + 2) * GiB, GiB, nullptr, 0));
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        hipMemGenericAllocationHandle_t hr, hc;
        CHK(hipMemCreate(&hr, GiB, &prop, 0));
        CHK(hipMemCreate(&hc, GiB, &prop, 0));
        float *out2;
        CHK(hipMalloc(&out2, 64));
        hipEvent_t a0, a1;
        CHK(hipEventCreate(&a0));
        CHK(hipEventCreate(&a1));
        auto probe2 = [&](const char *x, char *r) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                CHK(hipEventRecord(a0));
                hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)x, (vfloat4 *)r, GiB / 16, (64ull << 20) / 16, out2);
                CHK(hipEventRecord(a1));
                CHK(hipEventSynchronize(a1));
                float ms;
                CHK(hipEventElapsedTime(&ms, a0, a1));
                if (rep > 0 && ms < best) best = ms;
            }
            return best;
        };
        printf("address range at %p, %d GiB\n", (void *)va, span);
        const int refs_at[4] = {0, 3, span / 2 + 1, span - 7};
        int first_slow = -1;
        for (int q = 0; q < 4; q++) {
            char *r = va + (size_t)refs_at[q] * GiB;
            CHK(hipMemMap(r, GiB, 0, hr, 0));
            CHK(hipMemSetAccess(r, GiB, &ad, 1));
            printf("reference at + %d GiB:\n  ", refs_at[q]);
            for (int j = 0; j < span; j++) {
                if (j == refs_at[q]) { putchar('#'); continue; }
                char *c = va + (size_t)j * GiB;
                CHK(hipMemMap(c, GiB, 0, hc, 0));
                CHK(hipMemSetAccess(c, GiB, &ad, 1));
                const float ms = probe2(c, r);
                CHK(hipMemUnmap(c, GiB));
                putchar(ms > 0.185f ? '=' : '.');
                if (q == 0 && ms > 0.185f && first_slow < 0 && j > 4) first_slow = j;
            }
            putchar('\n');
            fflush(stdout);
            if (q == 0 && first_slow > 1) {
                printf("fine sweep, reference at + 0: candidate at + %d GiB - 1.5 GiB + k x 64 MiB:\n  ", first_slow);
                for (int k = 0; k < 48; k++) {
                    char *c = va + (size_t)first_slow * GiB - 3 * (GiB / 2) + (size_t)k * (64ull << 20);
                    CHK(hipMemMap(c, GiB, 0, hc, 0));
                    CHK(hipMemSetAccess(c, GiB, &ad, 1));
                    printf(" %.3f", probe2(c, r));
                    CHK(hipMemUnmap(c, GiB));
                }
                putchar('\n');
            }
            CHK(hipMemUnmap(r, GiB));
        }
        return 0;
    }
    if (mode == 2) {
        // the allocator's sequence (statmc_placement.hip): some plain memory first, a reference piece, then every new piece is
        // created, mapped into a probe window, probed and unmapped again; afterwards all of them are mapped side by side and
        // probed once more
        void *plain = nullptr;
        CHK(hipMalloc(&plain, 24 * GiB));
        char *win = nullptr, *all = nullptr;
        CHK(hipMemAddressReserve((void **)&win, 2 * GiB, GiB, nullptr, 0));
        CHK(hipMemAddressReserve((void **)&all, (size_t)n * GiB, GiB, nullptr, 0));
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        std::vector<hipMemGenericAllocationHandle_t> hh(n);
        CHK(hipMemCreate(&hh[0], GiB, &prop, 0));
        CHK(hipMemMap(win, GiB, 0, hh[0], 0));
        CHK(hipMemSetAccess(win, GiB, &ad, 1));
        CHK(hipMemset(win, 0, 64ull << 20));
        float *out2;
        CHK(hipMalloc(&out2, 64));
        hipEvent_t a0, a1;
        CHK(hipEventCreate(&a0));
        CHK(hipEventCreate(&a1));
        auto probe2 = [&](const char *x, char *r) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                CHK(hipEventRecord(a0));
                hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)x, (vfloat4 *)r, GiB / 16, (64ull << 20) / 16, out2);
                CHK(hipEventRecord(a1));
                CHK(hipEventSynchronize(a1));
                float ms;
                CHK(hipEventElapsedTime(&ms, a0, a1));
                if (rep > 0 && ms < best) best = ms;
            }
            return best;
        };
        std::vector<float> inc(n, 0.f), again(n, 0.f);
        for (int i = 1; i < n; i++) {
            CHK(hipMemCreate(&hh[i], GiB, &prop, 0));
            CHK(hipMemMap(win + GiB, GiB, 0, hh[i], 0));
            CHK(hipMemSetAccess(win + GiB, GiB, &ad, 1));
            inc[i] = probe2(win + GiB, win);
            CHK(hipMemUnmap(win + GiB, GiB));
        }
        for (int i = 1; i < n; i++) {
            CHK(hipMemMap(all + (size_t)i * GiB, GiB, 0, hh[i], 0));
        }
        CHK(hipMemSetAccess(all + GiB, (size_t)(n - 1) * GiB, &ad, 1));
        for (int i = 1; i < n; i++) again[i] = probe2(all + (size_t)i * GiB, win);
        if (argc > 3) {
            // one physical piece at many virtual addresses: unmap everything, then for a few pieces c map c at all + j GiB, probe, unmap
            CHK(hipMemUnmap(all + GiB, (size_t)(n - 1) * GiB));
            const int cs[4] = {1, 5, 40, n - 1};
            for (int q = 0; q < 4; q++) {
                printf("piece %3d at all + j GiB, j = 0 .. %d:\n ", cs[q], n - 1);
                for (int j = 0; j < n; j++) {
                    CHK(hipMemMap(all + (size_t)j * GiB, GiB, 0, hh[cs[q]], 0));
                    CHK(hipMemSetAccess(all + (size_t)j * GiB, GiB, &ad, 1));
                    printf(" %.3f", probe2(all + (size_t)j * GiB, win));
                    CHK(hipMemUnmap(all + (size_t)j * GiB, GiB));
                }
                printf("\n");
            }
            printf("win %p all %p\n", (void *)win, (void *)all);
        }
        printf("incremental (create, map, probe, unmap):\n ");
        for (int i = 1; i < n; i++) printf(" %.3f", inc[i]);
        printf("\nall mapped, probed again:\n ");
        for (int i = 1; i < n; i++) printf(" %.3f", again[i]);
        printf("\n");
        return 0;
    }
    char *base = nullptr;
    CHK(hipMemAddressReserve((void **)&base, (size_t)n * GiB, GiB, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < n; i++) {
        CHK(hipMemCreate(&h[i], GiB, &prop, 0));
        CHK(hipMemMap(base + (size_t)i * GiB, GiB, 0, h[i], 0));
    }
    CHK(hipMemSetAccess(base, (size_t)n * GiB, &acc, 1));
    float *out;
    CHK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (vfloat4 *)base, (size_t)n * GiB / 16);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto run = [&](int x, int r) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CHK(hipEventRecord(e0));
            if (copy_mode) hipLaunchKernelGGL(probe_copy, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)(base + (size_t)x * GiB), (vfloat4 *)(base + (size_t)r * GiB), GiB / 16);
            else hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, (const vfloat4 *)(base + (size_t)x * GiB), (vfloat4 *)(base + (size_t)r * GiB), GiB / 16,
                               (64ull << 20) / 16, out);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    };
    std::vector<int> refs = {0};
    std::vector<std::vector<float>> rows;
    for (size_t k = 0; k < refs.size() && k < 6; k++) {
        std::vector<float> t(n);
        float lo = 1e9f, hi = 0.f;
        for (int x = 0; x < n; x++) {
            t[x] = x == refs[k] ? 0.f : run(x, refs[k]);
            if (x != refs[k]) { lo = t[x] < lo ? t[x] : lo; hi = t[x] > hi ? t[x] : hi; }
        }
        const float cut = 0.5f * (lo + hi);
        {   // the distribution: sorted times, every 8th
            std::vector<float> srt;
            for (int x = 0; x < n; x++) if (x != refs[k]) srt.push_back(t[x]);
            std::sort(srt.begin(), srt.end());
            printf("  sorted:");
            for (size_t q = 0; q < srt.size(); q += 8) printf(" %.3f", srt[q]);
            printf(" %.3f\n", srt.back());
        }
        printf("ref %3d: %.3f .. %.3f ms per GiB (%.2f .. %.2f TB/s), cut %.3f\n  ", refs[k], lo, hi, GiB / hi / 1e9, GiB / lo / 1e9, cut);
        for (int x = 0; x < n; x++) putchar(x == refs[k] ? '#' : t[x] > cut ? '=' : '.');
        putchar('\n');
        fflush(stdout);
        rows.push_back(t);
        if (hi / lo < 1.04f) { printf("  (no contrast against this reference)\n"); break; }
        // next reference: the first piece that ran fast against every reference so far
        int next = -1;
        for (int x = 0; x < n && next < 0; x++) {
            bool fast_all = true;
            for (size_t q = 0; q < rows.size(); q++) {
                float l = 1e9f, hh = 0.f;
                for (int y = 0; y < n; y++) if (y != refs[q]) { l = rows[q][y] < l ? rows[q][y] : l; hh = rows[q][y] > hh ? rows[q][y] : hh; }
                if (x == refs[q] || rows[q][x] > 0.5f * (l + hh)) fast_all = false;
            }
            if (fast_all) next = x;
        }
        if (next < 0) break;
        refs.push_back(next);
    }
    // a few raw numbers for the record
    printf("times against ref 0 (ms), first 48 pieces:");
    for (int x = 0; x < n && x < 48; x++) printf(" %.3f", rows[0][x]);
    printf("\n");
    return 0;
}
