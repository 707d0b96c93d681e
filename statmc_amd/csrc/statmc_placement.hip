// statmc_placement.hip -- device memory dealt by interference class (statmc_malloc_placed, include/statmc.h).
//
// What this is for.  accumulate_kernel streams a read-once sample arena (44 B per pixel and sample) and read-modify-writes
// the running moments (224 B per pixel and launch).  On MI355X the price of those few writes depends on which memory
// holds the two.  Measured with everything carved out of one allocation (tools/experiments/acc_pool.py fastslow, all stat types):
//                                                        1080p / 256 spp      4K / 64 spp        1080p / 64 spp
//     arenas and state in GiB slots of the same class    3.92 ms  0.760       4.67 ms  0.675     1.095 ms  0.720   (of the 8 TB/s HBM peak)
//     arenas in one class, state in another              3.52 ms  0.846       3.99 ms  0.789     0.947 ms  0.832
// whichever of the two is where; same kernel, same bits.  What a "class" is (round 5's ledger):
//   * every GiB of device memory falls into one of three classes; about a third of the slots each, in runs of 4 .. 64 slots;
//     a read-only stream runs at the same rate from any of them -- only a stream READ beside WRITES into the same class pays
//     (9 % for a 25 % share of read-modify-writes: tools/microbench/rank_probe.hip, bimodal with an empty gap);
//   * the class travels with the PHYSICAL memory: one physical GiB mapped at 140 addresses -- each mapping made right after a
//     fresh physical allocation, i.e. with its page tables placed at another moment -- is in one class everywhere, while 140
//     distinct physical GiB mapped afterwards in one burst fall into all three classes, in runs (rank_probe mode 7).  (Two earlier
//     observations looked like the opposite -- pieces probed through ONE re-used address window all show one class, and show
//     different ones when mapped side by side later (mode 2) -- and are what one sees if a handle's backing is settled when it is
//     mapped and a window that is unmapped and mapped again gets the same memory back.)  Three equal classes, GiB-scale runs,
//     writes hurting reads of the same class only: the signature of the three ranks behind every channel of a 12-high HBM3E
//     stack, where a write followed by a read of the SAME rank pays the write-to-read turnaround inside the DRAM.  Nothing in
//     the HIP API says which rank a page is in, so this allocator MEASURES it.
//
// How.  One reserved address range per device, cut into GiB slots.  A slot is backed once, by its own 1-GiB physical
// allocation (hipMemCreate + hipMemMap; it keeps that memory for good), and probed IN PLACE against slot 0, which the allocator keeps for itself: the probe
// kernel streams the slot with non-temporal loads while every fourth step read-modify-writes 16 bytes of slot 0 (0.175 ms
// against 0.192 ms; a GiB is beyond the 256 MiB Infinity Cache, so the probe reaches the DRAM).  The first slot found apart
// from slot 0 becomes the second probe target, which tells the other two classes apart.  Slots of slot 0's class (A) hold
// STATE blocks; STREAM blocks go to ONE of the other classes (B) as long as the card has room: arenas inside one class and
// state in another measured 0.842 of the HBM peak at 1080p / 256 spp, arenas spread over both other classes 0.818, everything
// in one class 0.760 (tools/experiments/acc_pool.py classes / fastslow).  A block never spans slots of an unsuitable class.  A
// large block is a WINDOW: as many suitable slots as it needs, wherever they lie in the range, mapped a second
// time side by side in a second reserved range (one physical allocation may be mapped at several addresses, and the class is
// the memory's, not the address's; blocks above 2 GiB) -- so a 6-GiB arena needs six class-B slots, not six in a row: on a card whose classes come
// in short runs the first version backed 170 GiB to find three runs of six and still put an arena into class C.  Slots are
// backed until there are enough OF THE WANTED CLASS or the search's byte budget is spent: 3 x what the device's callers have asked
// for so far (+ 6 GiB), or STATMC_PLACEMENT_MAX_GIB; never more than 60 % / 75 % of the card (round 6; until then the two card
// fractions were the only bounds: 101 GiB backed to place 23).  After that the third class, then both, then anything.  Smaller blocks
// are carved out of whole slots dealt to their role.  What no role uses stays mapped and idle until statmc_placement_trim gives it
// back to the driver (a host calls it once its buffers are dealt: bench.py after the warm-up, statmc::Estimator at its first Denoise).  No contrast between the probes, no
// virtual-memory support, too little memory: the call degrades to slots as they come -- placement is an optimisation, never
// a requirement -- and statmc_placement_info says so.  STATMC_PLACEMENT=0 turns the call into hipMalloc.

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "statmc_device.h"

namespace {

constexpr size_t kSlot = 1ull << 30;          // mapping, physical allocation and probe unit
constexpr size_t kBlock = 2ull << 20;         // granularity of the blocks handed out
constexpr size_t kReserveSlots = 640;         // address range per device (more than the card holds)
constexpr size_t kProbeWindow = 64ull << 20;  // bytes of slot 0 the probe writes
constexpr int kProbeEvery = 4;                // one 16-byte read-modify-write per this many 16-byte loads
// x the fastest probe: the two levels sit at 1.00 .. 1.03 and 1.075 .. 1.10; what lies between is a piece that straddles classes.
// STATMC_PLACEMENT_SAME / STATMC_PLACEMENT_APART (experiments) move the two thresholds.
constexpr float kContrast = 1.055f;          // calibration: both levels have been seen once two probes differ by this much
inline float env_threshold(const char *name, float dflt, float lo, float hi) {
    const char *e = getenv(name);
    const float x = e ? (float)atof(e) : 0.f;
    return x > lo && x < hi ? x : dflt;
}
inline float same_above() { static const float v = env_threshold("STATMC_PLACEMENT_SAME", 1.055f, 1.03f, 1.10f); return v; }
inline float apart_below() { static const float v = env_threshold("STATMC_PLACEMENT_APART", 1.035f, 1.0f, 1.055f); return v; }
// The moments of a film are a few hundred MB in ONE slot: a slot that straddles a class boundary (the runs of a class do not end
// on the allocator's GiB marks; such a slot probes between the levels, 1.058 in a sample of 114 whose slow cluster starts at
// 1.073) may hold them in its minority part -- in the arenas' class.  The state's first choice are slots well inside the slow cluster.
inline float state_above() { static const float v = env_threshold("STATMC_PLACEMENT_STATE", 1.07f, 1.055f, 1.10f); return v; }
#define kSameAbove same_above()
#define kApartBelow apart_below()
constexpr float kSelfContrast = 1.08f;      // calibration: slot 0 against itself is this far above the fast level
constexpr int kCalibrationCap = 96;           // slots backed without seeing both levels: no classes to tell apart here

typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef unsigned vuint4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void slot_probe_kernel(const vfloat4 *x, vuint4 *ref, size_t n4, size_t window4, float *sink) {
    vfloat4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t k = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride, k++) {
        acc += __builtin_nontemporal_load(x + i);
        if ((k & (kProbeEvery - 1)) == 0) {
            const size_t j = (i / kProbeEvery) % window4;
            vuint4 v = ref[j];
            v.x += 1u;
            ref[j] = v;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

enum SlotClass { kUnknown = -1, kClassA = 0, kClassB = 1, kClassC = 2, kMixed = 3, kNotA = 4 };   // A: slot 0's; B: the second target's; kNotA: B or C, not probed against the second target yet
constexpr int kPrivate = -2;  // Slot::role of the allocator's own slots (probe targets)
constexpr int kReleased = -3; // Slot::role of a slot whose memory went back to the driver (statmc_placement_trim): an address hole

struct Slot {
    hipMemGenericAllocationHandle_t handle;
    float probe_ms[2] = {0.f, 0.f};   // against slot 0 / against the second target
    int role = -1;            // -1: not dealt to a role yet; STATMC_MEM_STATE / STATMC_MEM_STREAM: its space belongs to that role's free list
    bool as_it_came = false;  // dealt to a role without the wanted class
    int window = -1;          // >= 0: part of a window block -- mapped a second time at that slot of the window range
};
struct Window {               // a block of whole slots mapped side by side in the window range
    size_t first, n, bytes;   // window slots [first, first + n), bytes asked for
    int role;
    bool wanted;              // every slot of it has the class the role asks for
};

struct Placement {
    bool init_tried = false, vmm = false, calibrated = false, no_contrast = false;
    hipMemAllocationProp prop;
    std::vector<hipMemAccessDesc> access;    // the owning device first, then the peers granted so far (placement_grant_peer)
    bool peers_granted = false;              // false: only the owner (no cross-device copy yet, or hipMemSetAccess refused the peers)
    char *base = nullptr;                    // GiB-aligned start of the slots (inside the reservation)
    float *sink = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<Slot> slots;                 // slot 0 is the allocator's own (probe target)
    int target2 = -1;                        // the second probe target: the first slot found apart from slot 0 (private too)
    size_t total_bytes = 0;                  // of the card
    float fastest_ms = 0.f, slowest_ms = 0.f;
    float level = 0.f;                       // fast_level(): what the class thresholds are multiples of
    size_t peak_backed = 0;                  // most slots backed at any one time (what a search held before statmc_placement_trim)
    bool rebase_tried = false, rebased = false;   // the reference slot moved out of the locally dominant class (calibrate)
    float self_ms = 0.f;                     // slot 0 streamed beside writes into ITSELF: what "the same class" costs, by construction
    int n_probes = 0;
    const char *last_note = "-";             // why the last attempt to back a slot ended (diagnostics)
    std::map<size_t, size_t> free_blocks[2]; // per role: offset from base -> bytes (coalesced; never across slots of another role)
    std::map<size_t, std::pair<size_t, int>> live;   // offset -> (bytes, role)
    bool slot0_dealt = false;                // slot 0 behind its probe window belongs to the STATE role's free list
    int stream_class = -1;                   // kClassB or kClassC once the first STREAM block has chosen: the arenas' class on this device
    size_t expect[2] = {0, 0};               // per role: bytes the caller has announced and not asked for yet (statmc_placement_expect)
    char *win_base = nullptr;                // the window range (GiB-aligned; nullptr: no second range, blocks need runs of slots)
    std::vector<int> win_slot;               // per window slot: the slot mapped there, -1 = free
    std::map<size_t, Window> windows;        // offset from win_base -> block
};

std::mutex g_place_mu;
std::unordered_map<int, Placement> g_place;   // per device

bool placement_disabled() {
    const char *e = getenv("STATMC_PLACEMENT");
    return e && e[0] == '0';
}

// The level every threshold is a multiple of: the smallest probe against slot 0 that has two companions within 2 % of it -- the bottom
// of the first TIGHT cluster.  Its history (round 6): the smallest probe alone was dragged down by one probe a few per cent too fast (a
// clock still ramping, the first launches of a process: the genuine fast cluster then read "between the levels", three processes in
// a row 0.81 | 0.74 | 0.77 of the HBM peak, profiles/r06_bench_d_*.json); the third-smallest probe was a SLOW one whenever the
// calibration ended on fewer than three fast slots (one fast slot, then a run of slot 0's class: every slot read "apart from slot 0",
// the run went to the first arena -- six of the arenas' 22 GiB in the moments' class, 0.746 where torch's allocator got 0.764,
// profiles/r06_bench_x.json), and the third-smallest of the probes below slot 0's own took slots that straddle two classes (1.045 x
// the fast level, 0.1886 / 0.1890 beside 0.1805) for the fast level (0.73, gpurun_out r06_bench_z3).  A single outlier has no companions;
// straddling slots are rare and scattered.  0: no such cluster yet (the calibration goes on).
constexpr float kTight = 1.02f;
float fast_level_of(const float *probes, size_t n, float /*self_ms*/) {
    std::vector<float> t;
    t.reserve(n);
    for (size_t i = 0; i < n; i++)
        if (probes[i] > 0.f) t.push_back(probes[i]);
    std::sort(t.begin(), t.end());
    for (size_t i = 0; i + 2 < t.size(); i++)
        if (t[i + 2] <= kTight * t[i]) return t[i];
    return 0.f;
}
float fast_level(const Placement &P) {
    std::vector<float> t;
    t.reserve(P.slots.size());
    for (size_t i = 1; i < P.slots.size(); i++) t.push_back(P.slots[i].probe_ms[0]);
    return fast_level_of(t.data(), t.size(), P.self_ms);
}

int classify(const Placement &P, const Slot &s) {
    if (P.no_contrast || !P.calibrated || s.probe_ms[0] <= 0.f) return kUnknown;
    const float level = P.level;
    if (s.probe_ms[0] > kSameAbove * level) return kClassA;
    if (s.probe_ms[0] >= kApartBelow * level) return kMixed;
    if (s.probe_ms[1] <= 0.f) return kNotA;
    if (s.probe_ms[1] > kSameAbove * level) return kClassB;
    if (s.probe_ms[1] < kApartBelow * level) return kClassC;
    return kMixed;
}

// slots that hold memory (the range's holes -- statmc_placement_trim -- do not count against a search's cap)
size_t backed_count(const Placement &P) {
    size_t n = 0;
    for (const Slot &s : P.slots) n += s.role != kReleased ? 1 : 0;
    return n;
}
// bytes in live blocks of both roles
size_t live_total(const Placement &P) {
    size_t b = 0;
    for (const auto &kv : P.live) b += kv.second.first;
    for (const auto &kv : P.windows) b += kv.second.n * kSlot;
    return b;
}
// How many slots the search for a CLASS may have backed, for a device whose callers hold `live_total` and now ask for `need` more:
// STATMC_PLACEMENT_MAX_GIB if set (> 0), else 3 x the bytes asked for + 6 slots (the allocator's own two and what the calibration
// needs to see both levels).  A third of a card's slots is of any one class, in runs of 4 .. 64, so 3 x is what ONE class for all the
// arenas takes on average; what the search backs beyond the request is idle only until statmc_placement_trim.  Measured, 1080p /
// 256 spp in the step: arenas in one class 0.805 of the HBM peak, spread over both classes apart from the moments' 0.77 (what budgets
// of 1.5 x and 2 x gave on three boxes of five), in the moments' own class 0.72 (profiles/r06h_bench.json, r06w_bench_under_rocprof.json,
// r06f_bench.json).
size_t budget_slots(const Placement &P, size_t need) {
    static const double env_gib = [] { const char *e = getenv("STATMC_PLACEMENT_MAX_GIB"); return e ? atof(e) : 0.0; }();
    if (env_gib > 0.0) return (size_t)env_gib;
    return (size_t)(3.0 * (double)(live_total(P) + need) / (double)kSlot + 0.999) + 6;
}

// hipMemUnmap leaves the shaders' translation of the range in place on this runtime (ROCm 7.2, gfx950): a different allocation mapped
// at an address another one has just left is NOT what kernels see there -- they go on reading and writing the memory that left, even
// after it has gone back to the driver -- until something makes the driver rewrite the process's page tables the ordinary way.  An
// allocation made and freed with hipMalloc / hipFree between the unmap and the next map does (tools/microbench/vmm_remap.hip,
// profiles/r06_vmm_remap.log: scenarios 0 - 8; a synchronisation, a kernel or giving the address range back and reserving it again do
// not).  Called after every batch of unmaps, before any of the addresses can be mapped again: the reference slot's trade (which had
// never taken effect: profiles/r06_rebase_check.log), statmc_placement_trim (holes are filled again), windows (their slots of the
// window range are used again).
hipError_t flush_translations() {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, 2u << 20);
    if (e == hipSuccess) e = hipFree(p);
    return e;
}

// grants the mapping at `at` to the owner and, where the runtime accepts it, to the peers (a block of this allocator is then a
// valid operand of statmc_copy_rect / statmc_halo_exchange between devices, like a hipMalloc block under hipDeviceEnablePeerAccess)
hipError_t set_access(Placement &P, void *at) {
    if (P.access.size() > 1) {
        if (hipMemSetAccess(at, kSlot, P.access.data(), P.access.size()) == hipSuccess) {
            P.peers_granted = true;
            return hipSuccess;
        }
        (void)hipGetLastError();
        P.access.resize(1);            // this runtime takes no peer descriptors: owner only from here on (statmc_placement_info says so)
        P.peers_granted = false;
    }
    return hipMemSetAccess(at, kSlot, P.access.data(), 1);
}

hipError_t probe_slot(Placement &P, size_t index, int which = 0) {
    const char *cand = P.base + index * kSlot;
    char *target = which == 0 ? P.base : P.base + (size_t)P.target2 * kSlot;
    float best = 1e30f;
    hipError_t err = hipSuccess;
    for (int rep = 0; rep < 6 && err == hipSuccess; rep++) {      // the first one is the warm-up (first touch of the slot)
        (void)hipEventRecord(P.e0, P.stream);
        hipLaunchKernelGGL(slot_probe_kernel, dim3(2048), dim3(256), 0, P.stream, reinterpret_cast<const vfloat4 *>(cand),
                           reinterpret_cast<vuint4 *>(target), kSlot / 16, kProbeWindow / 16, P.sink);
        (void)hipEventRecord(P.e1, P.stream);
        err = hipEventSynchronize(P.e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, P.e0, P.e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (err != hipSuccess) return err;
    P.slots[index].probe_ms[which] = best;
    P.n_probes++;
    if (getenv("STATMC_PLACEMENT_DEBUG")) fprintf(stderr, "statmc placement: slot %zu against target %d: %.4f ms\n", index, which, best);
    if (P.fastest_ms == 0.f || best < P.fastest_ms) P.fastest_ms = best;
    if (best > P.slowest_ms) P.slowest_ms = best;
    if (which == 0) P.level = fast_level(P);
    return hipSuccess;
}

// once both levels are known: the first slot apart from slot 0 becomes the second target (the allocator's own, like slot 0),
// and every other slot apart from slot 0 is probed against it -- B (its class) or C
hipError_t split_not_a(Placement &P) {
    if (P.no_contrast || !P.calibrated) return hipSuccess;
    if (P.target2 < 0) {
        for (size_t i = 1; i < P.slots.size() && P.target2 < 0; i++)
            if (P.slots[i].role == -1 && classify(P, P.slots[i]) == kNotA) P.target2 = (int)i;
        if (P.target2 < 0) return hipSuccess;
        P.slots[P.target2].role = kPrivate;
        if (hipError_t e = hipMemsetAsync(P.base + (size_t)P.target2 * kSlot, 0, kProbeWindow, P.stream); e != hipSuccess) return e;
    }
    for (size_t i = 1; i < P.slots.size(); i++)
        if ((int)i != P.target2 && P.slots[i].role != kReleased && classify(P, P.slots[i]) == kNotA)
            if (hipError_t e = probe_slot(P, i, 1); e != hipSuccess) return e;
    return hipSuccess;
}

// backs the next slot of the range with memory -- the first hole statmc_placement_trim left, else the slot behind the last one --
// and probes it (index > 0); false when the card or the range has no room left
bool back_next_slot(Placement &P, hipError_t *err, size_t leave_free = 512ull << 20) {
    *err = hipSuccess;
    size_t index = P.slots.size();
    for (size_t i = 1; i < P.slots.size(); i++)
        if (P.slots[i].role == kReleased) { index = i; break; }
    if (index >= kReserveSlots) {
        P.last_note = "address range used up";
        return false;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < kSlot + leave_free) {
        P.last_note = "the card is (nearly) full";
        return false;
    }
    Slot s;
    if (hipError_t e = hipMemCreate(&s.handle, kSlot, &P.prop, 0); e != hipSuccess) {
        (void)hipGetLastError();
        P.last_note = hipGetErrorString(e);
        return false;
    }
    char *at = P.base + index * kSlot;
    hipError_t e = hipMemMap(at, kSlot, 0, s.handle, 0);
    if (e == hipSuccess) {
        e = set_access(P, at);
        if (e != hipSuccess) (void)hipMemUnmap(at, kSlot);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(s.handle);
        *err = e;
        return false;
    }
    if (index == P.slots.size()) P.slots.push_back(s);
    else P.slots[index] = s;            // (a hole filled: undealt, unprobed, like a new slot)
    P.peak_backed = std::max(P.peak_backed, backed_count(P));
    if (index > 0) {
        hipError_t pe = probe_slot(P, index);
        if (pe == hipSuccess) pe = split_not_a(P);
        if (pe != hipSuccess) {
            *err = pe;          // the slot stays mapped (unprobed = class unknown); the caller reports the error
            return false;
        }
    }
    return true;
}

bool init(Placement &P, int dev) {
    if (P.init_tried) return P.vmm;
    P.init_tried = true;
    if (placement_disabled()) return false;
    int vmm = 0;
    if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev) != hipSuccess || !vmm) {
        (void)hipGetLastError();
        return false;
    }
    memset(&P.prop, 0, sizeof(P.prop));
    P.prop.type = hipMemAllocationTypePinned;
    P.prop.location.type = hipMemLocationTypeDevice;
    P.prop.location.id = dev;
    hipMemAccessDesc own;
    memset(&own, 0, sizeof(own));
    own.location = P.prop.location;
    own.flags = hipMemAccessFlagsProtReadWrite;
    P.access.assign(1, own);
    // (peers are added when a cross-device copy first needs them: placement_grant_peer.  Mapping every slot for every device of the
    // node up front would put eight sets of page tables behind each GiB of every rank of a one-process-per-GPU run, for nothing.)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 4 * kSlot) return false;
    P.total_bytes = total_b;
    char *raw = nullptr;
    bool ok = hipMemAddressReserve(reinterpret_cast<void **>(&raw), (kReserveSlots + 1) * kSlot, kSlot, nullptr, 0) == hipSuccess;
    // (the runtime does not honour the alignment asked for: the slots start at the first GiB boundary inside the range)
    if (ok) P.base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + kSlot - 1) / kSlot * kSlot);
    if (ok && !getenv("STATMC_PLACEMENT_NO_WINDOWS")) {   // (the switch: experiments -- the first version's runs of slots)
        char *raw2 = nullptr;
        if (hipMemAddressReserve(reinterpret_cast<void **>(&raw2), (kReserveSlots + 1) * kSlot, kSlot, nullptr, 0) == hipSuccess)
            P.win_base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw2) + kSlot - 1) / kSlot * kSlot);
        else
            (void)hipGetLastError();
    }
    ok = ok && hipMalloc(&P.sink, 64) == hipSuccess && hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreate(&P.e0) == hipSuccess && hipEventCreate(&P.e1) == hipSuccess;
    hipError_t err = hipSuccess;
    ok = ok && back_next_slot(P, &err);                  // slot 0: the probe's write target, never handed out
    if (ok) P.slots[0].role = kPrivate;
    ok = ok && hipMemsetAsync(P.base, 0, kProbeWindow, P.stream) == hipSuccess && hipStreamSynchronize(P.stream) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        return false;                                    // what was reserved stays reserved: address space, at most one GiB
    }
    P.vmm = true;
    return true;
}

// Nothing is classified before BOTH levels of the probe have been seen: the classes come in runs of 4 .. 64 slots, so the
// first dozen slots may well all be of one class -- slot 0's or not -- and one level alone does not say which.  Slots are
// backed until there is a tight cluster of fast probes (fast_level) and the other level has been seen beside it (calibrate); none
// after kCalibrationCap: no classes to tell apart on this device.
void add_free(std::map<size_t, size_t> &fl, size_t off, size_t len);

hipError_t probe_self(Placement &P);
// (diagnostic, STATMC_PLACEMENT_DEBUG: do the shaders see the traded memory at the old addresses?)
__global__ void poke_pair_kernel(unsigned *a, unsigned va, unsigned *b, unsigned vb) { a[0] = va; b[0] = vb; }
__global__ void peek_pair_kernel(const unsigned *a, const unsigned *b, unsigned *out) { out[0] = a[0]; out[1] = b[0]; }
// Slot k's memory becomes the reference: the two physical allocations trade addresses, every probe is taken again (calibrate)
hipError_t rebase(Placement &P, size_t k) {
    if (hipError_t e = hipStreamSynchronize(P.stream); e != hipSuccess) return e;
    char *a0 = P.base, *ak = P.base + k * kSlot;
    // (a word poked into both slots before the trade and read back through both addresses after it: the trade is checked, not assumed)
    hipLaunchKernelGGL(poke_pair_kernel, dim3(1), dim3(1), 0, P.stream, reinterpret_cast<unsigned *>(a0 + kSlot - 64), 0xA0A0u,
                       reinterpret_cast<unsigned *>(ak + kSlot - 64), 0xB0B0u);
    if (hipError_t e = hipStreamSynchronize(P.stream); e != hipSuccess) return e;
    if (hipError_t e = hipMemUnmap(a0, kSlot); e != hipSuccess) return e;
    if (hipError_t e = hipMemUnmap(ak, kSlot); e != hipSuccess) return e;
    if (hipError_t e = flush_translations(); e != hipSuccess) return e;
    std::swap(P.slots[0].handle, P.slots[k].handle);
    for (char *at : {a0, ak}) {
        const size_t i = at == a0 ? 0 : k;
        if (hipError_t e = hipMemMap(at, kSlot, 0, P.slots[i].handle, 0); e != hipSuccess) return e;
        if (hipError_t e = set_access(P, at); e != hipSuccess) return e;
    }
    {
        unsigned host[2] = {0u, 0u};
        hipLaunchKernelGGL(peek_pair_kernel, dim3(1), dim3(1), 0, P.stream, reinterpret_cast<const unsigned *>(a0 + kSlot - 64),
                           reinterpret_cast<const unsigned *>(ak + kSlot - 64), reinterpret_cast<unsigned *>(P.sink));
        if (hipError_t e = hipStreamSynchronize(P.stream); e != hipSuccess) return e;
        if (hipError_t e = hipMemcpy(host, P.sink, 8, hipMemcpyDeviceToHost); e != hipSuccess) return e;
        if (getenv("STATMC_PLACEMENT_DEBUG"))
            fprintf(stderr, "statmc placement: after the trade slot 0 reads %#x (0xb0b0 = slot %zu's memory), slot %zu reads %#x\n", host[0], k, k, host[1]);
        if (host[0] != 0xB0B0u || host[1] != 0xA0A0u) return hipErrorUnknown;   // the shaders still see the old memory: no trade (the caller retires the allocator)
    }
    if (hipError_t e = hipMemsetAsync(P.base, 0, kProbeWindow, P.stream); e != hipSuccess) return e;
    if (P.target2 >= 0) {                       // (not chosen yet at this point; for completeness)
        P.slots[P.target2].role = -1;
        P.target2 = -1;
    }
    P.fastest_ms = P.slowest_ms = P.level = 0.f;
    for (size_t i = 1; i < P.slots.size(); i++) P.slots[i].probe_ms[0] = P.slots[i].probe_ms[1] = 0.f;
    if (hipError_t e = probe_self(P); e != hipSuccess) return e;
    for (size_t i = 1; i < P.slots.size(); i++)
        if (P.slots[i].role != kReleased)
            if (hipError_t e = probe_slot(P, i); e != hipSuccess) return e;
    P.rebased = true;
    if (getenv("STATMC_PLACEMENT_DEBUG")) fprintf(stderr, "statmc placement: slot %zu's memory is the reference now\n", k);
    return hipSuccess;
}

// Slot 0 probed against itself: the level of "same class as slot 0" without having met a second slot of that class.  Round 6: two
// processes in a row on one box declared "no contrast" after 96 probes -- slot 0 sat in a class of which the card's first 96 GiB held
// one more slot or none (probe_ms 0.178 .. 0.194 and 0.176 .. 0.181), and the rule "both levels on two slots each" never fired;
// everything came unclassified and the accumulation ran at 0.73 of the HBM peak (profiles/r06_bench_j1.json, _j2).  A reference in a
// RARE class is the best case, not a failure: with this probe a tight cluster of three slots 8 % faster than it is proof enough of
// contrast (calibrate).
hipError_t probe_self(Placement &P) {
    float best = 1e30f;
    hipError_t err = hipSuccess;
    for (int rep = 0; rep < 6 && err == hipSuccess; rep++) {
        (void)hipEventRecord(P.e0, P.stream);
        hipLaunchKernelGGL(slot_probe_kernel, dim3(2048), dim3(256), 0, P.stream, reinterpret_cast<const vfloat4 *>(P.base),
                           reinterpret_cast<vuint4 *>(P.base), kSlot / 16, kProbeWindow / 16, P.sink);
        (void)hipEventRecord(P.e1, P.stream);
        err = hipEventSynchronize(P.e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, P.e0, P.e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (err == hipSuccess) P.self_ms = best;
    if (getenv("STATMC_PLACEMENT_DEBUG")) fprintf(stderr, "statmc placement: slot 0 against itself: %.4f ms\n", best);
    return err;
}

hipError_t calibrate(Placement &P) {
    hipError_t err = hipSuccess;
    if (!P.calibrated && P.self_ms == 0.f) err = probe_self(P);
    if (err != hipSuccess) return err;
  again:
    while (!P.calibrated) {
        // a tight fast cluster (fast_level: three probes within 2 %) AND the other level seen -- on two slots (one noisy probe
        // must not invent a class), or as slot 0 against itself (probe_self: "the same class" by construction; 1.10 - 1.13 x the fast
        // level where a slot that straddles two classes reads 1.045 x and the slow cluster 1.075 - 1.10 x)
        int n_slow = 0;
        for (size_t i = 1; i < P.slots.size(); i++) n_slow += P.level > 0.f && P.slots[i].probe_ms[0] > kContrast * P.level ? 1 : 0;
        if (P.level > 0.f && (n_slow >= 2 || P.self_ms > kSelfContrast * P.level)) {
            P.calibrated = true;
        } else if (P.slots.size() >= (size_t)kCalibrationCap || !back_next_slot(P, &err, 8ull << 30)) {
            P.calibrated = true;
            P.no_contrast = true;
        }
    }
    // Round 6: the reference should not sit in the class the card has most of HERE.  Slot 0 is whatever the driver handed out first,
    // and everything placed afterwards is classified against it: the moments live in its class, the arenas must avoid it.  On a card
    // whose first dozens of GiB are one long run of slot 0's class a search on a byte budget never finds the arenas' 22 slots apart
    // from it and deals them as they come -- INTO the moments' class: 0.72 of the HBM peak where torch's allocator gets 0.79
    // (profiles/r06f_bench.json).  When the calibration has seen a long run of slot 0's class (eight slots or more, three times what it saw apart from it),
    // the first slot apart from it trades places with slot 0 -- the two physical allocations are mapped at each other's addresses --
    // and every slot is probed again against the new reference: the long run is now the arenas' class.  Once, before any block
    // exists.
    if (err == hipSuccess && !P.no_contrast && !P.rebase_tried && P.live.empty() && P.windows.empty() && !P.slot0_dealt && !getenv("STATMC_PLACEMENT_NO_REBASE")) {
        P.rebase_tried = true;
        size_t n_same = 0, n_apart = 0, first_apart = 0;
        for (size_t i = 1; i < P.slots.size(); i++) {
            const float t = P.slots[i].probe_ms[0];
            if (t <= 0.f || P.slots[i].role == kReleased) continue;
            if (t > kSameAbove * P.level) n_same++;
            else if (t < kApartBelow * P.level) { n_apart++; if (!first_apart) first_apart = i; }
        }
        // (STATMC_PLACEMENT_FORCE_REBASE=1: tests -- the trade on a card that does not call for it)
        if (first_apart && ((n_same >= 8 && n_same >= 3 * n_apart) || getenv("STATMC_PLACEMENT_FORCE_REBASE"))) {   // a clear long run, not a coin toss on four probes
            err = rebase(P, first_apart);
            // a trade that failed half-way may have left slot 0 without memory: no block exists yet, so the allocator simply retires
            // on this device -- every later statmc_malloc_placed is a hipMalloc (init() answers with P.vmm)
            if (err != hipSuccess) P.vmm = false;
            // Against the new reference the slots met so far may show no fast cluster at all (a card whose long run probed 1.04 - 1.07 x
            // against the traded slot -- neither level: classified from THAT the arenas went as they came, 0.73 where torch's allocator
            // got 0.76, gpurun_out r06_bench_w3): the calibration's own criterion decides again, and backs slots until it holds.
            if (err == hipSuccess) {
                P.calibrated = false;
                goto again;
            }
        }
    }
    if (err == hipSuccess) err = split_not_a(P);
    // The state's first home is slot 0 itself, behind the probe's 64-MiB window: every other slot is classified by what a stream of
    // it suffers beside writes into THAT memory, so "apart from slot 0" is "apart from the moments" without going through a second
    // slot's own classification (which is fuzzy where the card interleaves its classes finer than a GiB: runs of probe ratios
    // between the two levels, profiles/r05_acc_bisect5.log).  960 MiB: the moments and work images of a 1080p film twice over.
    if (err == hipSuccess && !P.no_contrast && !P.slot0_dealt) {
        add_free(P.free_blocks[STATMC_MEM_STATE], kProbeWindow, kSlot - kProbeWindow);
        P.slot0_dealt = true;
    }
    return err;
}

// class masks of the searches a role makes, strongest first
constexpr unsigned bit(int c) { return 1u << c; }
constexpr unsigned kAnyClass = ~0u;
constexpr unsigned kWellInsideA = 1u << 16;   // with bit(kClassA): only slots whose probe lies well inside the slow cluster (state_above)
bool suits(const Placement &P, const Slot &s, unsigned mask) {
    if (s.role != -1) return false;      // dealt, private or released
    if (P.no_contrast || mask == kAnyClass) return true;
    const int c = classify(P, s);
    if (c == kClassA && (mask & kWellInsideA) && !(s.probe_ms[0] > state_above() * P.level)) return false;
    return c >= 0 && (mask & bit(c));
}

void add_free(std::map<size_t, size_t> &fl, size_t off, size_t len) {
    auto next = fl.lower_bound(off);
    if (next != fl.end() && off + len == next->first) {
        len += next->second;
        next = fl.erase(next);
    }
    if (next != fl.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == off) {
            off = prev->first;
            len += prev->second;
            fl.erase(prev);
        }
    }
    fl[off] = len;
}

// Whole slots that lie inside free space of a role go back to the pool of undealt slots (same memory, same class): what a large
// block leaves behind can serve the other role's blocks or a window, not only its own role's next small block.
void undeal_free_slots(Placement &P, int role) {
    auto &fl = P.free_blocks[role];
    for (auto it = fl.begin(); it != fl.end();) {
        const size_t off = it->first, len = it->second;
        const size_t s0 = (off + kSlot - 1) / kSlot, s1 = (off + len) / kSlot;   // whole slots [s0, s1) of this free block
        if (s1 <= s0) {
            ++it;
            continue;
        }
        fl.erase(it);
        if (off < s0 * kSlot) fl[off] = s0 * kSlot - off;
        if (s1 * kSlot < off + len) fl[s1 * kSlot] = off + len - s1 * kSlot;
        for (size_t i = s0; i < s1; i++) {
            P.slots[i].role = -1;
            P.slots[i].as_it_came = false;
        }
        it = fl.lower_bound(s1 * kSlot);
    }
}

// deals `count` consecutive undealt slots, starting at slot `first`, to the role's free list
void deal(Placement &P, int role, size_t first, size_t count, bool wanted_class) {
    for (size_t i = first; i < first + count; i++) {
        P.slots[i].role = role;
        P.slots[i].as_it_came = !wanted_class && !P.no_contrast;
    }
    add_free(P.free_blocks[role], first * kSlot, count * kSlot);
}

// Finds `want_slots` undealt slots of the classes in `mask` in a row and deals them to the role; backs new slots at the end of
// the range, up to `cap_slots` in all, until there is such a run.  STATMC_ERR_UNSUPPORTED: none (the caller searches weaker).
int find_run(Placement &P, int role, size_t want_slots, unsigned mask, size_t cap_slots, bool wanted_class, size_t leave_free) {
    for (;;) {
        size_t run = 0;
        for (size_t i = 1; i < P.slots.size(); i++) {
            run = suits(P, P.slots[i], mask) ? run + 1 : 0;
            if (run == want_slots) {
                deal(P, role, i + 1 - want_slots, want_slots, wanted_class);
                return STATMC_OK;
            }
        }
        if (backed_count(P) >= cap_slots) return STATMC_ERR_UNSUPPORTED;
        hipError_t err = hipSuccess;
        if (!back_next_slot(P, &err, leave_free)) {
            if (err != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "placement: %s", hipGetErrorString(err));
            return STATMC_ERR_UNSUPPORTED;
        }
    }
}

// Gathers `want` undealt slots of the classes in `mask` (anywhere in the range; new slots are backed at its end, up to `cap_slots`
// in all, until there are enough) and maps them side by side in the window range.  STATMC_ERR_UNSUPPORTED: not enough of them.
int window_alloc(Placement &P, int role, size_t bytes, size_t want, unsigned mask, size_t cap_slots, bool wanted_class, size_t leave_free, void **out) {
    std::vector<size_t> chosen;
    for (;;) {
        chosen.clear();
        for (size_t i = 1; i < P.slots.size() && chosen.size() < want; i++)
            if (suits(P, P.slots[i], mask)) chosen.push_back(i);
        if (chosen.size() == want) break;
        if (backed_count(P) >= cap_slots) return STATMC_ERR_UNSUPPORTED;
        hipError_t err = hipSuccess;
        if (!back_next_slot(P, &err, leave_free)) {
            if (err != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "placement: %s", hipGetErrorString(err));
            return STATMC_ERR_UNSUPPORTED;
        }
    }
    // `want` free window slots in a row (first fit; the window range is as long as the slot range)
    size_t first = 0, run = 0;
    bool found = false;
    for (size_t w = 0; w < P.win_slot.size() && !found; w++) {
        run = P.win_slot[w] < 0 ? run + 1 : 0;
        if (run == want) { first = w + 1 - want; found = true; }
    }
    if (!found) {
        size_t tail = 0;                                   // free window slots at the end of what is in use
        while (tail < P.win_slot.size() && P.win_slot[P.win_slot.size() - 1 - tail] < 0) tail++;
        first = P.win_slot.size() - tail;
        if (first + want > kReserveSlots) return STATMC_ERR_UNSUPPORTED;
        P.win_slot.resize(first + want, -1);
    }
    for (size_t k = 0; k < want; k++) {
        char *at = P.win_base + (first + k) * kSlot;
        hipError_t e = hipMemMap(at, kSlot, 0, P.slots[chosen[k]].handle, 0);
        if (e == hipSuccess) {
            e = set_access(P, at);
            if (e != hipSuccess) (void)hipMemUnmap(at, kSlot);
        }
        if (e != hipSuccess) {
            for (size_t j = 0; j < k; j++) (void)hipMemUnmap(P.win_base + (first + j) * kSlot, kSlot);
            if (k > 0) (void)flush_translations();
            (void)hipGetLastError();
            return statmc::abi_fail(STATMC_ERR_HIP, "placement: mapping a slot into a window: %s", hipGetErrorString(e));
        }
    }
    for (size_t k = 0; k < want; k++) {
        Slot &s = P.slots[chosen[k]];
        s.role = role;
        s.as_it_came = !wanted_class && !P.no_contrast;
        s.window = (int)(first + k);
        P.win_slot[first + k] = (int)chosen[k];
    }
    P.windows[first * kSlot] = Window{first, want, bytes, role, wanted_class || P.no_contrast};
    *out = P.win_base + first * kSlot;
    return STATMC_OK;
}

// the window block that holds `ptr`, or end()
std::map<size_t, Window>::iterator window_of(Placement &P, const void *ptr) {
    if (!P.win_base || (const char *)ptr < P.win_base || (const char *)ptr >= P.win_base + P.win_slot.size() * kSlot) return P.windows.end();
    const size_t off = (size_t)((const char *)ptr - P.win_base);
    auto it = P.windows.upper_bound(off);
    if (it == P.windows.begin()) return P.windows.end();
    --it;
    return off < (it->second.first + it->second.n) * kSlot ? it : P.windows.end();
}

// The arenas' class: whichever of the two classes apart from the state's first has `want` undealt slots among those backed (slots
// are backed until one has, or `cap_slots` are); the arenas of a device then all go there (one class for all of them measured 0.842
// of the HBM peak, spread over both 0.818).  Which of the two the card has more of near the start of the range differs from box to
// box: always taking the second probe target's class (the first version) backed 176 GiB on a card whose first hundred slots were of
// the other one.
int pick_stream_class(Placement &P, size_t want, size_t cap_slots, size_t leave_free) {
    for (;;) {
        size_t n[2] = {0, 0};
        for (size_t i = 1; i < P.slots.size(); i++) {
            if (P.slots[i].role != -1) continue;
            const int c = classify(P, P.slots[i]);
            if (c == kClassB || c == kClassC) n[c - kClassB]++;
        }
        if (n[0] >= want || n[1] >= want) return n[0] >= want && n[0] >= n[1] ? kClassB : n[1] >= want ? kClassC : kClassB;
        hipError_t err = hipSuccess;
        if (backed_count(P) >= cap_slots || !back_next_slot(P, &err, leave_free)) return n[1] > n[0] ? kClassC : kClassB;
    }
}

int take_block(Placement &P, int role, size_t need, void **out) {
    auto &fl = P.free_blocks[role];
    for (auto it = fl.begin(); it != fl.end(); ++it) {
        if (it->second < need) continue;
        const size_t off = it->first, len = it->second;
        fl.erase(it);
        if (len > need) fl[off + need] = len - need;
        P.live[off] = std::make_pair(need, role);
        *out = P.base + off;
        return STATMC_OK;
    }
    return STATMC_ERR_UNSUPPORTED;
}

// may_back = false (the library's own workspaces, asked for from inside a filter call while the caller's kernels are in flight):
// only what is backed already -- no new slot, hence no probe beside somebody's kernel and no search under the lock
int placed_alloc(Placement &P, int role, size_t bytes, void **out, bool may_back = true) {
    const size_t need = (bytes + kBlock - 1) / kBlock * kBlock;
    if (take_block(P, role, need, out) == STATMC_OK) return STATMC_OK;
    if (may_back) {
        if (hipError_t e = calibrate(P); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "placement probe: %s", hipGetErrorString(e));
        if (take_block(P, role, need, out) == STATMC_OK) return STATMC_OK;   // (the calibration deals slot 0's tail to the state role)
    }
    // whole slots are dealt; the free list joins them with what the role already holds next to them
    const size_t want_slots = (need + kSlot - 1) / kSlot;
    // how many slots the search for the right class may have backed: the byte budget (3 x what has been asked for, or
    // STATMC_PLACEMENT_MAX_GIB), and never more than 60 % of the card for the first choice, 75 % at all (the rest of the process --
    // the caller's other allocations, the runtime's -- needs room too; beyond that: what is backed already, any class)
    // (what the caller has announced for this role counts as asked for: the class is then searched, and chosen, for ALL the arenas at
    // once -- arena by arena the first one's budget of 3 x 6 + 6 GiB settles for the class that has six slots at hand, and the later ones
    // for whatever is left: 16 slots of one class + 6 of the other, 0.77 of the HBM peak where one class for all gets 0.805)
    const size_t outlook = std::max(need, P.expect[role]);
    const size_t budget = may_back ? budget_slots(P, outlook) : 0;
    const size_t soft_cap = std::min<size_t>({kReserveSlots, (size_t)(0.60 * (double)P.total_bytes / (double)kSlot), budget});
    const size_t hard_cap = std::min<size_t>({kReserveSlots, (size_t)(0.75 * (double)P.total_bytes / (double)kSlot), budget});
    const size_t last_cap = may_back ? kReserveSlots : 0;
    struct Search { unsigned mask; size_t cap; bool wanted; };
    // which class serves which role: the state takes slot 0's class (A), the arenas ONE of the other two (pick_stream_class), then the third.
    // STATMC_PLACEMENT_ROLES=BCA etc. (experiment: are the classes interchangeable?) permutes that.
    static const int cls[3] = {[] { const char *e = getenv("STATMC_PLACEMENT_ROLES"); return e && strlen(e) == 3 ? e[0] - 'A' : 0; }(),
                               [] { const char *e = getenv("STATMC_PLACEMENT_ROLES"); return e && strlen(e) == 3 ? e[1] - 'A' : 1; }(),
                               [] { const char *e = getenv("STATMC_PLACEMENT_ROLES"); return e && strlen(e) == 3 ? e[2] - 'A' : 2; }()};
    const bool perm_ok = cls[0] >= 0 && cls[0] < 3 && cls[1] >= 0 && cls[1] < 3 && cls[2] >= 0 && cls[2] < 3 && cls[0] != cls[1] && cls[1] != cls[2] && cls[0] != cls[2];
    int cS = perm_ok ? cls[0] : kClassA, cT = perm_ok ? cls[1] : kClassB, cU = perm_ok ? cls[2] : kClassC;
    if (role == STATMC_MEM_STREAM && !P.no_contrast && !getenv("STATMC_PLACEMENT_ROLES")) {
        int c = P.stream_class;
        if (c < 0) {
            c = pick_stream_class(P, (outlook + kSlot - 1) / kSlot, soft_cap, 8ull << 30);
            if (want_slots >= 3) P.stream_class = c;       // an arena decides for the device; a small block takes what is at hand
        }
        if (c == kClassC) std::swap(cT, cU);
    }
    const unsigned not_state = bit(cT) | bit(cU) | (cS == kClassA ? bit(kNotA) : 0u);
    const Search state_order[] = {{bit(cS) | (cS == kClassA ? kWellInsideA : 0u), soft_cap, true}, {bit(cS), 0, true}, {bit(cS) | bit(kMixed), 0, false},
                                  {kAnyClass, hard_cap, false}, {kAnyClass, last_cap, false}};   // (last resort: until the card is full)
    const Search stream_order[] = {{bit(cT), soft_cap, true},                            // one class for all arenas
                                   {bit(cU), 0, true},                                   // ... or the other one
                                   {not_state, hard_cap, true},                          // both (still apart from the state)
                                   // ... then slots that straddle classes -- at worst half of such a slot is the moments' class -- before
                                   // anything of the moments' class itself (a card whose slots 34 .. 72 all read 1.03 - 1.06 x: the arenas
                                   // went "as they came" into slots 1 - 6, slot 0's own run: 0.718 where torch's allocator got 0.77,
                                   // profiles/r06_place_t2.json)
                                   {not_state | bit(kMixed), 0, false},
                                   {kAnyClass, hard_cap, false}, {kAnyClass, last_cap, false}};
    int rc = STATMC_ERR_UNSUPPORTED;
    const Search *order = role == STATMC_MEM_STATE ? state_order : stream_order;
    const int n_order = role == STATMC_MEM_STATE ? 5 : 6;
    // searching for a CLASS never takes the card's last 8 GiB (other allocators of the process need room); only the last resort --
    // any class, the request would fail otherwise -- goes down to half a GiB
    if (want_slots >= 3 && P.win_base) {                   // (a window takes whole slots: below 2 GiB a run of two is the better deal)
        // a run of first-choice slots among those already backed serves the block as it is (and its tail stays in the role's free list) ...
        if (find_run(P, role, want_slots, order[0].mask, 0, true, 8ull << 30) == STATMC_OK && take_block(P, role, need, out) == STATMC_OK) return STATMC_OK;
        // ... otherwise a window: the slots need not lie side by side
        for (int k = 0; k < n_order && rc == STATMC_ERR_UNSUPPORTED; k++)
            rc = window_alloc(P, role, bytes, want_slots, order[k].mask, order[k].cap, order[k].wanted, k + 1 < n_order ? (8ull << 30) : (512ull << 20), out);
        if (rc == STATMC_OK || rc == STATMC_ERR_HIP) return rc;
        if (!may_back) return STATMC_ERR_UNSUPPORTED;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        return statmc::abi_fail(STATMC_ERR_HIP, "statmc_malloc_placed: out of device memory (%zu slots backed, %d probes %.3f .. %.3f ms, %.1f GiB free, last: %s)",
                                P.slots.size(), P.n_probes, P.fastest_ms, P.slowest_ms, free_b / 1073741824.0, P.last_note);
    }
    for (int k = 0; k < n_order && rc == STATMC_ERR_UNSUPPORTED; k++)
        rc = find_run(P, role, want_slots, order[k].mask, order[k].cap, order[k].wanted, k + 1 < n_order ? (8ull << 30) : (512ull << 20));
    if (rc == STATMC_ERR_HIP) return rc;
    if (rc != STATMC_OK && !may_back) return STATMC_ERR_UNSUPPORTED;
    if (rc != STATMC_OK) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        return statmc::abi_fail(STATMC_ERR_HIP, "statmc_malloc_placed: out of device memory (%zu slots backed, %d probes %.3f .. %.3f ms, %.1f GiB free, last: %s)",
                                P.slots.size(), P.n_probes, P.fastest_ms, P.slowest_ms, free_b / 1073741824.0, P.last_note);
    }
    if (take_block(P, role, need, out) == STATMC_OK) return STATMC_OK;
    return statmc::abi_fail(STATMC_ERR_HIP, "statmc_malloc_placed: internal error (no block after dealing %zu slots)", want_slots);
}

}  // namespace

namespace statmc {

// Blocks of `owner`'s placed allocator become operands of copies that device `peer` executes (statmc_copy_rect / statmc_halo_exchange
// between two devices of one process: enable_peer in statmc_abi.hip calls this once per ordered pair).  hipDeviceEnablePeerAccess does
// not cover memory made by hipMemCreate / hipMemMap: every backed slot and every window mapping is granted to `peer` here, and slots
// backed later are mapped for the peers granted so far (set_access).  No allocator on `owner`, or no VMM: nothing to do.
hipError_t placement_grant_peer(int owner, int peer) {
    std::lock_guard<std::mutex> lk(g_place_mu);
    auto it = g_place.find(owner);
    if (it == g_place.end() || !it->second.vmm || owner == peer) return hipSuccess;
    Placement &P = it->second;
    for (const hipMemAccessDesc &d : P.access)
        if (d.location.id == peer) return hipSuccess;
    hipMemAccessDesc desc = P.access[0];
    desc.location.id = peer;
    for (size_t i = 0; i < P.slots.size(); i++) {
        if (P.slots[i].role == kReleased) continue;
        if (hipError_t e = hipMemSetAccess(P.base + i * kSlot, kSlot, &desc, 1); e != hipSuccess) return e;
    }
    for (size_t w = 0; w < P.win_slot.size(); w++) {
        if (P.win_slot[w] < 0) continue;
        if (hipError_t e = hipMemSetAccess(P.win_base + w * kSlot, kSlot, &desc, 1); e != hipSuccess) return e;
    }
    P.access.push_back(desc);
    P.peers_granted = true;
    return hipSuccess;
}

// The library's own workspaces (patch sums of the window filter, packed twins of pitched images) are written and read by
// every launch: on a device whose caller uses placed memory they live with the STATE role, so that what the filter leaves
// dirty in the caches is not written back into the class the next accumulation streams its samples from.  Plain hipMalloc
// where the placed allocator is not in use on the current device.
hipError_t workspace_alloc(void **p, size_t bytes) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_place_mu);
        auto it = g_place.find(dev);
        if (it != g_place.end() && it->second.vmm && it->second.calibrated && !it->second.no_contrast &&
            placed_alloc(it->second, STATMC_MEM_STATE, bytes ? bytes : 1, p, /*may_back=*/false) == STATMC_OK)
            return hipSuccess;
    }
    return hipMalloc(p, bytes);
}
hipError_t workspace_free(void *p) {
    if (p && placement_free(p) != 0) return hipSuccess;
    return hipFree(p);
}

// -1: not a block of the placed allocator's (or the allocator is not telling classes apart on that device); else the role it was
// dealt for.  What statmc_accumulate asks about its buffers: samples in STREAM blocks and moments in STATE blocks are known to
// lie in different interference classes, and the film-major launch shape is chosen with that in mind (launch_accumulate).
int placement_role_of(const void *ptr) {
    std::lock_guard<std::mutex> lk(g_place_mu);
    for (auto &kv : g_place) {
        Placement &P = kv.second;
        if (!P.base || !P.vmm || !P.calibrated || P.no_contrast) continue;
        if (auto w = window_of(P, ptr); w != P.windows.end()) return w->second.wanted ? w->second.role : -1;
        if ((const char *)ptr < P.base || (const char *)ptr >= P.base + P.slots.size() * kSlot) continue;
        const size_t off = (size_t)((const char *)ptr - P.base);
        // every slot the live block around `ptr` covers must have been dealt to its role with the wanted class (free ranges of
        // neighbouring slots coalesce whatever their class: a block may span a wanted slot and one taken as it came -- ADVICE r5)
        size_t b0 = off, b1 = off + 1;
        int role = -1;
        if (auto it = P.live.upper_bound(off); it != P.live.begin()) {
            --it;
            if (off < it->first + it->second.first) { b0 = it->first; b1 = it->first + it->second.first; role = it->second.second; }
        }
        if (role < 0) return -1;                  // not inside a live block
        for (size_t i = b0 / kSlot; i <= (b1 - 1) / kSlot; i++) {
            if (i == 0) {                         // slot 0: the allocator's own, and -- behind the probe window -- the state's first home
                if (!(P.slot0_dealt && role == STATMC_MEM_STATE && b0 >= kProbeWindow)) return -1;
                continue;
            }
            const Slot &s = P.slots[i];
            if (s.role != role || s.as_it_came) return -1;
        }
        return role;
    }
    return -1;
}

// statmc_free's question: is this pointer one of the placed allocator's?  1: it was a live block and is free now; 0: not this
// allocator's; -1: inside its ranges but not the start of a live block (an interior pointer, a block freed twice)
int placement_free(void *ptr) {
    std::lock_guard<std::mutex> lk(g_place_mu);
    for (auto &kv : g_place) {
        Placement &P = kv.second;
        if (auto w = window_of(P, ptr); w != P.windows.end()) {
            if (P.win_base + w->first != (char *)ptr) return -1;   // inside a window, not its start: nothing to do (and not hipFree's either)
            int cur = -1;
            if (hipGetDevice(&cur) == hipSuccess) {
                if (cur != kv.first) (void)hipSetDevice(kv.first);
                (void)hipDeviceSynchronize();
            }
            for (size_t k = 0; k < w->second.n; k++) {                // the slots keep their memory and their class: undealt again
                const size_t ws = w->second.first + k;
                (void)hipMemUnmap(P.win_base + ws * kSlot, kSlot);
                Slot &s = P.slots[(size_t)P.win_slot[ws]];
                s.role = -1;
                s.as_it_came = false;
                s.window = -1;
                P.win_slot[ws] = -1;
            }
            (void)flush_translations();        // before these slots of the window range serve another block (on the block's own device)
            if (cur >= 0 && cur != kv.first) (void)hipSetDevice(cur);
            P.windows.erase(w);
            return 1;
        }
        if (!P.base || (char *)ptr < P.base || (char *)ptr >= P.base + P.slots.size() * kSlot) continue;
        auto it = P.live.find((size_t)((char *)ptr - P.base));
        if (it == P.live.end()) return -1;            // inside the range, not a live block: nothing to do (and not hipFree's either)
        // like hipFree: work that may still use the block has finished before its memory can be handed out again
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess) {
            if (cur != kv.first) (void)hipSetDevice(kv.first);
            (void)hipDeviceSynchronize();
            if (cur != kv.first) (void)hipSetDevice(cur);
        }
        const int role = it->second.second;
        add_free(P.free_blocks[role], it->first, it->second.first);
        P.live.erase(it);
        undeal_free_slots(P, role);
        return 1;
    }
    return 0;
}

}  // namespace statmc

extern "C" {

int statmc_malloc_placed(void **dev_ptr, size_t bytes, int role) {
    if (!dev_ptr) return statmc::abi_fail(STATMC_ERR_INVALID, "null dev_ptr");
    if (role != STATMC_MEM_STATE && role != STATMC_MEM_STREAM) return statmc::abi_fail(STATMC_ERR_INVALID, "role must be STATMC_MEM_STATE or STATMC_MEM_STREAM");
    if (bytes == 0) bytes = 1;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_place_mu);
    Placement &P = g_place[dev];
    if (!init(P, dev)) {                                   // no virtual-memory management, too little memory, or switched off
        void *p = nullptr;
        if (hipError_t e = hipMalloc(&p, bytes); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
        *dev_ptr = p;
        return STATMC_OK;
    }
    const int rc = placed_alloc(P, role, bytes, dev_ptr);
    if (rc == STATMC_OK) P.expect[role] -= std::min(P.expect[role], bytes);
    if (rc != STATMC_OK && !P.vmm) {                       // the allocator retired during this very call (a trade that did not take): plain memory
        void *p = nullptr;
        if (hipError_t e = hipMalloc(&p, bytes); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
        *dev_ptr = p;
        return STATMC_OK;
    }
    return rc;
}

int statmc_placement_expect(int role, size_t bytes) {
    if (role != STATMC_MEM_STATE && role != STATMC_MEM_STREAM) return statmc::abi_fail(STATMC_ERR_INVALID, "role must be STATMC_MEM_STATE or STATMC_MEM_STREAM");
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_place_mu);
    g_place[dev].expect[role] = bytes;
    return STATMC_OK;
}

int statmc_placement_info(statmc_placement_info_t *out) {
    if (!out) return statmc::abi_fail(STATMC_ERR_INVALID, "null out");
    memset(out, 0, sizeof(*out));
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_place_mu);
    auto it = g_place.find(dev);
    if (it == g_place.end()) return STATMC_OK;
    const Placement &P = it->second;
    out->active = P.vmm && P.calibrated && !P.no_contrast ? 1 : 0;
    out->virtual_memory = P.vmm ? 1 : 0;
    out->slots = (int)backed_count(P);
    out->slots_released = (int)(P.slots.size() - backed_count(P));
    out->peer_devices = P.peers_granted ? (int)P.access.size() - 1 : 0;
    out->peak_slots = (int)P.peak_backed;
    out->rebased = P.rebased ? 1 : 0;
    out->probes = P.n_probes;
    out->fast_probe_ms = P.fastest_ms;
    out->slow_probe_ms = P.slowest_ms;
    for (size_t i = 1; i < P.slots.size(); i++) {
        const Slot &s = P.slots[i];
        if (s.role == kReleased) continue;          // a hole in the range: no memory, no class
        const int c = classify(P, s);
        (c == kClassA ? out->slots_a : c == kClassB ? out->slots_b : c == kClassC ? out->slots_c : out->slots_unclear)++;
        if (s.role == kPrivate) continue;
        if (s.role == -1) out->slots_idle++;
        else {
            out->slab_bytes[s.role] += kSlot;
            if (s.as_it_came) out->slots_as_they_came[s.role]++;
        }
    }
    if (P.slot0_dealt) out->slab_bytes[STATMC_MEM_STATE] += kSlot - kProbeWindow;
    for (const auto &kv : P.live) out->live_bytes[kv.second.second] += kv.second.first;
    for (const auto &kv : P.windows) out->live_bytes[kv.second.role] += kv.second.bytes;
    return STATMC_OK;
}

// Test hook: what statmc_accumulate learns about a buffer -- the role of the placed block `ptr` lies in (any address inside it),
// -1 when it is not one, was dealt without the wanted class, or the device tells no classes apart
int statmc_debug_placement_role(const void *ptr) { return statmc::placement_role_of(ptr); }
float statmc_debug_placement_fast_level(const float *probes_ms, int n, float self_ms) {
    return probes_ms && n > 0 ? fast_level_of(probes_ms, (size_t)n, self_ms) : 0.f;
}

// Test / experiment hook (include/statmc_debug.h): the allocator's probe on memory of the caller's -- streams `stream_bytes` at
// `stream_ptr` while every fourth step read-modify-writes 16 bytes inside [rmw_ptr, rmw_ptr + rmw_bytes) (their values change: + 1 in
// every fourth word).  Best of five in *ms.
int statmc_debug_interference_probe(const void *stream_ptr, size_t stream_bytes, void *rmw_ptr, size_t rmw_bytes, float *ms) {
    if (!stream_ptr || !rmw_ptr || !ms || stream_bytes < (1u << 20) || rmw_bytes < (1u << 20) || (stream_bytes | rmw_bytes) % 16)
        return statmc::abi_fail(STATMC_ERR_INVALID, "statmc_debug_interference_probe: two buffers of at least 1 MiB, sizes multiples of 16");
    float *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipMalloc(&sink, 64);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6 && err == hipSuccess; rep++) {
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(slot_probe_kernel, dim3(2048), dim3(256), 0, nullptr, static_cast<const vfloat4 *>(stream_ptr), static_cast<vuint4 *>(rmw_ptr),
                           stream_bytes / 16, rmw_bytes / 16, sink);
        (void)hipEventRecord(e1, nullptr);
        err = hipEventSynchronize(e1);
        float t = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && t < best) best = t;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (sink) (void)hipFree(sink);
    if (err != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "interference probe: %s", hipGetErrorString(err));
    *ms = best;
    return STATMC_OK;
}

// Gives the memory of every idle slot (backed, probed, dealt to no role) back to the driver; the slots stay as holes in the
// address range.  For a host that has made its allocations and wants the rest of the card for something else: the next
// statmc_malloc_placed that needs room fills the holes first (new memory, probed anew; the driver may well hand the same
// memory out again), then goes on at the end of the range.  Returns the number of slots released, or a negative error.
int statmc_placement_trim(void) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_place_mu);
    auto it = g_place.find(dev);
    if (it == g_place.end() || !it->second.vmm) return 0;
    Placement &P = it->second;
    (void)hipDeviceSynchronize();
    int n = 0;
    for (size_t i = 1; i < P.slots.size(); i++) {
        Slot &s = P.slots[i];
        if (s.role != -1) continue;
        if (hipError_t e = hipMemUnmap(P.base + i * kSlot, kSlot); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipMemUnmap: %s", hipGetErrorString(e));
        (void)hipMemRelease(s.handle);
        s.role = kReleased;
        n++;
    }
    if (n > 0) (void)flush_translations();     // before a hole can be filled again
    return n;
}

// One character per backed slot: '#' the allocator's own, a / b / c an idle slot of that class, A / B / C one dealt to a role
// (upper case S / T when it was dealt without the wanted class: S state, T stream), '?' unclear or unprobed.
int statmc_placement_map(char *out, int capacity) {
    if (!out || capacity < 1) return statmc::abi_fail(STATMC_ERR_INVALID, "null out");
    out[0] = 0;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_place_mu);
    auto it = g_place.find(dev);
    if (it == g_place.end()) return STATMC_OK;
    const Placement &P = it->second;
    int n = 0;
    for (size_t i = 0; i < P.slots.size() && n + 1 < capacity; i++) {
        const Slot &s = P.slots[i];
        const int c = classify(P, s);
        char ch = c == kClassA ? 'a' : c == kClassB ? 'b' : c == kClassC ? 'c' : '?';
        if (s.role == kPrivate) ch = '#';
        else if (s.role == kReleased) ch = '_';
        else if (s.role >= 0 && s.as_it_came) ch = s.role == STATMC_MEM_STATE ? 'S' : 'T';
        else if (s.role >= 0 && ch != '?') ch = (char)(ch - 'a' + 'A');
        out[n++] = ch;
    }
    out[n] = 0;
    return STATMC_OK;
}

}  // extern "C"
