// statmc_filter_sym.hip -- the pair-symmetric window filter (gfx950): filter<float3> under every filter spec (Welch degrees
// of freedom included, since round 4: the Welch builds), filter<float> two buffers per launch under the symmetric gate (any
// gate under Welch); compile-time radius 20, runtime radius 1 .. 19; six or eight feature planes (Welch: six).
//
// Replaces the window part of cv::cuda::stat_denoiser::filter<float3> / filter<float> (call sites
// src/statistics/estimator.cpp:465-487 and 437-459 of the reference; CUDA source not in the tree, arithmetic = this
// build's spec v2, oracle/statmc_oracle.c:oracle_filter_spec_run).
//
// Why.  The filter is a (2r+1)^2 = 1681-tap stencil bound by VALU issue (33 fp32 lane-operations per
// (tap, pixel) pair, against 72 B of HBM traffic per pixel).  Under spec v2 both the range weight and the
// membership gate of a pair have the SAME BITS for (p, q) and (q, p):
//     e_pq = ds |p - q|^2 + sum_g DR_g |G_p - G_q|^2          (squares of differences)
//     member_pq  <=>  max_c fma(d_c, d_c, -(D_p,c + D_q,c)) <= 0,   d_c = mc_p,c - mc_q,c
// so every unordered pair is evaluated ONCE -- by its upper pixel p, over the half window {dy > 0} -- and the
// weight is added to both sides: w * colour_q to p (registers) and w * colour_p to q (LDS accumulator rows).
// 29 operations for weight + gate and 4 + 4 for the two accumulations = 37 per pair = 18.5 per directed tap.
//
// Shape of the work.
//   * Tile = 128 x 8 pixels on a grid fixed in FILM coordinates (so that block-decomposed multi-GPU runs
//     form every sum in the same order as a whole-film run: bit-identical results).  A 512-thread workgroup
//     = 8 waves; wave (t, h), t = 0..3, h = 0..1, owns tile rows t (lanes 0-31) and t + 4 (lanes 32-63) and
//     share h of the window columns; a lane owns 4 adjacent pixels.  Both waves of a row keep the pixels' own
//     data in registers; their partial sums meet in the epilogue.  The shares (round 6): whole READ GROUPS --
//     half 0 the groups [0, G), half 1 the rest, G = 7 or 6 of the 11 by build (gsplit_of) -- so that no group's
//     operands are read by both waves and the older wave of every SIMD pair, which the issue arbitration
//     prefers, has the larger share (rounds 2 - 5: h = 0 dx in [-20, 0], h = 1 dx in [1, 20]).
//   * Step s = window row dy = s, s = 0..20.  At step s tile row t sweeps image row y0 + t + s: 8 live rows
//     + 1 being staged = a ring of 9 LDS slots.  A slot holds the 15 input planes of a row (168 columns =
//     tile + 2 x 20 halo) AND its 2 x 4 accumulator planes (Sigma w*colour, Sigma w; one copy per window
//     half, because the two waves of a row hit the same columns at the same time).  Everything a row needs
//     lives exactly as long as the row: 9 x 168 x 23 floats = 139 KB, + the spatial table and the LDS-DMA
//     landing area = 151 KB of the CU's 160 KB.
//   * dy = 0: the pairs inside a row are the taps dx in [1, 20], all with half 1 (the accumulator row is the wave's own
//     row); half 0 only adds the pixel's own tap.
//   * q-side scatter: per read group a lane reads the 4 x 2 accumulator values of its two taps, adds its four
//     pixels' w * colour_p with the same packed FMAs that serve the p side, and writes them back.  Plain
//     read-modify-write: within a step one wave owns an accumulator row copy, and the LDS executes a wave's
//     instructions in order, so the lane that continues a column sees its neighbour's sum.  (ds_add_f32
//     costs ~190 cycles per wave instruction on gfx950 -- tools/microbench/lds_scatter.hip -- 30x the RMW.)
//   * When a row leaves the ring its two accumulator copies are added and written to the work item's PATCH in
//     global memory; the p-side sums of the tile go there too.  combine_sym_kernel then gathers, for every
//     output pixel, the patches that hold a share of it (its own tile's p-side, the q-side rows of the tile
//     rows above it and of the horizontal neighbours whose halo covers it) in a fixed order, and
//     normalises.  The sweep of a tile can be split over `parts` workgroups (steps [s_a, s_b) each) with no
//     further mechanism: a part is just a patch.
//   * filter<float> (PAIR): two 1-channel buffers per launch ride in the (x, y) channels of the same planes;
//     they share the range weight, gate and normalise separately (the four sums per pixel become
//     Sigma w0 c0, Sigma w1 c1, Sigma w0, Sigma w1).
//   * Spec options as launch modes: pooled channels (a sum where the default has a max3, still symmetric); the
//     one-sided gate (one evaluation per pair, but two tests and two weights -- one per direction); the clamped
//     border (this kernel sums the taps inside the image, border_virtual_kernel in statmc_filter.hip adds the taps
//     beyond it for the pixels next to an edge, combine_sym_kernel joins the two).
#include <algorithm>
#include <array>
#include <cmath>
#include <functional>
#include <map>
#include <mutex>
#include <vector>
#include <set>
#include <type_traits>
#include <utility>

#include "statmc_device.h"
#include "statmc_filter_common.h"

#include "statmc_sym_experiments.h"   // the STATMC_SYM_* diagnostic / experiment switches: all zero in the product build

namespace statmc {
namespace sym {

constexpr int kHkAtEndForced = STATMC_SYM_HK_END;   // -1: per build (hk_at_end below); 0 / 1: every build (A/B)
constexpr bool kStamps = STATMC_SYM_STAMPS;    // diagnostic: per-wave clock sums per step (tools/experiments/stamps_sym.py)
constexpr bool kPipe = STATMC_SYM_PIPE;        // experiment: hand-placed LDS reads one phase ahead of the arithmetic
constexpr int kPrio = STATMC_SYM_PRIO;         // experiment (s_setprio)
constexpr int kSplit = STATMC_SYM_SPLIT;       // window half 0 sweeps dx <= kSplit, half 1 the rest
constexpr int kGSplitForced = STATMC_SYM_GSPLIT;    // -1: per build (gsplit_of below); 0: the window split at dx = kSplit (rounds 2 - 5); G > 0: every build
constexpr bool kGroupSplitRT = STATMC_SYM_GSPLIT_RT != 0;   // the runtime-radius builds: split at a read-group boundary too
constexpr int kAblate = STATMC_SYM_ABLATE;     // timing only: 1 no q side, 2 no row staging, 4 no flush, 8 no sweep arithmetic, 16 no barrier
// membership / buffer mode of a launch: one RGB buffer, every channel passes (default spec) | two float buffers
// (filter<float>) | one RGB buffer, channels pooled (STATMC_CHANNELS_JOINT: sum_c fma(d_c, d_c, -(D_p,c + D_q,c)) <= 0,
// as symmetric in (p, q) as the default test)
constexpr int kModeRgb = 0, kModePair = 1, kModeJoint = 2, kModeAsym = 3, kModeAsymJoint = 4;
// kModeAsym / kModeAsymJoint: the one-sided gate (STATMC_GATE_ASYMMETRIC), channels one by one / pooled.  The pair is
// still evaluated once, but its two directions are two tests -- fma(d, d, -D_q) <= D_p decides whether q enters p's
// sums, fma(d, d, -D_p) <= D_q whether p enters q's -- so the pair carries two weights.
// kModeCentre / kModeCentreJoint (STATMC_GATE_CENTRE: Moon et al. 2013, the reference's -DMEMFNC=1): d * d <= D_p decides q's
// membership in p's window, d * d <= D_q p's in q's -- two tests per pair like the one-sided gate, sharing one square.
constexpr int kModeCentre = 5, kModeCentreJoint = 6;
constexpr int kModes = 7;     // columns of the kernel table (the two Welch modes below have one build of their own)
constexpr bool mode_centre(int m) { return m == kModeCentre || m == kModeCentreJoint; }
constexpr bool mode_asym(int m) { return m == kModeAsym || m == kModeAsymJoint || mode_centre(m); }   // a pair carries two weights
// Welch-Satterthwaite degrees of freedom (STATMC_DOF_WELCH), channels one by one / pooled: the discriminator image holds
// v = s^2 / n, a fourth statistics image E = v^2 / (n - 1) is staged with it, and the pair looks its squared quantile up
// at floor(nu), nu = (v_p + v_q)^2 / (E_p + E_q) -- symmetric in (p, q) like everything else the pair needs, so the pair
// is still evaluated once.  These modes exist in ONE build per feature-plane count (register staging, runtime radius): 18 input
// + 8 accumulator planes leave no room for the LDS-DMA landing area; with eight feature planes the three E planes shrink to one
// plane n - 1 and the taps form E themselves (Planes<8, true>, chunk).
// Each in two builds: the first reads the quantiles from a band of the table in LDS and flags the items in which a pair
// asked for an entry outside it; the second ("far") reads the table in global memory and runs behind the first over the
// flagged items only (none in a film of uniform sample count).
// kModeWelchPair: filter<float>, two buffers per launch as in kModePair, each with its own sample counts and its own test.
constexpr int kModeWelch = 7, kModeWelchJoint = 8, kModeWelchFar = 9, kModeWelchJointFar = 10, kModeWelchPair = 11, kModeWelchPairFar = 12;
constexpr bool mode_welch(int m) { return m >= kModeWelch && m <= kModeWelchPairFar; }
constexpr bool mode_welch_joint(int m) { return m == kModeWelchJoint || m == kModeWelchJointFar; }
constexpr bool mode_welch_far(int m) { return m == kModeWelchFar || m == kModeWelchJointFar || m == kModeWelchPairFar; }
constexpr bool mode_pair(int m) { return m == 1 /* kModePair */ || m == kModeWelchPair || m == kModeWelchPairFar; }
// How the two waves of a row share the window, and when the staging waves keep house -- per build, measured (round 6; 1080p, r = 20,
// back to back, one box, ms: profiles/r06_modes.log; HISTORY.md 4.3d):
//                     RGB     pooled   one-sided  Moon    8 planes  2 float buffers
//   dx = 0, hk first  1.446   1.452    1.863      1.812   1.602     1.430      (rounds 2 - 5)
//   G = 6,  hk first  1.388   1.383    1.783      1.732   1.592     1.394
//   G = 6,  hk last   1.344   1.342    1.749      1.715   1.631     1.455
//   G = 7,  hk first  1.339   1.352    1.833      1.871   1.670     1.470
//   G = 7,  hk last   1.331   1.319    1.859      1.867   1.727     1.534
// G: half 0 sweeps read groups [0, G), half 1 the rest -- no group is evaluated by both waves, and the OLDER wave of a SIMD pair
// (half 0: it wins the issue arbitration and would otherwise wait at the barrier) takes the larger share.  hk last: flush, staging
// and LDS-DMA after the wave's sweep instead of before it.
// The split decides the ORDER of every pixel's sums, and the builds must agree on it: a block + halo image with 17 channels runs the
// eight-plane build where the whole film with the same G-buffers runs the six-plane one, and the two must leave the same bits
// (tests/test_peer_gpu.py caught a per-build G).  So ONE G for every build -- 6: best or within 1 % of the best everywhere -- and
// only the housekeeping's place, which changes no sum, is chosen per build.
// ... with ONE exception, decided by the G-buffer SET and not by the build (the last session of round 6): a call whose G-buffers are
// exactly two RGB images -- the shipped normal + albedo -- runs a six-plane build whether it filters the whole film or a 15-channel block +
// halo image, so those calls may have a G of their own: 7 under the default and the pooled test (G7 builds; 1.603 - 1.611 against 1.641 -
// 1.650 ms in the step at 1080p, three A/B pairs, profiles/r06_ab_g7_step.log).  Every other set keeps 6 in every build.
constexpr int gsplit_of(int, int, bool g7 = false) { return kGSplitForced >= 0 ? kGSplitForced : g7 ? 7 : 6; }
constexpr bool hk_at_end(int mode, int ng) {
    if (kHkAtEndForced >= 0) return kHkAtEndForced != 0;
    return ng == 6 && !mode_pair(mode);
}
constexpr int kR = 20;
constexpr int kPx = 4;                    // pixels per lane
constexpr int kW = 32 * kPx;              // 128 tile columns: half a wave per row
constexpr int kRows = 8;                  // tile rows: wave (t, h) owns rows t (lanes 0-31) and t + 4 (lanes 32-63)
constexpr int kSlots = kRows + 1;         // LDS row ring
constexpr int kP = kW + 2 * kR;           // 168 staged columns per row
constexpr int kQ = 8;                     // accumulator planes per row: 2 copies x (r, g, b, w)
constexpr int kThreads = 512;
constexpr int kSteps = kR + 1;            // dy = 0 .. 20
constexpr int kTabW = 2 * kR + 7;         // entries per window row of the spatial table: index dx + kR + 3
constexpr int kTabPad = 2 * (kTabW + 1);  // LDS copy: pairs (tab[t], tab[t+1])
constexpr int kChunks = 2 * kR / 4 + 1;   // 11 read groups per window row
constexpr int kMid = kChunks / 2;         // 5: the group that holds dx = 0
constexpr int kPatchP = kRows * kW;       // float4 per patch: p-side piece
// LDS-DMA staging: the four waves of window half 0 each fetch and stage their own 44 / 44 / 40 / 40 of the 168 columns
// of a row, five RGB images (or one 15-float AoS image) = 165 / 150 pieces of 16 B per wave and row, landing in a
// wave-private raw area.  Half 0 because its waves are the older ones of every SIMD pair: they win the issue
// arbitration, finish their sweep first and would otherwise idle at the barrier; while they wait for their fetches
// and LDS reads the half-1 wave of the SIMD sweeps (tools/experiments/stamps_sym.py).
constexpr int kWaveCols = 44;
__host__ __device__ inline int wave_col0(int wave) { return wave < 2 ? 44 * wave : 88 + 40 * (wave - 2); }
__host__ __device__ inline int wave_cols(int wave) { return wave < 2 ? 44 : wave < 4 ? 40 : 0; }
// NG = feature planes per staged row.  6: up to two RGB G-buffers (the shipped normal + albedo).  8: the same plus up to
// two 1-channel G-buffers (depth, material id: statpath.cpp:828-835, 1096-1130) -- 17 input + 8 accumulator planes per
// row: 9 x 25 x 168 floats = 151 200 B of ring, and the LDS-DMA landing area shrinks to exactly the 168 columns a row
// has (wave w's columns at wave_col0(w) x 17 floats) so that the lot still fits the CU's 160 KiB: 163 392 B.
// Welch: the part of the squared-quantile table the pairs of one item can ask for lives in LDS (what is left of the 160 KiB
// beside the ring): entries lo .. lo + kWelchBand - 1, lo = (least n of the item's pixels) - 2.  nu lies between
// min(n_p, n_q) - 1 and n_p + n_q - 2, so a film of uniform n <= 720 never leaves the band; a pair that does takes the
// table in global memory (a wave-uniform branch).
constexpr int kWelchBand = 1440;
struct WelchTab {
    const float *table;     // global: entries 0 .. 4096 (entry 0 = entry 1)
    const float *band;      // LDS, at address 0: band[j] = table[min(lo + j, 4096)]
    unsigned neg4lo;        // -4 lo: (dof << 2) + neg4lo is the byte offset of a quantile in the band
    unsigned oob;           // a dof >= oob lies beyond the band (~0u when the band reaches the table's end)
};

template <int NG, bool W = false>
struct Planes {
    // input planes per row: features, mean, -D, colour[, Welch: the three planes E -- or, beside eight feature planes, where
    // the CU's LDS has room for ONE more plane, n - 1: the taps form E = v * v / (n - 1) themselves (chunk)]
    static constexpr int kIn = NG + 9 + (W ? (NG == 8 ? 1 : 3) : 0);
    static constexpr int cMC = NG, cND = NG + 3, cCOL = NG + 6, cE = NG + 9;
    static constexpr int kSlotFloats = (kIn + kQ) * kP;
    static constexpr int kRawTotal = W ? 0 : NG == 6 ? 4 * kWaveCols * 15 : kP * kIn;
    static constexpr int kBandTotal = W ? kWelchBand + 16 : 0;   // Welch: a band of the squared-quantile table + the item's {min n, max n}
    static constexpr int kTabFloats = 2 * kTabPad;   // the spatial table in LDS: two rows (this step's and the next one's)
    static constexpr size_t kLdsBytes = (size_t)(kSlots * kSlotFloats + kTabFloats + kRawTotal + kBandTotal) * sizeof(float);
    __host__ __device__ static inline int raw_off(int wave) { return W ? 0 : NG == 6 ? (wave & 3) * kWaveCols * 15 : wave_col0(wave & 3) * kIn; }
};
static_assert(Planes<6>::kLdsBytes <= 160 * 1024 && Planes<8>::kLdsBytes <= 160 * 1024 && Planes<6, true>::kLdsBytes <= 160 * 1024 &&
                  Planes<8, true>::kLdsBytes <= 160 * 1024, "LDS budget");

// steps = window rows dy = 0 .. r a tile sweeps: kSteps in the r = 20 builds, radius + 1 in the runtime-radius ones
__host__ __device__ inline int step_lo(int part, int n_parts, int steps) { return (steps * part) / n_parts; }
__host__ __device__ inline int q_rows_max(int n_parts, int steps) { return (steps + n_parts - 1) / n_parts + kRows - 1; }  // rows s_a .. s_b+6

// The lane's own 4 pixels.  Their values enter the packed instructions as broadcasts of ONE half of a register
// pair (op_sel), so two different scalars share every pair: 6 + 6 + 4 pairs per pixel instead of 15 + 15
// registers holding (x, x) duplicates -- which is what a `v2f{x, x}` splat of a scalar compiles to.
struct LaneWelch { v2f pe[kPx][2]; unsigned far; };   // Welch: (E_r, E_g), (E_b, -) of the lane's pixels; the greatest dof asked for
struct LaneNoWelch {};
template <int NG, bool W = false>
struct Lane : std::conditional<W, LaneWelch, LaneNoWelch>::type {
    v2f pg[kPx][NG / 2];   // scaled features: (n.x, n.y), (n.z, a.x), (a.y, a.z)[, (depth, material id)]
    v2f ms[kPx][3];    // per channel (corrected mean, discriminator)
    v2f pc[kPx][2];    // colour (r, g), (b, -)
    v2f acc[kPx][3];   // .x even taps, .y odd taps of every read group
    v2f sw[kPx];
#if STATMC_SYM_COUNT
    unsigned n_groups = 0, n_empty = 0, n_empty_halves = 0;
#endif
};
// The three packed instructions that take a broadcast half of a pair, spelled out: written as shuffles the
// broadcasts are loop-invariant, get hoisted out of the sweep and come back as (x, x) register pairs of their own.
// `half` is a constant after unrolling.
__device__ __forceinline__ v2f sub_bc(const v2f &pair, int half, const v2f &t) {   // bc(pair.half) - t
    v2f d;
    if (half) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(pair), "v"(t));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(pair), "v"(t));
    return d;
}
__device__ __forceinline__ v2f rsub_bc(const v2f &t, const v2f &pair, int half) {  // t - bc(pair.half)
    v2f d;
    if (half) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(t), "v"(pair));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(t), "v"(pair));
    return d;
}
__device__ __forceinline__ v2f add_bc(const v2f &pair, int half, const v2f &t) {   // bc(pair.half) + t
    v2f d;
    if (half) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(pair), "v"(t));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(pair), "v"(t));
    return d;
}
__device__ __forceinline__ v2f fma_bc(const v2f &w, const v2f &pair, int half, const v2f &acc) {  // w * bc(pair.half) + acc
    v2f d;
    if (half) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(w), "v"(pair), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(w), "v"(pair), "v"(acc));
    return d;
}

__device__ __forceinline__ unsigned cvt_u32(float x) {   // truncation; NaN and negative -> 0, above 2^32 - 1 saturates (the hardware's rule)
    unsigned i;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(i) : "v"(x));
    return i;
}
__device__ __forceinline__ unsigned lshl2_add_u32(unsigned a, unsigned b) {   // (a << 2) + b, b wave-uniform
    unsigned d;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(d) : "v"(a), "s"(b));
    return d;
}
__device__ __forceinline__ float lds_at0(unsigned byte_offset) {   // the float at an LDS address (the Welch band starts at 0)
    return *reinterpret_cast<const __attribute__((address_space(3))) float *>((unsigned long)byte_offset);
}
__device__ __forceinline__ unsigned max3_u32(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float min_finite(float x) {   // min(x, FLT_MAX); a NaN becomes FLT_MAX (v_min_f32 returns the other operand)
    float r;
    asm("v_min_f32 %0, 0x7f7fffff, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ v2f fma_sq_nbc(const v2f &d, const v2f &pair) {  // d * d - bc(pair.y)
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(d), "v"(pair));
    return r;
}

// bit (i*4+k): tap i of read group J lies at dx in [LO, HI] from pixel k
template <int J, int LO, int HI>
struct Mask {
    static constexpr unsigned value() {
        unsigned m = 0;
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) {
                const int dx = 4 * J - kR + i - k;
                if (dx >= LO && dx <= HI) m |= 1u << (i * 4 + k);
            }
        return m;
    }
};

// tap pair H (taps 2H, 2H+1) of a read group against pixel k
template <int H, unsigned MASK>
struct Taps {
    static constexpr int i0 = 2 * H;
    static constexpr bool in0(int k) { return (MASK & (1u << (i0 * 4 + k))) != 0; }
    static constexpr bool in1(int k) { return (MASK & (1u << ((i0 + 1) * 4 + k))) != 0; }
    static constexpr bool on(int k) { return in0(k) || in1(k); }
    static constexpr bool any() { return on(0) || on(1) || on(2) || on(3); }
};

template <int H>
__device__ __forceinline__ v2f pair_of(const v4f &v) {
    return H == 0 ? __builtin_shufflevector(v, v, 0, 1) : __builtin_shufflevector(v, v, 2, 3);
}

// range weight exponent of tap pair H against the lane's 4 pixels: tab - |k_n dn|^2 - |k_a da|^2 (log2 domain)
template <int H, unsigned MASK, int NG, class LaneT>
__device__ __forceinline__ void range_exponent(const LaneT &st, const v4f *g, const float *__restrict__ tab, int j, v2f (&e)[kPx]) {
    using M = Taps<H, MASK>;
    v2f tp[kPx];
#pragma unroll
    for (int k = 0; k < kPx; k++) if (M::on(k)) tp[k] = *reinterpret_cast<const v2f *>(tab + 2 * (4 * j + 2 * H - k + 3));
#pragma unroll
    // (the spatial term as the first addend of the chain saves one packed add per pair and is 14 % SLOWER: the chain
    // then starts behind the table read)
    for (int k = 0; k < kPx; k++) if (M::on(k)) { const v2f d = sub_bc(st.pg[k][0], 0, pair_of<H>(g[0])); e[k] = -d * d; }
#pragma unroll
    for (int ch = 1; ch < NG; ch++) {
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) { const v2f d = sub_bc(st.pg[k][ch >> 1], ch & 1, pair_of<H>(g[ch])); e[k] = __builtin_elementwise_fma(-d, d, e[k]); }
    }
#pragma unroll
    for (int k = 0; k < kPx; k++) if (M::on(k)) e[k] += tp[k];
}

// membership gate and weight of tap pair H: w = member ? exp2(e) : 0   (mcn: corrected mean planes 0..2, -D planes 3..5)
// PAIR (two float buffers in the (x, y) channels): one weight per buffer, w for buffer 0 and wb for buffer 1 -- the
// buffers share the range weight and gate separately (filter<float>: every buffer is its own 1-channel test).
// Welch (en: the three E planes; tq2: the device's table of SQUARED quantiles, indexed by dof): per channel
//     nu = s^2 / (E_p + E_q),  s = v_p + v_q;   dof = nu >= 1 ? min((int)nu, 4096) : 1;   u = fma(d, d, -(t_dof^2 * s))
// The quotient is a reciprocal with one residual correction: correctly rounded except in rare last-bit cases, which can
// move a pair whose nu lies within an ulp of an integer to the neighbouring table entry.  The correction is not optional:
// nu is an exact integer whenever one side's variance is 0 (nu = n_p - 1 then: black next to lit pixels), and the bare
// s^2 * rcp(.) lands below it half of the time (tried; tests/test_gpu_parity.py caught it at 3 spp).  The first product
// goes through min(., FLT_MAX), which also turns a NaN into FLT_MAX: with that the correction reproduces the oracle's
// quotient in the special cases too -- E_p + E_q = 0 gives inf (s^2 > 0) or NaN (0 / 0), an infinite E_p + E_q gives
// NaN where the oracle has 0 -- both dof 1 --, infinite s^2 gives inf or NaN as there; no select on the result.
// A NaN becomes dof 1 in the conversion, exactly the oracle's `nu >= 1.f` branch.
template <int H, unsigned MASK, int MODE, int NG, class LaneT>
__device__ __forceinline__ void gate_weight(LaneT &st, const v4f *mcn, const v4f *en, const WelchTab &tq2,
                                            const v2f (&e)[kPx], v2f (&w)[kPx], v2f (&wb)[kPx]) {
    using M = Taps<H, MASK>;
    constexpr bool PAIR = mode_pair(MODE);
    constexpr int NC = PAIR ? 2 : 3;
    if constexpr (mode_welch(MODE)) {
        // Software pipeline over the lane's pixels: the table look-ups of pixel k are in flight (six ds_read_b32 from the band
        // in LDS; the far build: six loads from the table in global memory, 1 ms more at 256 spp, where a wave's 64 lanes
        // spread over eight cache lines) while pixel k - 1 is tested -- issued one pixel at a time they cost a full latency
        // per channel (the file is compiled without the machine scheduler: source order is issue order).
        v2f t2[2][3];
        auto lookup = [&](int k, v2f (&out)[3]) {
            unsigned dx[3] = {0u, 0u, 0u}, dy[3] = {0u, 0u, 0u};   // (PAIR: two channels = the two buffers)
            // (all six reciprocals ahead of the corrections -- no s_nop behind them -- needs twelve more registers and times
            // the same: 3.385 against 3.361 ms)
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                const v2f sn = rsub_bc(pair_of<H>(mcn[3 + ch]), st.ms[k][ch], 1);      // -(v_p + v_q)
                const v2f den = add_bc(st.pe[k][ch >> 1], ch & 1, pair_of<H>(en[ch]));   // E_p + E_q
                const v2f s2 = sn * sn;
                v2f nu;
                if constexpr ((STATMC_SYM_WELCH_ABLATE & 4) != 0) {   // timing only
                    nu = s2 + den;
                } else {
                    const v2f r = v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                    const v2f p = s2 * r;
                    const v2f q0 = v2f{min_finite(p.x), min_finite(p.y)};
                    const v2f er = __builtin_elementwise_fma(-q0, den, s2);
                    nu = __builtin_elementwise_fma(er, r, q0);
                }
                // v_cvt_u32_f32 truncates, sends a NaN and everything below 1 to 0 and saturates above (spelled out: the C++
                // conversion of such values is undefined); entry 0 of the table is entry 1, which finishes the oracle's
                // `nu >= 1 ? min((int)nu, 4096) : 1` but for the upper clamp -- that is in the band's fill
                dx[ch] = cvt_u32(nu.x);
                dy[ch] = cvt_u32(nu.y);
                if constexpr (mode_welch_far(MODE)) {
                    out[ch] = v2f{tq2.table[min(dx[ch], 4096u)], tq2.table[min(dy[ch], 4096u)]};
                } else if constexpr ((STATMC_SYM_WELCH_ABLATE & 1) != 0) {   // timing only: no gather
                    out[ch] = v2f{(float)dx[ch] * 1e-3f + 9.f, (float)dy[ch] * 1e-3f + 9.f};
                } else {
                    // byte offset in the band: the table's upper clamp, one shift-and-add, one minimum.  A dof beyond the band's
                    // end reads its LAST entry (and flags the item, below: the far build computes it again from the whole
                    // table); a dof below the band's start wraps around (unsigned) and lands on the last entry too -- that is
                    // dof 0 next to sample counts > 2, i.e. a NaN nu from 0 / 0, where s = 0 makes the quantile immaterial
                    // (or, with E_p + E_q denormal, a quotient without a correct digit).  The clamp to 4096 comes first
                    // (ADVICE r4): without it a garbage dof >= 2^30 wrapped in the shift and could land INSIDE the band.
                    const unsigned bx = min(lshl2_add_u32(min(dx[ch], 4096u), tq2.neg4lo), 4u * (unsigned)kWelchBand - 4u);
                    const unsigned by = min(lshl2_add_u32(min(dy[ch], 4096u), tq2.neg4lo), 4u * (unsigned)kWelchBand - 4u);
                    out[ch] = v2f{lds_at0(bx), lds_at0(by)};
                }
            }
            // the greatest dof of the item: beyond the band's end (films whose sample counts differ by more than the band
            // holds) the item is flagged and computed again by the build that reads the table in global memory
            if constexpr (!mode_welch_far(MODE)) st.far = max3_u32(st.far, max3_u32(dx[0], dy[0], dx[1]), max3_u32(dy[1], dx[2], dy[2]));
        };
        auto test = [&](int k, const v2f (&tt)[3]) {
            v2f u[3];
#pragma unroll
            for (int ch = 0; ch < NC; ch++) {
                const v2f d = sub_bc(st.ms[k][ch], 0, pair_of<H>(mcn[ch]));
                const v2f sn = rsub_bc(pair_of<H>(mcn[3 + ch]), st.ms[k][ch], 1);      // (the compiler keeps the look-up's)
                u[ch] = __builtin_elementwise_fma(d, d, tt[ch] * sn);
            }
            w[k] = v2f{__builtin_amdgcn_exp2f(e[k].x), __builtin_amdgcn_exp2f(e[k].y)};
            if constexpr (PAIR) {   // every buffer its own test (a pixel that takes no part in a buffer has a NaN mean there)
                const v2f x = w[k];
                w[k] = v2f{M::in0(k) && u[0].x <= 0.f ? x.x : 0.f, M::in1(k) && u[0].y <= 0.f ? x.y : 0.f};
                wb[k] = v2f{M::in0(k) && u[1].x <= 0.f ? x.x : 0.f, M::in1(k) && u[1].y <= 0.f ? x.y : 0.f};
            } else if constexpr (mode_welch_joint(MODE)) {
                const v2f m = (u[0] + u[1]) + u[2];
                w[k] = v2f{M::in0(k) && m.x <= 0.f ? w[k].x : 0.f, M::in1(k) && m.y <= 0.f ? w[k].y : 0.f};
            } else {
                const float m0 = __builtin_fmaxf(__builtin_fmaxf(u[0].x, u[1].x), u[2].x);
                const float m1 = __builtin_fmaxf(__builtin_fmaxf(u[0].y, u[1].y), u[2].y);
                w[k] = v2f{M::in0(k) && m0 <= 0.f ? w[k].x : 0.f, M::in1(k) && m1 <= 0.f ? w[k].y : 0.f};
            }
        };
        int prev = -1;      // (constant after unrolling)
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) {
            lookup(k, t2[k & 1]);
            if (prev >= 0) test(prev, t2[prev & 1]);
            prev = k;
        }
        if (prev >= 0) test(prev, t2[prev & 1]);
        return;
    }
    if constexpr (mode_asym(MODE)) {
        // one-sided gate: w decides q's membership in p's window (p side), wb p's membership in q's (q side); plain
        // compares, as the oracle writes them (a NaN statistic fails every one)
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) {
            v2f upq[3], uqp[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const v2f d = sub_bc(st.ms[k][ch], 0, pair_of<H>(mcn[ch]));
                if constexpr (mode_centre(MODE)) {
                    upq[ch] = uqp[ch] = d * d;                                          // against D_p, and against D_q
                } else {
                    upq[ch] = __builtin_elementwise_fma(d, d, pair_of<H>(mcn[3 + ch]));   // fma(d, d, -D_q)
                    uqp[ch] = fma_sq_nbc(d, st.ms[k][ch]);                                 // fma(d, d, -D_p)
                }
            }
            const v2f x = v2f{__builtin_amdgcn_exp2f(e[k].x), __builtin_amdgcn_exp2f(e[k].y)};
            bool p0, p1, q0, q1;
            if constexpr (MODE == kModeAsymJoint || MODE == kModeCentreJoint) {
                const v2f lp = (upq[0] + upq[1]) + upq[2], lq = (uqp[0] + uqp[1]) + uqp[2];
                const float rp = (st.ms[k][0].y + st.ms[k][1].y) + st.ms[k][2].y;
                const v2f rq = -((pair_of<H>(mcn[3]) + pair_of<H>(mcn[4])) + pair_of<H>(mcn[5]));   // (D_q0 + D_q1) + D_q2
                p0 = lp.x <= rp; p1 = lp.y <= rp;
                q0 = lq.x <= rq.x; q1 = lq.y <= rq.y;
            } else {
                p0 = p1 = q0 = q1 = true;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const v2f dq = -pair_of<H>(mcn[3 + ch]);
                    p0 = p0 && upq[ch].x <= st.ms[k][ch].y; p1 = p1 && upq[ch].y <= st.ms[k][ch].y;
                    q0 = q0 && uqp[ch].x <= dq.x; q1 = q1 && uqp[ch].y <= dq.y;
                }
            }
            w[k] = v2f{M::in0(k) && p0 ? x.x : 0.f, M::in1(k) && p1 ? x.y : 0.f};
            wb[k] = v2f{M::in0(k) && q0 ? x.x : 0.f, M::in1(k) && q1 ? x.y : 0.f};
        }
        return;
    }
    // two pixels at a time: 6 independent chains are enough to keep the pipe busy and halve the live statistics
#pragma unroll
    for (int k0 = 0; k0 < kPx; k0 += 2) {
        v2f u[2][3];
        // membership statistic per channel: fma(d, d, -(D_p + D_q))
#pragma unroll
        for (int ch = 0; ch < NC; ch++) {
#pragma unroll
            for (int kk = 0; kk < 2; kk++) if (M::on(k0 + kk)) {
                const v2f d = sub_bc(st.ms[k0 + kk][ch], 0, pair_of<H>(mcn[ch]));
                const v2f s = rsub_bc(pair_of<H>(mcn[3 + ch]), st.ms[k0 + kk][ch], 1);
                u[kk][ch] = __builtin_elementwise_fma(d, d, s);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) if (M::on(k0 + kk)) w[k0 + kk] = v2f{__builtin_amdgcn_exp2f(e[k0 + kk].x), __builtin_amdgcn_exp2f(e[k0 + kk].y)};
        if constexpr (PAIR) {
            // a pixel that takes no part in a buffer is staged with NaN in that buffer's mean: the compare fails
#pragma unroll
            for (int kk = 0; kk < 2; kk++) if (M::on(k0 + kk)) {
                const int k = k0 + kk;
                const v2f x = w[k];
                w[k] = v2f{M::in0(k) && u[kk][0].x <= 0.f ? x.x : 0.f, M::in1(k) && u[kk][0].y <= 0.f ? x.y : 0.f};
                wb[k] = v2f{M::in0(k) && u[kk][1].x <= 0.f ? x.x : 0.f, M::in1(k) && u[kk][1].y <= 0.f ? x.y : 0.f};
            }
        } else if constexpr (MODE == kModeJoint) {
            // pooled channels: (u_0 + u_1) + u_2 <= 0, the oracle's order; a NaN anywhere fails the compare
#pragma unroll
            for (int kk = 0; kk < 2; kk++) if (M::on(k0 + kk)) {
                const int k = k0 + kk;
                const v2f m = (u[kk][0] + u[kk][1]) + u[kk][2];
                w[k] = v2f{M::in0(k) && m.x <= 0.f ? w[k].x : 0.f, M::in1(k) && m.y <= 0.f ? w[k].y : 0.f};
            }
        } else {
            // all three channels pass <=> max_c <= 0; v_max3 drops NaN operands, which is why a pixel that takes no
            // part is staged with NaN in all three channels of its mean
#pragma unroll
            for (int kk = 0; kk < 2; kk++) if (M::on(k0 + kk)) {
                const int k = k0 + kk;
                const float m0 = __builtin_fmaxf(__builtin_fmaxf(u[kk][0].x, u[kk][1].x), u[kk][2].x);
                const float m1 = __builtin_fmaxf(__builtin_fmaxf(u[kk][0].y, u[kk][1].y), u[kk][2].y);
                w[k] = v2f{M::in0(k) && m0 <= 0.f ? w[k].x : 0.f, M::in1(k) && m1 <= 0.f ? w[k].y : 0.f};
            }
        }
    }
}

// p side: the lane's pixels collect w * colour_q; q side (SYM): the taps' accumulators collect w * colour_p.
// PAIR: the four sums per pixel are (sum w0 c0, sum w1 c1, sum w0, sum w1) -- acc[0], acc[1], acc[2], sw -- instead of
// (sum w r, sum w g, sum w b, sum w); the same four packed operations per side.
template <int H, unsigned MASK, bool SYM, int MODE, int NG, class LaneT>
__device__ __forceinline__ void accumulate(LaneT &st, const v4f *col, const v2f (&w)[kPx], const v2f (&wb)[kPx], v2f (&qv)[4]) {
    using M = Taps<H, MASK>;
    constexpr bool PAIR = mode_pair(MODE);
    if constexpr (PAIR) {
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) { st.acc[k][2] += w[k]; st.sw[k] += wb[k]; }
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) {
            st.acc[k][0] = __builtin_elementwise_fma(w[k], pair_of<H>(col[0]), st.acc[k][0]);
            st.acc[k][1] = __builtin_elementwise_fma(wb[k], pair_of<H>(col[1]), st.acc[k][1]);
        }
        if constexpr (SYM) {
#pragma unroll
            for (int k = 0; k < kPx; k++) if (M::on(k)) {
                qv[0] = fma_bc(w[k], st.pc[k][0], 0, qv[0]);
                qv[1] = fma_bc(wb[k], st.pc[k][0], 1, qv[1]);
            }
#pragma unroll
            for (int k = 0; k < kPx; k++) if (M::on(k)) { qv[2] += w[k]; qv[3] += wb[k]; }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kPx; k++) if (M::on(k)) st.sw[k] += w[k];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) st.acc[k][ch] = __builtin_elementwise_fma(w[k], pair_of<H>(col[ch]), st.acc[k][ch]);
    }
    if constexpr (SYM) {
        // the q side takes the weight of the pair's other direction under the one-sided gate
        const v2f (&wq)[kPx] = mode_asym(MODE) ? wb : w;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
#pragma unroll
            for (int k = 0; k < kPx; k++) if (M::on(k)) qv[ch] = fma_bc(wq[k], st.pc[k][ch >> 1], ch & 1, qv[ch]);
        }
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) qv[3] += wq[k];
    }
}

// LDS byte address of a pointer into the workgroup's shared array
__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float *)p;
}
// ds_read_b128 placed by hand: the compiler sinks its own loads next to their first use, which leaves the LDS
// latency of every phase exposed (a wave is mostly alone on its SIMD while it sweeps: the older wave of a pair wins
// the issue arbitration, runs ahead and waits at the barrier -- tools/experiments/stamps_sym.py).  The register
// is written when the data lands: every use sits behind an lds_wait that names it.
template <int OFF>
__device__ __forceinline__ v4f lds_read128(unsigned addr) {
    v4f v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// wait until at most N LDS operations issued after the named registers' loads are outstanding (in-order return)
template <int N>
__device__ __forceinline__ void lds_wait(v4f &a, v4f &b, v4f &c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(v4f &a, v4f &b, v4f &c, v4f &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// One read group (4 taps) against the lane's 4 pixels.  Every LDS operand is one ds_read_b128 of a channel plane
// (consecutive lanes read consecutive 16 B: conflict-free); the group runs in three phases -- range exponents
// (6 feature planes), gates and weights (6 statistics planes), accumulation (3 colour planes + the 4 accumulator
// planes of the taps).  PIPE: the statistics planes are requested before the feature planes and the colour /
// accumulator planes before the gates, by hand-placed reads, so that two of the three phases find their operands
// in registers; otherwise the compiler's own loads (each phase waits for its operands).
template <unsigned MASK, bool SYM, bool PIPE, int MODE, int NG, class LaneT>
__device__ __forceinline__ void chunk(LaneT &st, const float *__restrict__ row, const float *__restrict__ tab, float *__restrict__ qrow, int j,
                                      const WelchTab &tq2) {
    using M0 = Taps<0, MASK>;
    using M1 = Taps<1, MASK>;
    constexpr bool PAIR = mode_pair(MODE);
    constexpr bool W = mode_welch(MODE);
    static_assert(!(W && PIPE), "the Welch modes use the compiler-placed reads");
    constexpr int C_MC = Planes<NG, W>::cMC, C_COL = Planes<NG, W>::cCOL;
    const float *r = row + 4 * j;
    v4f g[NG], mcn[6], col[3], q4[4], en[3];
    v2f e0[kPx], e1[kPx], w0[kPx], w1[kPx], wb0[kPx], wb1[kPx];   // wb*: second buffer's weights (PAIR)
    unsigned ra = 0, qa = 0;
    if constexpr ((kAblate & 32) != 0) {   // timing only: operands from nowhere (no LDS reads in the sweep)
#pragma unroll
        for (int ch = 0; ch < NG; ch++) asm volatile("" : "=v"(g[ch]));
#pragma unroll
        for (int ch = 0; ch < 6; ch++) asm volatile("" : "=v"(mcn[ch]));
#pragma unroll
        for (int ch = 0; ch < 3; ch++) asm volatile("" : "=v"(col[ch]));
#pragma unroll
        for (int v = 0; v < 4; v++) asm volatile("" : "=v"(q4[v]));
        if constexpr (M0::any()) range_exponent<0, MASK, NG>(st, g, tab, j, e0);
        if constexpr (M1::any()) range_exponent<1, MASK, NG>(st, g, tab, j, e1);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) asm volatile("" : "=v"(en[ch]));
        if constexpr (M0::any()) gate_weight<0, MASK, MODE, NG>(st, mcn, en, tq2, e0, w0, wb0);
        if constexpr (M1::any()) gate_weight<1, MASK, MODE, NG>(st, mcn, en, tq2, e1, w1, wb1);
        v2f qa2[4], qb2[4];
#pragma unroll
        for (int v = 0; v < 4; v++) { qa2[v] = pair_of<0>(q4[v]); qb2[v] = pair_of<1>(q4[v]); }
        if constexpr (M0::any()) accumulate<0, MASK, SYM, MODE, NG>(st, col, w0, wb0, qa2);
        if constexpr (M1::any()) accumulate<1, MASK, SYM, MODE, NG>(st, col, w1, wb1, qb2);
        if constexpr (SYM) {
#pragma unroll
            for (int v = 0; v < 4; v++) asm volatile("" ::"v"(qa2[v]), "v"(qb2[v]));
        }
        return;
    }
    if constexpr (PIPE) {
        ra = lds_addr(r);
        qa = lds_addr(qrow + 4 * j);
        mcn[0] = lds_read128<(C_MC + 0) * kP * 4>(ra);
        mcn[1] = lds_read128<(C_MC + 1) * kP * 4>(ra);
        mcn[2] = lds_read128<(C_MC + 2) * kP * 4>(ra);
        mcn[3] = lds_read128<(C_MC + 3) * kP * 4>(ra);
        mcn[4] = lds_read128<(C_MC + 4) * kP * 4>(ra);
        mcn[5] = lds_read128<(C_MC + 5) * kP * 4>(ra);
    }
#pragma unroll
    for (int ch = 0; ch < NG; ch++) g[ch] = *reinterpret_cast<const v4f *>(r + ch * kP);
    if constexpr (M0::any()) range_exponent<0, MASK, NG>(st, g, tab, j, e0);
    if constexpr (M1::any()) range_exponent<1, MASK, NG>(st, g, tab, j, e1);
    if constexpr (PIPE) {
        col[0] = lds_read128<(C_COL + 0) * kP * 4>(ra);
        col[1] = lds_read128<(C_COL + 1) * kP * 4>(ra);
        col[2] = lds_read128<(C_COL + 2) * kP * 4>(ra);
        if constexpr (SYM) {
            q4[0] = lds_read128<0 * kP * 4>(qa);
            q4[1] = lds_read128<1 * kP * 4>(qa);
            q4[2] = lds_read128<2 * kP * 4>(qa);
            q4[3] = lds_read128<3 * kP * 4>(qa);
        }
        // the statistics planes were requested before the feature planes, which phase 1 has consumed
        lds_wait<SYM ? 7 : 3>(mcn[0], mcn[1], mcn[2]);
        lds_wait<SYM ? 7 : 3>(mcn[3], mcn[4], mcn[5]);
    } else {
        // (PAIR: the third channel's planes hold nothing and are not read)
#pragma unroll
        for (int ch = 0; ch < 6; ch++)
            if (!(PAIR && ch % 3 == 2)) mcn[ch] = *reinterpret_cast<const v4f *>(r + (C_MC + ch) * kP);
        if constexpr (W && NG == 8) {
            // eight feature planes: the ring holds n - 1 and the taps divide -- the oracle's own expression, Dp * Dp / ((float)n - 1.f),
            // an IEEE division each (12 per read group; a reciprocal with a correction is a last bit off now and then, which
            // moves nu across an integer exactly where one side's variance is 0).  A pixel that takes no part has -D = 0 and a
            // NaN mean: whatever E comes out, its weight is 0.  PAIR: the second buffer's n - 1 sits in the (otherwise unread)
            // third channel's mean plane.
            const v4f nm1 = *reinterpret_cast<const v4f *>(r + Planes<NG, W>::cE * kP);
            const v4f nm1b = PAIR ? *reinterpret_cast<const v4f *>(r + (C_MC + 2) * kP) : nm1;
            en[0] = (mcn[3] * mcn[3]) / nm1;
            en[1] = (mcn[4] * mcn[4]) / nm1b;
            if constexpr (!PAIR) en[2] = (mcn[5] * mcn[5]) / nm1;
        } else if constexpr (W) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) en[ch] = *reinterpret_cast<const v4f *>(r + (Planes<NG, W>::cE + ch) * kP);
        }
    }
    if constexpr (M0::any()) gate_weight<0, MASK, MODE, NG>(st, mcn, en, tq2, e0, w0, wb0);
    if constexpr (M1::any()) gate_weight<1, MASK, MODE, NG>(st, mcn, en, tq2, e1, w1, wb1);
#if STATMC_SYM_COUNT
    if constexpr (MASK == 0xFFFFu && SYM && MODE == kModeRgb) {   // full read groups of the default mode
        bool any0 = false, any1 = false;
#pragma unroll
        for (int k = 0; k < kPx; k++) {
            any0 = any0 || w0[k].x != 0.f || w0[k].y != 0.f;
            any1 = any1 || w1[k].x != 0.f || w1[k].y != 0.f;
        }
        const bool e0w = __builtin_amdgcn_ballot_w64(any0) == 0, e1w = __builtin_amdgcn_ballot_w64(any1) == 0;
        st.n_groups++;
        st.n_empty += (e0w && e1w) ? 1u : 0u;
        st.n_empty_halves += (e0w ? 1u : 0u) + (e1w ? 1u : 0u);
    }
#endif
    if constexpr (PIPE) {
        // ... and the colour / accumulator planes before the gates; the wait names the weights too, so that it stays
        // behind the arithmetic that produced them (plain arithmetic may otherwise be scheduled after the wait)
        if constexpr (M0::any()) asm volatile("" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]));
        if constexpr (M1::any()) asm volatile("" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]));
        lds_wait<0>(col[0], col[1], col[2]);
        if constexpr (SYM) lds_wait<0>(q4[0], q4[1], q4[2], q4[3]);
    } else {
#pragma unroll
        for (int ch = 0; ch < (PAIR ? 2 : 3); ch++) col[ch] = *reinterpret_cast<const v4f *>(r + (C_COL + ch) * kP);
        if constexpr (SYM) {
#pragma unroll
            for (int v = 0; v < 4; v++) q4[v] = *reinterpret_cast<const v4f *>(qrow + 4 * j + v * kP);
        }
    }
    v2f qa2[4], qb2[4];
    if constexpr (SYM) {
#pragma unroll
        for (int v = 0; v < 4; v++) { qa2[v] = pair_of<0>(q4[v]); qb2[v] = pair_of<1>(q4[v]); }
    }
    if constexpr (M0::any()) accumulate<0, MASK, SYM, MODE, NG>(st, col, w0, wb0, qa2);
    if constexpr (M1::any()) accumulate<1, MASK, SYM, MODE, NG>(st, col, w1, wb1, qb2);
    if constexpr (SYM) {
        if constexpr ((kAblate & 64) != 0) {   // timing only: no write-back of the accumulators
#pragma unroll
            for (int v = 0; v < 4; v++) asm volatile("" ::"v"(qa2[v]), "v"(qb2[v]));
        } else {
#pragma unroll
            for (int v = 0; v < 4; v++)
                *reinterpret_cast<v4f *>(qrow + 4 * j + v * kP) = v4f{qa2[v].x, qa2[v].y, qb2[v].x, qb2[v].y};
        }
    }
}

// Taps dx in [LO, HI] of one window row: the read groups that lie wholly inside the range run as a rolled loop, the
// (at most two per side) groups cut by an end of the range get their static masks.
template <int J, int LO, int HI>
constexpr unsigned mask_of() { return (J >= 0 && J < kChunks) ? Mask<(J >= 0 && J < kChunks) ? J : 0, LO, HI>::value() : 0u; }
template <int LO, int HI>
struct Range {
    static constexpr unsigned kFull = 0xFFFFu;
    static constexpr unsigned m(int j) {
        unsigned r = 0;
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) {
                const int dx = 4 * j - kR + i - k;
                if (dx >= LO && dx <= HI) r |= 1u << (i * 4 + k);
            }
        return r;
    }
    static constexpr int first() { for (int j = 0; j < kChunks; j++) if (m(j)) return j; return kChunks; }
    static constexpr int last() { for (int j = kChunks - 1; j >= 0; j--) if (m(j)) return j; return -1; }
    static constexpr int first_full() { for (int j = 0; j < kChunks; j++) if (m(j) == kFull) return j; return kChunks; }
    static constexpr int last_full() { for (int j = kChunks - 1; j >= 0; j--) if (m(j) == kFull) return j; return -1; }
};

template <int LO, int HI, bool SYM, int MODE, int NG, class LaneT>
__device__ __forceinline__ void sweep_range(LaneT &st, const float *row, const float *tab, float *qrow, const WelchTab &tq2) {
    using R = Range<LO, HI>;
    constexpr int j0 = R::first(), j1 = R::last(), f0 = R::first_full(), f1 = R::last_full();
    static_assert(j0 <= j1, "empty range");
    constexpr bool has_full = f0 <= f1;
    constexpr int lo_end = has_full ? f0 : j1 + 1;      // masked groups j0 .. lo_end-1, full f0 .. f1, masked f1+1 .. j1
    static_assert(lo_end - j0 <= 2 && (!has_full || j1 - f1 <= 2), "more than two cut groups at an end");
    if constexpr (j0 < lo_end) chunk<R::m(j0), SYM, kPipe && !mode_welch(MODE), MODE, NG>(st, row, tab, qrow, j0, tq2);
    if constexpr (j0 + 1 < lo_end) chunk<R::m(j0 + 1), SYM, kPipe && !mode_welch(MODE), MODE, NG>(st, row, tab, qrow, j0 + 1, tq2);
    if constexpr (has_full) {
#pragma unroll 1
        for (int j = f0; j <= f1; j++) chunk<R::kFull, SYM, kPipe && !mode_welch(MODE), MODE, NG>(st, row, tab, qrow, j, tq2);
        if constexpr (f1 + 1 <= j1) chunk<R::m(f1 + 1 <= j1 ? f1 + 1 : 0), SYM, kPipe && !mode_welch(MODE), MODE, NG>(st, row, tab, qrow, f1 + 1, tq2);
        if constexpr (f1 + 2 <= j1) chunk<R::m(f1 + 2 <= j1 ? f1 + 2 : 0), SYM, kPipe && !mode_welch(MODE), MODE, NG>(st, row, tab, qrow, f1 + 2, tq2);
    }
}

// Read groups [J0, J1] of one window row, whole (experiment kGSplit): the two end groups of the window carry their static masks
template <int J0, int J1, bool SYM, int MODE, int NG, class LaneT>
__device__ __forceinline__ void sweep_groups(LaneT &st, const float *row, const float *tab, float *qrow, const WelchTab &tq2) {
    using R = Range<-kR, kR>;
    constexpr bool pipe = kPipe && !mode_welch(MODE);
    constexpr int f0 = J0 == 0 ? 1 : J0, f1 = J1 == kChunks - 1 ? kChunks - 2 : J1;
    static_assert(R::m(1) == R::kFull && R::m(kChunks - 2) == R::kFull, "only the end groups are cut by the window");
    if constexpr (J0 == 0) chunk<R::m(0), SYM, pipe, MODE, NG>(st, row, tab, qrow, 0, tq2);
#pragma unroll 1
    for (int j = f0; j <= f1; j++) chunk<R::kFull, SYM, pipe, MODE, NG>(st, row, tab, qrow, j, tq2);
    if constexpr (J1 == kChunks - 1) chunk<R::m(kChunks - 1), SYM, pipe, MODE, NG>(st, row, tab, qrow, kChunks - 1, tq2);
}

// The window columns of one window row, split between the two waves of a row at dx = kSplit: wave half 0 sweeps
// dx in [-20, kSplit], half 1 dx in [kSplit + 1, 20], every pair feeding both its pixels.  kSplit = 0: the middle.
// (The per-wave clocks -- tools/experiments/stamps_sym.py -- show the older wave of every SIMD pair finishing its
// half well before the younger one and idling at the barrier, which suggests giving it more columns; measured, any
// uneven split is slower: 1.43 ms at kSplit = 0; 1.46 at 4 and 1.48 at -4, which like 0 cut only one read group;
// 1.56 at 7 and 1.75 at 11, which cut two.  The SIMD is busy either way.)
// dy = 0: the pairs inside a row are the taps dx >= 1 (the accumulator row is the wave's own row); the pixel's own
// tap dx = 0 feeds the p side only.
// RT (runtime radius r < 20): the same staging geometry (20 halo columns) and the same split of the window at dx = 0; the
// read groups a half sweeps are groups [j_lo, 4] + the cut group 5 (half 0) and the cut group 5 + groups [6, j_hi] (half 1),
// where j_lo / j_hi are the outermost groups that hold a tap with |dx| <= r; taps of those groups beyond r carry a
// spatial exponent of -inf in the table (weight 0).
template <int HF, int MODE, int NG, bool RT, bool G7, class LaneT>
__device__ __forceinline__ void eval_half_row(LaneT &st, const float *row, const float *tab, float *qrow, bool dy0, int j_lo, int j_hi, const WelchTab &tq2) {
    constexpr bool kPipe = sym::kPipe && !mode_welch(MODE);
    if constexpr (RT) {
        static_assert(kSplit == 0, "the runtime-radius build splits the window in the middle");
        // From nine read groups up (r >= 16), in every build (ONE rule for all of them: see gsplit_of): whole read groups
        // [j_lo, g) to half 0, [g, j_hi] to half 1, g = j_lo + 9/14 of the groups; taps beyond the radius carry -inf in the table, so
        // no group needs a mask.  1080p, r = 19: 1.436 ms with the middle split and the housekeeping first, 1.377 with it last, 1.322
        // with this share on top (a 6/11 share: 1.372); Welch r = 20 3.48 -> 3.39 (pooled 3.47 -> 3.30).  Below nine groups the middle
        // split is as good or better (r = 10 0.58 | 0.60 ms, Welch r = 6 0.70 | 0.72; profiles/r06_rt.log, r06_modes_product*.log).
        if (kGroupSplitRT && j_hi - j_lo + 1 >= 9) {
            const int g = j_lo + ((j_hi - j_lo + 1) * 9 + 7) / 14;
            if (dy0) {
                if constexpr (HF == 0) {
                    sweep_range<0, 0, false, MODE, NG>(st, row, tab, qrow, tq2);
                } else {
                    chunk<Range<1, kR>::m(kMid), true, kPipe, MODE, NG>(st, row, tab, qrow, kMid, tq2);
#pragma unroll 1
                    for (int j = kMid + 1; j <= j_hi; j++) chunk<0xFFFFu, true, kPipe, MODE, NG>(st, row, tab, qrow, j, tq2);
                }
            } else if constexpr (HF == 0) {
#pragma unroll 1
                for (int j = j_lo; j < g; j++) chunk<0xFFFFu, true, kPipe, MODE, NG>(st, row, tab, qrow, j, tq2);
            } else {
#pragma unroll 1
                for (int j = g; j <= j_hi; j++) chunk<0xFFFFu, true, kPipe, MODE, NG>(st, row, tab, qrow, j, tq2);
            }
            return;
        }
        if constexpr (HF == 0) {
            if (dy0) {
                sweep_range<0, 0, false, MODE, NG>(st, row, tab, qrow, tq2);
            } else {
#pragma unroll 1
                for (int j = j_lo; j < kMid; j++) chunk<0xFFFFu, true, kPipe, MODE, NG>(st, row, tab, qrow, j, tq2);
                chunk<Range<-kR, 0>::m(kMid), true, kPipe, MODE, NG>(st, row, tab, qrow, kMid, tq2);
            }
        } else {
            chunk<Range<1, kR>::m(kMid), true, kPipe, MODE, NG>(st, row, tab, qrow, kMid, tq2);
#pragma unroll 1
            for (int j = kMid + 1; j <= j_hi; j++) chunk<0xFFFFu, true, kPipe, MODE, NG>(st, row, tab, qrow, j, tq2);
        }
        return;
    }
    constexpr int kGSplit = gsplit_of(MODE, NG, G7);
    if constexpr (kGSplit > 0) {   // the split at a read-group boundary (dy = 0: the pairs inside the row stay with half 1)
        static_assert(kSplit == 0 && kGSplit < kChunks - 1, "one split at a time");
        if (dy0) {
            if constexpr (HF == 0) sweep_range<0, 0, false, MODE, NG>(st, row, tab, qrow, tq2);
            else sweep_range<1, kR, true, MODE, NG>(st, row, tab, qrow, tq2);
        } else {
            if constexpr (HF == 0) sweep_groups<0, kGSplit - 1, true, MODE, NG>(st, row, tab, qrow, tq2);
            else sweep_groups<kGSplit, kChunks - 1, true, MODE, NG>(st, row, tab, qrow, tq2);
        }
        return;
    }
    if constexpr (HF == 0) {
        if (dy0) {
            sweep_range<0, 0, false, MODE, NG>(st, row, tab, qrow, tq2);
            if constexpr (kSplit >= 1) sweep_range<1, kSplit, true, MODE, NG>(st, row, tab, qrow, tq2);
        } else {
            sweep_range<-kR, kSplit, true, MODE, NG>(st, row, tab, qrow, tq2);
        }
    } else {
        if constexpr (kSplit >= 1) {
            sweep_range<kSplit + 1, kR, true, MODE, NG>(st, row, tab, qrow, tq2);
        } else {
            if (dy0) sweep_range<1, kR, true, MODE, NG>(st, row, tab, qrow, tq2);
            else sweep_range<kSplit + 1, kR, true, MODE, NG>(st, row, tab, qrow, tq2);
        }
    }
}

// Where a launch's features come from.  NG = 6: the first two G-buffers of the argument list (RGB, factor k0 / k1; an
// absent one has factor 0 and is never read).  NG = 8: up to two RGB and up to two 1-channel G-buffers in any order of the
// argument list, sorted into slots by the host (FilterArgs::SymGeom::rgb / sc).
struct Feat {
    const float *g0, *g1, *s0, *s1;
    float k0, k1, k2, k3;
};
template <int NG>
__device__ __forceinline__ Feat features_of(const FilterArgs &a) {
    if constexpr (NG == 6) return Feat{a.g[0].data, a.g[1].data, nullptr, nullptr, a.gscale0, a.gscale1, 0.f, 0.f};
    else return Feat{a.sym.rgb[0], a.sym.rgb[1], a.sym.sc[0], a.sym.sc[1], a.sym.rgb_scale[0], a.sym.rgb_scale[1], a.sym.sc_scale[0], a.sym.sc_scale[1]};
}
struct Staged {   // one staged pixel: the 15 values of the common layout + the two 1-channel features (NG = 8)
    StagedPixel p;
    float s0, s1;
    float nm1, nm1b;   // Welch: n - 1 as the oracle forms it, (float)n - 1.f; nm1b: the second buffer's (filter<float>)
};

// Welch builds: the sample count of pixel p -- its own image, or channel 15 of a 16-channel block + halo image;
// filter<float> (two buffers per launch): n0 / n1 = the two buffers' counts, else both the pixel's
__device__ __forceinline__ void sample_counts(const FilterArgs &a, long long p, int &n0, int &n1) {
    if (a.sym.pair) {
        n0 = a.f_n[0][p];
        n1 = a.f_active > 1 ? a.f_n[1][p] : n0;
    } else {
        n0 = n1 = a.packed ? __float_as_int(a.packed[p * a.packed_ch + (a.packed_ch == 18 ? 17 : 15)]) : a.n[p];
    }
}
// pixel (x, yrow) of the input images (an absent G-buffer has factor 0 and is not read)
template <int NG>
__device__ __forceinline__ Staged load_px(const FilterArgs &a, const Feat &F, int x, int yrow) {
    Staged r;
    StagedPixel &s = r.p;
    r.s0 = r.s1 = 0.f;
    r.nm1 = r.nm1b = 1.f;
    s.valid = x >= 0 && x < a.width && yrow >= 0 && yrow < a.height;
    if (s.valid) {
        const long long q = (long long)yrow * a.width + x;
        if (a.dof == STATMC_DOF_WELCH && !a.packed) {
            int n0, n1;
            sample_counts(a, q, n0, n1);
            r.nm1 = (float)n0 - 1.f;
            r.nm1b = (float)n1 - 1.f;
        }
        if (a.packed) {
            const float *pf = a.packed + q * a.packed_ch;
            const f3 *px = reinterpret_cast<const f3 *>(pf);
            s.mc = px[0]; s.d = px[1]; s.col = px[2]; s.g0 = px[3]; s.g1 = px[4];
            if (a.packed_ch == 16) r.nm1 = r.nm1b = (float)__float_as_int(pf[15]) - 1.f;   // Welch builds: the sample count's bits
            if constexpr (NG == 8) {
                if (a.packed_ch >= 17) { r.s0 = pf[15]; r.s1 = pf[16]; }
                if (a.packed_ch == 18) r.nm1 = r.nm1b = (float)__float_as_int(pf[17]) - 1.f;
            }
            return r;
        }
        s.mc = reinterpret_cast<const f3 *>(a.mean_corr)[q];
        s.d = reinterpret_cast<const f3 *>(a.disc)[q];
        s.col = reinterpret_cast<const f3 *>(a.colour)[q];
        s.g0 = F.k0 != 0.f ? reinterpret_cast<const f3 *>(F.g0)[q] : f3{0.f, 0.f, 0.f};
        s.g1 = F.k1 != 0.f ? reinterpret_cast<const f3 *>(F.g1)[q] : f3{0.f, 0.f, 0.f};
        if constexpr (NG == 8) {
            r.s0 = F.k2 != 0.f ? F.s0[q] : 0.f;
            r.s1 = F.k3 != 0.f ? F.s1[q] : 0.f;
        }
    }
    return r;
}

// stage column i of a row: inputs (NG = 6: exactly as the one-sided kernel stages them), accumulators cleared
// W (Welch): + the three planes E = v * v / (n - 1), the per-pixel addend of the Welch-Satterthwaite denominator, formed
// as the oracle forms it (pair_member: Dp * Dp / ((float)n[p] - 1.f))
template <int NG, bool W = false>
__device__ __forceinline__ void stage_store(float *slot, int i, const Staged &r, const Feat &F, bool rgb) {
    const StagedPixel &s = r.p;
    if constexpr (NG == 6) {
        store_pixel(slot, kP, i, s, F.k0, F.k1, rgb);
    } else {
        const bool v = s.valid && features_finite(s.g0, s.g1, r.s0, r.s1);
        const Validity ok = pixel_validity(s.mc, s.d, s.col, v, rgb);
        const f3 mc = canonical_mean(s.mc, ok);
        float *p = slot + i;
        p[0 * kP] = v ? s.g0.x * F.k0 : 0.f; p[1 * kP] = v ? s.g0.y * F.k0 : 0.f; p[2 * kP] = v ? s.g0.z * F.k0 : 0.f;
        p[3 * kP] = v ? s.g1.x * F.k1 : 0.f; p[4 * kP] = v ? s.g1.y * F.k1 : 0.f; p[5 * kP] = v ? s.g1.z * F.k1 : 0.f;
        p[6 * kP] = v ? r.s0 * F.k2 : 0.f;   p[7 * kP] = v ? r.s1 * F.k3 : 0.f;
        p[8 * kP] = mc.x; p[9 * kP] = mc.y; p[10 * kP] = mc.z;
        p[11 * kP] = ok.x ? -s.d.x : 0.f; p[12 * kP] = ok.y ? -s.d.y : 0.f; p[13 * kP] = ok.z ? -s.d.z : 0.f;
        p[14 * kP] = ok.x ? s.col.x : 0.f; p[15 * kP] = ok.y ? s.col.y : 0.f; p[16 * kP] = ok.z ? s.col.z : 0.f;
    }
    if constexpr (W && NG == 8) {
        slot[Planes<NG, W>::cE * kP + i] = r.nm1;
        if (!rgb) slot[(Planes<NG, W>::cMC + 2) * kP + i] = r.nm1b;   // (filter<float>: the third channel's planes are not read)
    } else if constexpr (W) {
        const Validity ok = pixel_validity(s.mc, s.d, s.col, s.valid && features_finite(s.g0, s.g1), rgb);
        float *p = slot + Planes<NG, W>::cE * kP + i;
        p[0 * kP] = ok.x ? s.d.x * s.d.x / r.nm1 : 0.f;
        p[1 * kP] = ok.y ? s.d.y * s.d.y / (rgb ? r.nm1 : r.nm1b) : 0.f;
        p[2 * kP] = ok.z ? s.d.z * s.d.z / r.nm1 : 0.f;
    }
#pragma unroll
    for (int v = 0; v < kQ; v++) slot[(Planes<NG, W>::kIn + v) * kP + i] = 0.f;
}

// the accumulators of row `rel` (tile-relative) leave the ring: copy A + copy B -> patch
template <int NG, bool W = false>
__device__ __forceinline__ void flush_q(const float *slot, int i, float4 *patch_q_row) {
    const float *q = slot + Planes<NG, W>::kIn * kP + i;
    patch_q_row[i] = make_float4(q[0 * kP] + q[4 * kP], q[1 * kP] + q[5 * kP], q[2 * kP] + q[6 * kP], q[3 * kP] + q[7 * kP]);
}

// Issue the LDS-DMA transfers of image row `yrow`, columns [xw0, xw0 + ncols) -> the wave's raw area.  Nothing
// passes through registers and nothing waits: the data is used one step later.  Pieces never straddle the image
// border (xw0 and the image width are multiples of 4 pixels = 3 or 15 pieces, or 1 piece of a 1-channel image); pieces
// outside the image are skipped and their stale bytes are never looked at (validity is decided from coordinates).
// Raw area of a wave: the five RGB images' pieces one after the other (ncols x 3 floats each), then (NG = 8) the two
// 1-channel images' (ncols floats each).
template <int NG>
__device__ __forceinline__ void dma_row(const FilterArgs &a, const Feat &F, float *raw_w, int lane, int xw0, int yrow, int ncols) {
    if (yrow < 0 || yrow >= a.height) return;
    const int per_img = ncols * 3 / 4, per_sc = ncols / 4;
    const int total = a.packed ? ncols * a.packed_ch / 4 : 5 * per_img + (NG == 8 ? 2 * per_sc : 0);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int idx = 64 * j + lane;
        if (idx < total) {
            const float *src;
            bool inside;
            if (a.packed) {
                const long long f = (long long)xw0 * a.packed_ch + 4 * idx;   // float offset inside the AoS row
                inside = f >= 0 && f + 4 <= (long long)a.width * a.packed_ch;
                src = a.packed + (long long)yrow * a.width * a.packed_ch + f;
            } else if (NG == 8 && idx >= 5 * per_img) {
                const int i2 = idx - 5 * per_img, m = i2 / per_sc, p = i2 - m * per_sc;
                const float *img = m == 0 ? F.s0 : F.s1;
                const long long f = (long long)xw0 + 4 * p;
                inside = f >= 0 && f + 4 <= (long long)a.width && img != nullptr;
                src = img + (long long)yrow * a.width + f;
            } else {
                const int m = idx / per_img, p = idx - m * per_img;
                const float *img = m == 0 ? a.mean_corr : m == 1 ? a.disc : m == 2 ? a.colour : m == 3 ? F.g0 : F.g1;
                const long long f = (long long)xw0 * 3 + 4 * p;
                inside = f >= 0 && f + 4 <= (long long)a.width * 3 && img != nullptr;
                src = img + (long long)yrow * a.width * 3 + f;
            }
            if (inside)
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(raw_w + 256 * j), 16, 0, 0);
        }
    }
}

// column c of the wave's raw area -> the staged pixel (image order in the raw area: mean, discriminator, colour, g0, g1[, s0, s1])
template <int NG>
__device__ __forceinline__ Staged raw_pixel(const FilterArgs &a, const Feat &F, const float *raw_w, int c, int ncols, int x, int yrow) {
    Staged out;
    StagedPixel &s = out.p;
    s.valid = x >= 0 && x < a.width && yrow >= 0 && yrow < a.height;
    const int im = a.packed ? 3 : ncols * 3, px = a.packed ? a.packed_ch : 3;   // float stride between images / pixels
    const float *r = raw_w + c * px;
    s.mc = f3{r[0], r[1], r[2]};
    s.d = f3{r[im], r[im + 1], r[im + 2]};
    s.col = f3{r[2 * im], r[2 * im + 1], r[2 * im + 2]};
    s.g0 = F.k0 != 0.f ? f3{r[3 * im], r[3 * im + 1], r[3 * im + 2]} : f3{0.f, 0.f, 0.f};
    s.g1 = F.k1 != 0.f ? f3{r[4 * im], r[4 * im + 1], r[4 * im + 2]} : f3{0.f, 0.f, 0.f};
    out.s0 = out.s1 = 0.f;
    if constexpr (NG == 8) {
        if (a.packed) {   // (a 15-channel packed image on this build: no 1-channel features, their factors are 0)
            out.s0 = F.k2 != 0.f ? r[15] : 0.f;
            out.s1 = F.k3 != 0.f ? r[16] : 0.f;
        } else {
            out.s0 = F.k2 != 0.f ? raw_w[15 * ncols + c] : 0.f;
            out.s1 = F.k3 != 0.f ? raw_w[16 * ncols + c] : 0.f;
        }
    }
    return out;
}

// DMA = rows are staged by LDS-DMA (two RGB G-buffers or the packed image, 16-byte aligned images whose width and
// film x-origin are multiples of 4 pixels); otherwise through registers (any layout the one-sided kernel accepts).
// PAIR = filter<float>, two 1-channel buffers per launch: the (x, y) channels of the three statistics / colour images
// hold buffer 0 and buffer 1 (pack_pair_kernel), the third channel is empty; the buffers share the range weight, gate
// and normalise separately.
// G7: the call's G-buffers are exactly two RGB images (see gsplit_of): the window is shared after read group 7, not 6
template <bool DMA, int MODE, int NG, bool RT, bool G7 = false>
__global__ __launch_bounds__(kThreads, 2) void window_filter_sym(FilterArgs a) {
    static_assert(!G7 || (!RT && NG == 6 && (MODE == kModeRgb || MODE == kModeJoint)), "G7 builds: r = 20, six planes, the symmetric gate on one RGB buffer");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // Welch (band build): the quantile band sits at LDS address 0 -- a look-up's byte offset is its address -- and the
    // ring behind it
    float *const lds = smem + Planes<NG, mode_welch(MODE)>::kBandTotal;
    constexpr bool PAIR = mode_pair(MODE);
    constexpr bool W = mode_welch(MODE);
    static_assert(!W || (!DMA && RT), "the Welch modes exist in one build per feature-plane count: register staging, runtime radius");
    constexpr int kSlotFloats = Planes<NG, W>::kSlotFloats, kIn = Planes<NG, W>::kIn;
    const Feat F = features_of<NG>(a);
    WelchTab tq2{nullptr, nullptr, 0u, ~0u};
    // XCD-aware work mapping (as in the one-sided kernel): each XCD walks a contiguous range of items
    // With a tail split the items of the n_parts-tiles (the long ones) go first in every XCD's range and the short
    // items of the parts_hi-tiles fill the end of the launch.
    const int n_items = gridDim.x, b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int n_long = a.sym.parts_hi ? a.sym.n_lo_items : n_items, n_short = n_items - n_long;
    int u;
    {
        const int ql = n_long >> 3, rl = n_long & 7;
        const int long_here = ql + (xcd < rl ? 1 : 0), long_start = xcd < rl ? xcd * (ql + 1) : rl * (ql + 1) + (xcd - rl) * ql;
        if (idx < long_here) {
            u = long_start + idx;
        } else {
            // short items of this XCD: what its share of the grid leaves after its long items; their start is the sum over the XCDs before it
            const int qt = n_items >> 3, rt = n_items & 7;
            int short_start = 0;
            for (int x = 0; x < xcd; x++) short_start += qt + (x < rt ? 1 : 0) - (ql + (x < rl ? 1 : 0));
            u = n_long + short_start + (idx - long_here);
            if (u >= n_items) return;   // (cannot happen: the shares add up to n_short; a guard against a grid of another size)
        }
        (void)n_short;
    }

    unsigned long long t_start = 0, t_swept = 0, c_pro = 0, s_hk = 0, s_ev = 0, s_bar = 0;
    unsigned long long rt_start = 0;     // (the 100 MHz counter every XCD shares: when, within the launch, the item started)
    if constexpr (kStamps) { t_start = __builtin_amdgcn_s_memtime(); rt_start = __builtin_amdgcn_s_memrealtime(); }
    int part, tile, n_parts;
    if (!a.sym.parts_hi || u < a.sym.n_lo_items) {
        n_parts = a.n_parts;
        part = u % n_parts;
        tile = u / n_parts;
    } else {
        n_parts = a.sym.parts_hi;
        part = (u - a.sym.n_lo_items) % n_parts;
        tile = a.sym.n_lo_tiles + (u - a.sym.n_lo_items) / n_parts;
    }
    if constexpr (mode_welch_far(MODE)) {
        if (a.sym.redo[u] == 0) return;   // (block-uniform) the band build served every pair of this item
    }
    const int x0 = kW * (a.sym.tx0 + tile % a.sym.ntx) - a.sym.fx0;     // local coordinates of the tile
    const int y0 = kRows * (a.sym.ty0 + tile / a.sym.ntx) - a.sym.fy0;
    const int steps = RT ? a.sym.steps : kSteps;
    const int s_a = step_lo(part, n_parts, steps);
    const int s_b = min(step_lo(part + 1, n_parts, steps), a.height - y0);  // window rows below the image are not swept
    // spatial exponents: [dy + 20][..] of the (2r+1)-row table (r = 20) / [dy][..] of the runtime-radius table
    const float *stab = RT ? a.sym.tab_rt : a.spatial_tab + kR * kTabW;
    const int j_lo = RT ? (kR - a.radius) / 4 : 0, j_hi = RT ? min(kChunks - 1, (a.radius + kR + 3) / 4) : kChunks - 1;
    const int q_first = s_a;   // accumulator rows the item flushes: tile-relative rows s_a .. s_b+2
    float4 *patch = a.sym.patch + (long long)u * a.sym.item_stride4;
    float4 *patch_q = patch + kPatchP;
    float *tab_lds = lds + kSlots * kSlotFloats;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int lane32 = lane & 31;                       // 4-pixel column group of the lane inside its row
    const int trow = (wave & 3) + 4 * (lane >> 5);      // lower / upper half of the wave: rows t and t + 4
    const int half = wave >> 2;
    constexpr int tw = kTabW;
    if constexpr (kPrio == 1) { if (half == 1) __builtin_amdgcn_s_setprio(2); }

    // ---- the lane's own 4 pixels (loads clamped into the image; outside it the pixel takes no part)
    Lane<NG, W> st;
    const int py = y0 + trow;
    const int pyc = min(max(py, 0), a.height - 1);
#pragma unroll
    for (int k = 0; k < kPx; k++) {
        const int px = x0 + kPx * lane32 + k;
        const int pxc = min(max(px, 0), a.width - 1);
        const long long p = (long long)pyc * a.width + pxc;
        f3 mc, d, g0, g1, col;
        float sc0 = 0.f, sc1 = 0.f;
        if (a.packed) {
            const float *pf = a.packed + p * a.packed_ch;
            const f3 *q = reinterpret_cast<const f3 *>(pf);
            mc = q[0]; d = q[1]; col = q[2]; g0 = q[3]; g1 = q[4];
            if constexpr (NG == 8) {
                if (a.packed_ch >= 17) { sc0 = pf[15]; sc1 = pf[16]; }
            }
        } else {
            mc = reinterpret_cast<const f3 *>(a.mean_corr)[p];
            d = reinterpret_cast<const f3 *>(a.disc)[p];
            col = reinterpret_cast<const f3 *>(a.colour)[p];
            g0 = F.k0 != 0.f ? reinterpret_cast<const f3 *>(F.g0)[p] : f3{0.f, 0.f, 0.f};
            g1 = F.k1 != 0.f ? reinterpret_cast<const f3 *>(F.g1)[p] : f3{0.f, 0.f, 0.f};
            if constexpr (NG == 8) {
                sc0 = F.k2 != 0.f ? F.s0[p] : 0.f;
                sc1 = F.k3 != 0.f ? F.s1[p] : 0.f;
            }
        }
        bool inside = px >= 0 && px < a.width && py >= 0 && py < a.height;
        if (!features_finite(g0, g1, sc0, sc1)) {   // spec v2.1: such a pixel takes no part; its features enter no exponent as NaN
            inside = false;
            g0 = g1 = f3{0.f, 0.f, 0.f};
            sc0 = sc1 = 0.f;
        }
        const Validity ok = pixel_validity(mc, d, col, inside, !PAIR);   // per pixel (RGB) / per buffer (PAIR)
        mc = canonical_mean(mc, ok);
        st.pg[k][0] = v2f{g0.x * F.k0, g0.y * F.k0};
        st.pg[k][1] = v2f{g0.z * F.k0, g1.x * F.k1};
        st.pg[k][2] = v2f{g1.y * F.k1, g1.z * F.k1};
        if constexpr (NG == 8) st.pg[k][3] = v2f{sc0 * F.k2, sc1 * F.k3};
        st.ms[k][0] = v2f{mc.x, d.x};
        st.ms[k][1] = v2f{mc.y, d.y};
        st.ms[k][2] = v2f{mc.z, d.z};
        // a pixel that takes no part adds nothing to its taps: weight 0, colour 0 (0 * NaN would poison them)
        st.pc[k][0] = v2f{ok.x ? col.x : 0.f, ok.y ? col.y : 0.f};
        st.pc[k][1] = v2f{ok.z ? col.z : 0.f, 0.f};
        if constexpr (W) {
            int n0, n1;
            sample_counts(a, p, n0, n1);
            const float nm1 = (float)n0 - 1.f, nm1b = (float)n1 - 1.f;
            st.pe[k][0] = v2f{d.x * d.x / nm1, d.y * d.y / nm1b};
            st.pe[k][1] = v2f{d.z * d.z / nm1, 0.f};
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) st.acc[k][ch] = v2f{0.f, 0.f};
        st.sw[k] = v2f{0.f, 0.f};
    }
    if constexpr (W) st.far = 0u;

    // wave-local staging geometry (DMA): this wave's columns of every staged row
    constexpr bool kHkAtEnd = hk_at_end(MODE, NG);
    float *raw_w = tab_lds + Planes<NG, W>::kTabFloats + Planes<NG, W>::raw_off(wave);
    const int wcol0 = wave_col0(wave);                                 // first staged column (0..167) of the wave
    const int ncols = wave_cols(wave);                                 // 44, 44, 40, 40, then none
    if (s_a < s_b) {
        if constexpr (W && mode_welch_far(MODE)) tq2 = WelchTab{a.tq2, nullptr, 0u, ~0u};
        if constexpr (W && !mode_welch_far(MODE)) {
            // ---- the band of the squared-quantile table this item's pairs can ask for: least sample count over the rows the
            // item touches (its own rows and the staged ones, rel 0 .. s_b+6) x the staged columns, then kWelchBand entries
            // from (least n) - 2 into LDS.  (The two below the mathematical minimum min(n_p, n_q) - 1 take the computed
            // quotient's last-bit errors; a computed nu below even that -- possible only where E_p + E_q is denormal --
            // wraps around in the unsigned offset and reads the band's last entry, like a dof beyond the band: gate_weight.)
            // The band is addressed from LDS address 0 (lds_at0): `smem` is the kernel's only LDS allocation, which the
            // compiler places at 0; the diagnostic builds (STATMC_SYM_STAMPS) check it.
            float *band = smem;
            int *nrange = reinterpret_cast<int *>(band + kWelchBand);   // {least n, item flag}
            if (threadIdx.x == 0) { nrange[0] = 0x7fffffff; nrange[1] = 0; }
            __syncthreads();
            {
                const int ya = max(y0, 0), yb = min(y0 + s_b + kRows - 2, a.height - 1);
                const int xa = max(x0 - kR, 0), xb = min(x0 + kW + kR - 1, a.width - 1);
                const int wc = xb - xa + 1, cnt = wc * (yb - ya + 1);
                int mn = 0x7fffffff;
                for (int idx2 = threadIdx.x; idx2 < cnt; idx2 += kThreads) {
                    const int yy = idx2 / wc, xx = idx2 - yy * wc;
                    int n0, n1;
                    sample_counts(a, (long long)(ya + yy) * a.width + xa + xx, n0, n1);
                    mn = min(mn, min(n0, n1));
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
                if (lane == 0) atomicMin(&nrange[0], mn);
            }
            __syncthreads();
            const int n_lo = __builtin_amdgcn_readfirstlane(nrange[0]);
            const unsigned lo = (unsigned)min(max(n_lo - 2, 0), 4096);
            for (int j = threadIdx.x; j < kWelchBand; j += kThreads) band[j] = a.tq2[min(lo + (unsigned)j, 4096u)];
            tq2 = WelchTab{a.tq2, band, 0u - 4u * lo, lo + (unsigned)kWelchBand - 1u >= 4096u ? ~0u : lo + (unsigned)kWelchBand - 1u};
            // (the barrier after the prologue's staging comes before the first look-up)
        }
        if constexpr (DMA) {   // the first row the sweep will need beyond the prologue: on its way during the prologue
            if (s_a + 1 < s_b && !(kAblate & 2)) dma_row<NG>(a, F, raw_w, lane, x0 - kR + wcol0, y0 + s_a + kRows, ncols);
        }
        // ---- prologue: rows rel = s_a .. s_a+7 (image rows y0 + rel) into slots rel % 9.  All of a thread's fetches
        // are issued before the first is staged: one memory latency per item instead of three.
        constexpr int kProIters = (kRows * kP + kThreads - 1) / kThreads;
        Staged pro[kProIters];
#pragma unroll
        for (int it = 0; it < kProIters; it++) {
            const int idx2 = (int)threadIdx.x + it * kThreads;
            pro[it].p.valid = false;
            if (idx2 < kRows * kP) {
                const int rr = idx2 / kP, i = idx2 - rr * kP;
                pro[it] = load_px<NG>(a, F, x0 - kR + i, y0 + s_a + rr);
            }
        }
#pragma unroll
        for (int it = 0; it < kProIters; it++) {
            const int idx2 = (int)threadIdx.x + it * kThreads;
            if (idx2 < kRows * kP) {
                const int rr = idx2 / kP, i = idx2 - rr * kP;
                stage_store<NG, W>(lds + ((s_a + rr) % kSlots) * kSlotFloats, i, pro[it], F, !PAIR);
            }
        }
        if ((int)threadIdx.x < tw) {
            const float *t = stab + s_a * tw + threadIdx.x;
            *reinterpret_cast<v2f *>(tab_lds + 2 * threadIdx.x) = v2f{t[0], (int)threadIdx.x + 1 < tw ? t[1] : 0.f};
        }
        __syncthreads();

        // ---- sweep
        unsigned long long tk0 = 0, c_hk = 0, c_ev = 0, c_bar = 0;
        if constexpr (kStamps) c_pro = __builtin_amdgcn_s_memtime() - t_start;
        for (int s = s_a; s < s_b; s++) {
            if constexpr (kStamps) tk0 = __builtin_amdgcn_s_memtime();
            const int i = DMA ? wcol0 + lane : (int)threadIdx.x;             // the staged column this thread looks after
            const bool mine = DMA ? lane < ncols : i < kP;
            const bool stage = s + 1 < s_b && mine && !(kAblate & 2);
            Staged nxt;
            nxt.p.valid = false;
            if constexpr (!DMA) {
                if (stage) nxt = load_px<NG>(a, F, x0 - kR + i, y0 + s + kRows);
            }
            // Once per step every wave (a) hands the accumulators of the row that went dead at the last barrier to the
            // patch, (b) turns the row fetched during the last step into a ring slot (the dead row's), and (c) starts
            // the fetch of the row after it -- before its sweep or after it, per build (hk_at_end: measured, round 6; staggering
            // the two waves of a SIMD, the other wave of the pair, or both waves in turn were all slower: HISTORY.md 4.3 / 4.3d).
            auto housekeeping = [&]() {
                const int dead = s - 1;
                if (mine && dead >= q_first && y0 + dead >= 0 && !(kAblate & 4))
                    flush_q<NG, W>(lds + (dead % kSlots) * kSlotFloats, i, patch_q + (long long)(dead - q_first) * kP);
                if constexpr (DMA) {
                    if (s + 1 < s_b && !(kAblate & 2)) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's own transfers
                        if (stage) {
                            const Staged sp = raw_pixel<NG>(a, F, raw_w, lane, ncols, x0 - kR + i, y0 + s + kRows);
                            stage_store<NG, W>(lds + ((s + kRows) % kSlots) * kSlotFloats, i, sp, F, !PAIR);
                        }
                        if (s + 2 < s_b) {
                            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the raw area has been read
                            dma_row<NG>(a, F, raw_w, lane, x0 - kR + wcol0, y0 + s + 1 + kRows, ncols);
                        }
                    }
                }
            };
            if constexpr (kPrio == 2) { if (half == 0) __builtin_amdgcn_s_setprio(3); }
            if (!DMA || (!kHkAtEnd && half == 0)) housekeeping();
            if constexpr (kPrio == 2) { if (half == 0) __builtin_amdgcn_s_setprio(0); }
            if constexpr (kStamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_hk += t - tk0; tk0 = t; }
            const int ti = (int)threadIdx.x - (kThreads - 64);
            const bool tstage = s + 1 < s_b && ti >= 0 && ti < tw;
            v2f tnext = v2f{0.f, 0.f};
            if (tstage) {
                const float *t = stab + (s + 1) * tw + ti;
                tnext = v2f{t[0], ti + 1 < tw ? t[1] : 0.f};
            }

            float *slot = lds + ((s + trow) % kSlots) * kSlotFloats;   // per lane: the two halves of a wave differ
            const float *row = slot + kPx * lane32;
            float *qrow = slot + (kIn + 4 * half) * kP + kPx * lane32;
            const float *tab = tab_lds + ((s - s_a) & 1) * kTabPad;
            if (kAblate & 8) {
            } else if (half == 0) {
                eval_half_row<0, MODE, NG, RT, G7>(st, row, tab, qrow, s == 0, j_lo, j_hi, tq2);
            } else {
                eval_half_row<1, MODE, NG, RT, G7>(st, row, tab, qrow, s == 0, j_lo, j_hi, tq2);
            }

            if constexpr (kStamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_ev += t - tk0; tk0 = t; }
            if (DMA && kHkAtEnd && half == 0) housekeeping();
            if (tstage) *reinterpret_cast<v2f *>(tab_lds + ((s - s_a + 1) & 1) * kTabPad + 2 * ti) = tnext;
            if constexpr (!DMA) {
                if (stage) stage_store<NG, W>(lds + ((s + kRows) % kSlots) * kSlotFloats, i, nxt, F, !PAIR);
            }
            if (!(kAblate & 16)) __syncthreads();
            if constexpr (kStamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_bar += t - tk0; tk0 = t; }
        }
        if constexpr (kStamps) {
            t_swept = __builtin_amdgcn_s_memtime();
            s_hk = c_hk; s_ev = c_ev; s_bar = c_bar;
        }
        // ---- the rows still in the ring: rel = s_b-1 .. s_b+6
        for (int idx2 = threadIdx.x; idx2 < kRows * kP; idx2 += kThreads) {
            const int rr = idx2 / kP, i = idx2 - rr * kP;
            const int rel = s_b - 1 + rr;
            if (rel >= q_first && y0 + rel >= 0 && y0 + rel < a.height)
                flush_q<NG, W>(lds + (rel % kSlots) * kSlotFloats, i, patch_q + (long long)(rel - q_first) * kP);
        }
        if constexpr (W && !mode_welch_far(MODE)) {
            if (st.far >= tq2.oob) reinterpret_cast<int *>(smem + kWelchBand)[1] = 1;
        }
        __syncthreads();
    }
    if constexpr (W && !mode_welch_far(MODE)) {   // (an item without a sweep: nothing to do again)
        if (threadIdx.x == 0) a.sym.redo[u] = s_a < s_b ? reinterpret_cast<const int *>(smem + kWelchBand)[1] : 0;
    }

    // ---- p side: half 1 hands its sums to half 0 through LDS, half 0 writes the patch
    float *ex = lds;  // [trow][v][kW]
    if (half == 1) {
#pragma unroll
        for (int k = 0; k < kPx; k++) {
            float *e = ex + trow * 4 * kW + kPx * lane32 + k;
            e[0 * kW] = st.acc[k][0].x + st.acc[k][0].y;
            e[1 * kW] = st.acc[k][1].x + st.acc[k][1].y;
            e[2 * kW] = st.acc[k][2].x + st.acc[k][2].y;
            e[3 * kW] = st.sw[k].x + st.sw[k].y;
        }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < kPx; k++) {
            const float *e = ex + trow * 4 * kW + kPx * lane32 + k;
            patch[trow * kW + kPx * lane32 + k] =
                make_float4((st.acc[k][0].x + st.acc[k][0].y) + e[0 * kW], (st.acc[k][1].x + st.acc[k][1].y) + e[1 * kW],
                            (st.acc[k][2].x + st.acc[k][2].y) + e[2 * kW], (st.sw[k].x + st.sw[k].y) + e[3 * kW]);
        }
    }
#if STATMC_SYM_COUNT
    __syncthreads();
    if (lane == 0 && s_a < s_b)
        patch[a.sym.item_stride4 - 8 + wave] = make_float4((float)st.n_groups, (float)st.n_empty, (float)st.n_empty_halves, 0.f);
#endif
    if constexpr (kStamps) {
        // the item's last 16 float4 (columns of its last accumulator row: the results of a stamps build are wrong):
        // per wave (housekeeping, sweep, barrier clocks summed over the steps, steps) and (before the first step,
        // after the last step, whole item)
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (lane == 0 && s_a < s_b) {
            patch[a.sym.item_stride4 - 8 + wave] = make_float4((float)s_hk, (float)s_ev, (float)s_bar, (float)(s_b - s_a));
            patch[a.sym.item_stride4 - 16 + wave] = make_float4((float)c_pro, (float)(t_end - t_swept), (float)(t_end - t_start), (float)(rt_start & 0xFFFFFFull));
            unsigned xcc = 0, hw = 0;      // where the wave ran (tools/experiments/stamps_roi.py)
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            patch[a.sym.item_stride4 - 24 + wave] = make_float4(__uint_as_float(xcc), __uint_as_float(hw), __uint_as_float(blockIdx.x), 0.f);
        }
    }
}

__device__ __forceinline__ int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// Gathers the shares of every ROI pixel from the patches, in a fixed (film-anchored) order, and normalises.
__global__ __launch_bounds__(256) void combine_sym_kernel(FilterArgs a) {
    const int x = a.rx0 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.ry0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.rx1 || y >= a.ry1) return;
    const int X = x + a.sym.fx0, Y = y + a.sym.fy0;
    const int ty_own = floordiv(Y, kRows), tx_own = floordiv(X, kW);
    const int steps = a.sym.steps, r = a.radius;
    const int ty_lo = max(floordiv(Y - (steps + kRows - 2) + kRows - 1, kRows), a.sym.ty0);  // rows of tile row Ty: kRows Ty .. kRows Ty + steps + kRows - 2
    const int ty_hi = min(ty_own, a.sym.ty0 + a.sym.nty - 1);
    const int tx_lo = max(floordiv(X - r, kW), a.sym.tx0);              // tiles whose pixels lie within r columns of X: X - kW Tx in [-r, kW - 1 + r]
    const int tx_hi = min(floordiv(X + r, kW), a.sym.tx0 + a.sym.ntx - 1);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int Ty = ty_lo; Ty <= ty_hi; Ty++) {
        const int rel = Y - kRows * Ty;  // 0 .. kSteps + kRows - 2
        for (int Tx = tx_lo; Tx <= tx_hi; Tx++) {
            const int c = X - kW * Tx + kR;  // 0 .. kP - 1
            const bool hi = a.sym.parts_hi != 0 && Ty >= a.sym.split_ty;
            const int n_parts = hi ? a.sym.parts_hi : a.n_parts;
            const long long tile_local = (long long)(Ty - a.sym.ty0) * a.sym.ntx + (Tx - a.sym.tx0);   // lo tiles first: rows below split_ty
            const long long item0 = hi ? a.sym.n_lo_items + (tile_local - a.sym.n_lo_tiles) * n_parts : tile_local * n_parts;
            for (int k = 0; k < n_parts; k++) {
                const float4 *patch = a.sym.patch + (item0 + k) * a.sym.item_stride4;
                const int s_a = step_lo(k, n_parts, steps), s_b = step_lo(k + 1, n_parts, steps), q_first = s_a;
                if (Ty == ty_own && Tx == tx_own) {
                    const float4 v = patch[rel * kW + (X - kW * Tx)];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                if (rel >= q_first && rel < s_b + kRows - 1) {
                    const float4 v = patch[kPatchP + (long long)(rel - q_first) * kP + c];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
            }
        }
    }
    const long long p = (long long)y * a.width + x;
    if (a.sym.border_extra && (x < r || x >= a.width - r || y < r || y >= a.height - r)) {
        const float4 v = a.sym.border_extra[p];   // clamped border: the taps beyond the image
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    if (a.sym.pair) {   // (sum w0 c0, sum w1 c1, sum w0, sum w1) of two float buffers
        a.f_out[0][p] = t.z > 0.f ? t.x / t.z : a.f_colour[0][p];
        if (a.f_active > 1) a.f_out[1][p] = t.w > 0.f ? t.y / t.w : a.f_colour[1][p];
        return;
    }
    f3 o;
    if (t.w > 0.f) {
        o.x = t.x / t.w; o.y = t.y / t.w; o.z = t.z / t.w;
    } else {
        o = a.packed ? reinterpret_cast<const f3 *>(a.packed + p * a.packed_ch)[2] : reinterpret_cast<const f3 *>(a.colour)[p];
    }
    reinterpret_cast<f3 *>(a.out)[p] = o;
}

// filter<float>: the statistics and colours of two 1-channel buffers -> the (x, y) channels of three RGB-shaped
// images, so that rows stage exactly as for filter<float3>.  An absent second buffer gets a NaN mean: it takes no part.
__global__ __launch_bounds__(256) void pack_pair_kernel(FilterArgs a, float *mc3, float *d3, float *c3, long long p0, long long p1) {
    const long long p = p0 + (long long)blockIdx.x * 256 + threadIdx.x;   // pixels [p0, p1): the rows the launch's tiles stage
    if (p >= p1) return;
    const bool two = a.f_active > 1;
    reinterpret_cast<f3 *>(mc3)[p] = f3{a.f_mean_corr[0][p], two ? a.f_mean_corr[1][p] : __builtin_nanf(""), 0.f};
    reinterpret_cast<f3 *>(d3)[p] = f3{a.f_disc[0][p], two ? a.f_disc[1][p] : 0.f, 0.f};
    reinterpret_cast<f3 *>(c3)[p] = f3{a.f_colour[0][p], two ? a.f_colour[1][p] : 0.f, 0.f};
}

static int floordiv_h(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

}  // namespace sym

int sym_diagnostic_bits() { return STATMC_SYM_DIAGNOSTIC_BITS; }

// Tile range of a launch: every tile of the film grid that holds a pixel whose upper half-window reaches the
// ROI (rows ry0-20 .. ry1-1, columns rx0-20 .. rx1+19, clipped to the local image).
void sym_geometry(FilterArgs &a) {
    using namespace sym;
    const int r = a.radius;
    const int ex0 = std::max(0, a.rx0 - r), ex1 = std::min(a.width, a.rx1 + r);
    const int ey0 = std::max(0, a.ry0 - r), ey1 = a.ry1;
    a.sym.steps = r + 1;
    a.sym.tx0 = floordiv_h(ex0 + a.sym.fx0, kW);
    a.sym.ty0 = floordiv_h(ey0 + a.sym.fy0, kRows);
    a.sym.ntx = floordiv_h(ex1 - 1 + a.sym.fx0, kW) - a.sym.tx0 + 1;
    a.sym.nty = floordiv_h(ey1 - 1 + a.sym.fy0, kRows) - a.sym.ty0 + 1;
}
int sym_tiles(const FilterArgs &a) { return a.sym.ntx * a.sym.nty; }

// Parts per tile: the grid runs one workgroup per CU, so its makespan is ceil(items / CUs) rounds of the longest part,
// ceil(21 / parts) steps, plus what a further item costs: 0.95 steps before its first and after its last step
// (stamps_sym.py), 7 more accumulator rows to flush and to gather in the combine.  Fitted at 1.35 steps on 1080p runs
// with 1 .. 4 parts (1.46 / 1.61 / 1.62 / 1.92 ms); the same model orders the parts of a 1920 x 135 / 270 / 540 block
// (tools/experiments/block_parts.py: 3 parts best for all three).
int sym_choose_parts(int tiles, int n_cus, int steps) {
    int best = 1;
    double best_cost = 1e30;
    for (int k = 1; k <= 8 && k <= steps; k++) {
        const double rounds = (double)(((long long)tiles * k + n_cus - 1) / n_cus);
        const double cost = rounds * ((double)((steps + k - 1) / k) + 1.35);
        if (cost < best_cost * 0.98) {
            best_cost = cost;
            best = k;
        }
    }
    return best;
}
// Work items of a launch (sym geometry and split applied)
long long sym_items(const FilterArgs &a) {
    if (!a.sym.parts_hi) return (long long)sym_tiles(a) * a.n_parts;
    return (long long)a.sym.n_lo_items + (long long)(sym_tiles(a) - a.sym.n_lo_tiles) * a.sym.parts_hi;
}
void sym_apply_split(FilterArgs &a) {
    if (!a.sym.parts_hi) {
        a.sym.n_lo_tiles = sym_tiles(a);
        a.sym.n_lo_items = a.sym.n_lo_tiles * a.n_parts;
        return;
    }
    const int lo_rows = std::max(0, std::min(a.sym.ty0 + a.sym.nty, a.sym.split_ty) - a.sym.ty0);
    a.sym.n_lo_tiles = lo_rows * a.sym.ntx;
    a.sym.n_lo_items = a.sym.n_lo_tiles * a.n_parts;
}
// (n_parts: the smaller part count of the launch -- its items have the most accumulator rows and set the item stride)
size_t sym_patch_floats(const FilterArgs &a, int n_parts) {
    return (size_t)sym_items(a) * (sym::kPatchP + (size_t)sym::q_rows_max(n_parts, a.radius + 1) * sym::kP) * 4;
}

// Makespan of a launch in step units: one workgroup per CU, items handed out in order to whichever CU is free.
static double sym_makespan(long long n_long, double c_long, long long n_short, double c_short, int n_cus) {
    std::vector<double> cu((size_t)n_cus, 0.0);   // a binary heap of finish times
    auto run = [&](long long n, double c) {
        for (long long i = 0; i < n; i++) {
            std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
            cu.back() += c;
            std::push_heap(cu.begin(), cu.end(), std::greater<double>());
        }
    };
    run(n_long, c_long);
    run(n_short, c_short);
    return *std::max_element(cu.begin(), cu.end());
}

// Parts for the whole local image: the uniform choice of sym_choose_parts, or -- when that leaves the last round of
// workgroups mostly empty -- the same with the last tile rows swept by more parts (a "tail split"): their short items fill
// the end of the launch.  1280 x 720 (900 tiles): 4 rounds of 21 steps -> 3 rounds + half-length items.  Cost of an item as in
// sym_choose_parts: ceil(steps / parts) + 1.35 steps.  Results are cached per (tile grid, CUs, steps): the search simulates
// the dispatch.  A pinned split (statmc_set_filter_split) is uniform: no tail.
void sym_choose_split(FilterArgs &w, int n_cus) {
    w.sym.parts_hi = 0;
    w.sym.split_ty = 0;
    if (w.force_parts > 0) {
        w.n_parts = sym_filter_parts(w, n_cus);
        return;
    }
    struct Choice { int lo, hi, tail_rows; };
    static std::mutex mu;
    static std::map<std::array<int, 4>, Choice> cache;
    const int steps = w.radius + 1, ntx = w.sym.ntx, nty = w.sym.nty;
    const std::array<int, 4> key = {ntx, nty, n_cus, steps};
    Choice c{1, 0, 0};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) {
            c = it->second;
        } else {
            const long long tiles = (long long)ntx * nty;
            auto cost = [&](int k) { return (double)((steps + k - 1) / k) + 1.35; };
            c.lo = sym_choose_parts((int)tiles, n_cus, steps);
            double best = sym_makespan(tiles * c.lo, cost(c.lo), 0, 0.0, n_cus);
            for (int lo = 1; lo <= 3 && lo <= steps; lo++)
                for (int hi = 2 * lo; hi <= 4 * lo && hi <= steps && hi <= 8; hi += lo) {
                    // the long items fill whole rounds; the tail starts at the first tile row after them (whole rows: the
                    // item order of any rectangular sub-range then stays "long first")
                    const long long full_rounds = tiles * lo / n_cus;
                    for (long long r = full_rounds; r >= 0 && r + 1 >= full_rounds; r--) {
                        const long long long_tiles = std::min(tiles, r * n_cus / lo);
                        const int lo_rows = (int)(long_tiles / ntx), tail_rows = nty - lo_rows;
                        if (tail_rows <= 0 || tail_rows >= nty) continue;
                        const double t = sym_makespan((long long)lo_rows * ntx * lo, cost(lo), (long long)tail_rows * ntx * hi, cost(hi), n_cus);
                        if (t < best * 0.97) {     // a mixed launch has to be worth its second patch geometry
                            best = t;
                            c = Choice{lo, hi, tail_rows};
                        }
                    }
                }
            cache[key] = c;
        }
    }
    w.n_parts = c.lo;
    if (c.hi) {
        w.sym.parts_hi = c.hi;
        w.sym.split_ty = w.sym.ty0 + nty - c.tail_rows;
    }
}

// filter<float3> and filter<float> (two buffers per launch), radius 1..20, every spec: Welch degrees of freedom run the Welch builds
// (six or eight feature planes, like the others).
// G-buffers: up to two RGB images (six feature planes, the shipped normal + albedo), or up to two RGB and up to two
// 1-channel images in any order (eight feature planes: + depth + material id; block + halo calls carry them in a
// 17-channel packed image)
bool sym_eligible(const FilterArgs &a, int channels) {
    if (a.radius < 1 || a.radius > sym::kR || (channels != 1 && channels != 3)) return false;
    // Welch degrees of freedom: the pair needs the sample counts -- from their own images (one RGB buffer, or two float
    // buffers per launch), or from the last channel of a 16- / 18-channel block + halo image (and those images are for the Welch builds only)
    if (a.dof != STATMC_DOF_PIXEL && a.packed && a.packed_ch != 16 && a.packed_ch != 18) return false;
    if (a.packed && (a.packed_ch == 16 || a.packed_ch == 18) && a.dof != STATMC_DOF_WELCH) return false;
    // the pair-symmetric kernel implements both gates and both channel rules; the clamped border's taps beyond the image
    // are added by border_virtual_kernel
    // float buffers with the one-sided gate (four weights per pair for two buffers): one-sided kernel
    if (a.gate != STATMC_GATE_SYMMETRIC && channels != 3 && a.dof == STATMC_DOF_PIXEL) return false;   // (Welch: one test, no gate form)
    // (a clamped border on a block + halo image: the border kernel reads the packed image -- round 5; before, such calls ran the
    // one-sided kernel and a block decomposition was not bit-identical to the whole film, which runs this kernel)
    int n_rgb = 0, n_sc = 0;
    for (int g = 0; g < a.n_g; g++) {
        if (a.g[g].channels == 3) n_rgb++;
        else if (a.g[g].channels == 1) n_sc++;
        else return false;
        if (!(a.g[g].dr <= 0.f) || !std::isfinite(a.g[g].dr)) return false;
    }
    if (n_rgb > 2 || n_sc > 2) return false;
    if (a.packed && n_sc > 0 && a.packed_ch != 17 && a.packed_ch != 18) return false;   // 1-channel features travel in the 17- / 18-channel block + halo image
    // (a 17-channel image with no 1-channel feature -- FilmShards packs every set other than exactly two RGB G-buffers that way --
    // runs the eight-plane build with its 1-channel slots at scale 0; the pack kernel writes zeros there)
    // (Welch with 1-channel features: the eight-plane Welch builds -- from the separate images, or from an 18-channel block + halo
    // image, which carries the 1-channel features AND the sample counts)
    return true;
}

void sym_feature_slots(FilterArgs &a) {
    a.sym.g8 = 0;
    for (int i = 0; i < 2; i++) {
        a.sym.rgb[i] = a.sym.sc[i] = nullptr;
        a.sym.rgb_scale[i] = a.sym.sc_scale[i] = 0.f;
    }
    int n_rgb = 0, n_sc = 0;
    for (int g = 0; g < a.n_g; g++) {
        const float scale = sqrtf(-a.g[g].dr * kLog2e);
        if (a.g[g].channels == 3 && n_rgb < 2) { a.sym.rgb[n_rgb] = a.g[g].data; a.sym.rgb_scale[n_rgb++] = scale; }
        else if (a.g[g].channels == 1 && n_sc < 2) { a.sym.sc[n_sc] = a.g[g].data; a.sym.sc_scale[n_sc++] = scale; }
    }
    a.sym.g8 = n_sc > 0 || (a.packed && (a.packed_ch == 17 || a.packed_ch == 18));
}

hipError_t launch_sym(FilterArgs a, hipStream_t s) {
    using namespace sym;
    if (a.rx1 <= a.rx0 || a.ry1 <= a.ry0) return hipSuccess;
    a.sym.steps = a.radius + 1;
    a.sym.item_stride4 = kPatchP + (long long)q_rows_max(a.n_parts, a.sym.steps) * kP;
    const bool welch = a.dof == STATMC_DOF_WELCH;
    const bool rt = a.radius != kR || welch;    // (the Welch modes exist in the runtime-radius build only; it serves r = 20 as well)
    if (rt && a.sym.tab_rt == nullptr) return hipErrorInvalidValue;
    if (welch && (a.tq2 == nullptr || (a.sym.g8 && a.packed && a.packed_ch != 18) || (a.sym.pair ? (a.packed != nullptr || a.f_n[0] == nullptr || (a.f_active > 1 && a.f_n[1] == nullptr))
                                                                 : a.packed ? (a.packed_ch != 16 && a.packed_ch != 18) : a.n == nullptr)))
        return hipErrorInvalidValue;
    // LDS-DMA staging needs whole 16-byte pieces: images 16-byte aligned, width and film x-origin multiples of 4 pixels
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool g8 = a.sym.g8 != 0;
    // the feature images the kernel will read (NG = 6: the argument list's first two; NG = 8: the sorted slots)
    const float *f_img[4] = {g8 ? a.sym.rgb[0] : (a.gscale0 != 0.f ? a.g[0].data : nullptr), g8 ? a.sym.rgb[1] : (a.gscale1 != 0.f ? a.g[1].data : nullptr),
                             g8 ? a.sym.sc[0] : nullptr, g8 ? a.sym.sc[1] : nullptr};
    const bool f_al = al16(f_img[0]) && al16(f_img[1]) && al16(f_img[2]) && al16(f_img[3]);
    bool dma = a.width % 4 == 0 && a.sym.fx0 % 4 == 0;
    if (a.packed) dma = dma && al16(a.packed);
    else dma = dma && al16(a.mean_corr) && al16(a.disc) && al16(a.colour) && f_al;
    const bool pair = a.sym.pair != 0;
    if (pair) {   // the launch's two float buffers -> the RGB-shaped images the rows are staged from
        float *mc3 = a.sym.pair_images, *d3 = mc3 + (size_t)a.width * a.height * 3, *c3 = d3 + (size_t)a.width * a.height * 3;
        // only the rows this launch's tiles read (ROI rows, the window above and below, rounded out to whole tiles): a
        // band of the Upload / Denoise / Download pipeline neither repacks the whole image nor touches rows whose copies
        // are still in flight
        const int y_lo = std::max(0, a.ry0 - kR - (kRows - 1)), y_hi = std::min(a.height, a.ry1 + kR);
        const long long p0 = (long long)y_lo * a.width, p1 = (long long)y_hi * a.width;
        hipLaunchKernelGGL(pack_pair_kernel, dim3((unsigned)((p1 - p0 + 255) / 256)), dim3(256), 0, s, a, mc3, d3, c3, p0, p1);
        a.mean_corr = mc3;
        a.disc = d3;
        a.colour = c3;
        dma = a.width % 4 == 0 && a.sym.fx0 % 4 == 0 && al16(mc3) && al16(d3) && al16(c3) && f_al;
    }
    // float buffers have one channel: pooled == per channel
    const bool joint = a.channel_rule == STATMC_CHANNELS_JOINT, asym = a.gate == STATMC_GATE_ASYMMETRIC, centre = a.gate == STATMC_GATE_CENTRE;
    const int mode = pair ? kModePair : centre ? (joint ? kModeCentreJoint : kModeCentre) : asym ? (joint ? kModeAsymJoint : kModeAsym) : joint ? kModeJoint : kModeRgb;
#define STATMC_SYM_K(D, M, G, R) reinterpret_cast<const void *>(&window_filter_sym<D, M, G, R>)
#define STATMC_SYM_ROW(D, G, R) {STATMC_SYM_K(D, kModeRgb, G, R), STATMC_SYM_K(D, kModePair, G, R), STATMC_SYM_K(D, kModeJoint, G, R), STATMC_SYM_K(D, kModeAsym, G, R), STATMC_SYM_K(D, kModeAsymJoint, G, R), STATMC_SYM_K(D, kModeCentre, G, R), STATMC_SYM_K(D, kModeCentreJoint, G, R)}
    // [runtime radius][eight feature planes][LDS-DMA staging][mode]
    const void *kernels[2][2][2][kModes] = {{{STATMC_SYM_ROW(false, 6, false), STATMC_SYM_ROW(true, 6, false)}, {STATMC_SYM_ROW(false, 8, false), STATMC_SYM_ROW(true, 8, false)}},
                                            {{STATMC_SYM_ROW(false, 6, true), STATMC_SYM_ROW(true, 6, true)}, {STATMC_SYM_ROW(false, 8, true), STATMC_SYM_ROW(true, 8, true)}}};
#undef STATMC_SYM_ROW
#undef STATMC_SYM_K
    const void *kernel = kernels[rt ? 1 : 0][g8 ? 1 : 0][dma ? 1 : 0][mode];
    // exactly two RGB G-buffers (whole film, or its 15-channel block + halo image), r = 20, the symmetric gate on one RGB buffer: the G7 builds
    if (!rt && !g8 && !welch && a.n_g == 2 && a.g[0].channels == 3 && a.g[1].channels == 3 && (mode == kModeRgb || mode == kModeJoint)) {
        if (mode == kModeRgb) kernel = dma ? reinterpret_cast<const void *>(&window_filter_sym<true, kModeRgb, 6, false, true>) : reinterpret_cast<const void *>(&window_filter_sym<false, kModeRgb, 6, false, true>);
        else kernel = dma ? reinterpret_cast<const void *>(&window_filter_sym<true, kModeJoint, 6, false, true>) : reinterpret_cast<const void *>(&window_filter_sym<false, kModeJoint, 6, false, true>);
    }
    const void *kernel_far = nullptr;
    if (welch) {   // (the gate field has no meaning under Welch: there is one test, symmetric in the pair)
        if (a.sym.redo == nullptr) return hipErrorInvalidValue;
#define STATMC_SYM_W(M) (g8 ? reinterpret_cast<const void *>(&window_filter_sym<false, M, 8, true>) : reinterpret_cast<const void *>(&window_filter_sym<false, M, 6, true>))
        kernel = pair ? STATMC_SYM_W(kModeWelchPair) : joint ? STATMC_SYM_W(kModeWelchJoint) : STATMC_SYM_W(kModeWelch);
        kernel_far = pair ? STATMC_SYM_W(kModeWelchPairFar) : joint ? STATMC_SYM_W(kModeWelchJointFar) : STATMC_SYM_W(kModeWelchFar);
#undef STATMC_SYM_W
    }
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const void *kf : {kernel, kernel_far}) {
            if (kf == nullptr || done.count({dev, kf})) continue;
            if (hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
            done.insert({dev, kf});
        }
    }
    const dim3 grid((unsigned)sym_items(a));
    void *kargs[] = {&a};
    const size_t lds_bytes = welch ? (g8 ? Planes<8, true>::kLdsBytes : Planes<6, true>::kLdsBytes) : g8 ? Planes<8>::kLdsBytes : Planes<6>::kLdsBytes;
    if (hipError_t e = hipLaunchKernel(kernel, grid, dim3(kThreads), kargs, lds_bytes, s); e != hipSuccess) return e;
    if (kernel_far != nullptr) {   // Welch: the items the band build flagged, again with the table in global memory
        if (hipError_t e = hipLaunchKernel(kernel_far, grid, dim3(kThreads), kargs, lds_bytes, s); e != hipSuccess) return e;
    }
    if (a.sym.border_extra) {
        if (hipError_t e = launch_border_virtual(a, s); e != hipSuccess) return e;
    }
    const dim3 cgrid((a.rx1 - a.rx0 + 63) / 64, (a.ry1 - a.ry0 + 3) / 4);
    hipLaunchKernelGGL(combine_sym_kernel, cgrid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace statmc
