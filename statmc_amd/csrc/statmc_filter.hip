// statmc_filter.hip -- the statistics-gated cross-bilateral window filter (gfx950).
//
// Replaces the window part of cv::cuda::stat_denoiser::filter<T> (call sites
// src/statistics/estimator.cpp:437-487 of the reference; its CUDA source is not in the tree,
// the arithmetic is this build's own frozen spec: DESIGN.md "Filter spec v1.1", restated on the
// CPU in oracle/statmc_oracle.c:oracle_filter).
//
// Two implementations:
//
//  * window_filter_lds<R, K>  -- the hot kernel.  Two 3-channel G-buffers (the shipped
//    configuration: filtering under normal + albedo), radius <= 20; K = 0 is filter<float3>
//    (one RGB buffer), K = 1..3 is filter<float> (up to three 1-channel buffers per launch, K real).
//    This is a stencil, not a contraction: 1681 taps x 17 fp32 VALU instructions per pixel
//    against 72 B of compulsory HBM traffic, so the design goal is VALU issue rate, with LDS as
//    the operand feed:
//      - a 512-thread workgroup (8 waves = 2 per SIMD) owns a 256 x 8 output tile; each wave owns
//        one row, each lane 4 adjacent pixels of it;
//      - the 15 per-pixel floats a tap needs (scaled normal 3, scaled albedo 3, corrected mean 3,
//        -discriminator 3, colour 3) are staged in LDS as 15 channel planes, one image row at a
//        time, in a ring of 9 rows (8 live + 1 being filled) -- the full (256+40) x (8+40) halo
//        would need 850 KB, the ring needs 156 KB of the CU's 160 KB;
//      - per window row a lane reads its 44-column span in 11 read groups of 15 ds_read_b128
//        (conflict-free: consecutive lanes read consecutive 16 B) and evaluates 4 taps x 4 pixels
//        per group in registers, so every LDS value is used 4 times;
//      - the work is done on TAP PAIRS with v_pk_add/mul/fma_f32 (both taps of a pair per
//        instruction, the pixel's own value broadcast through op_sel): 17 instructions per
//        (tap, pixel) pair.  G-buffers are pre-multiplied by sqrt(-DR_g * log2 e) when staged and
//        the spatial term comes from a pairs table of the window row kept in LDS, so the range
//        weight is 6 x (pk_sub, pk_mul|pk_fma) + pk_add per tap pair and one v_exp_f32 per tap;
//      - membership keeps the oracle's arithmetic t_c = fma(d_c, d_c, -D_q,c) and tests
//        max_c (t_c - D_p,c) <= 0 (same truth value as AND_c t_c <= D_p,c, no scalar-unit
//        round trip); the work is written stage by stage across the lane's 4 pixels so every
//        dependent step is followed by independent instructions;
//      - a last tile column of at most half a tile (1920 = 7.5 x 256) is covered by DUAL tiles,
//        128 x 15 pixels with the two halves of every wave on different rows, work items of the
//        same grid: 1080p is 945 + 72 workgroups = 4 rounds of the 1-workgroup-per-CU grid;
//      - where the tile count does not fill the rounds, the window rows of a tile are split over
//        `parts` workgroups whose partial sums combine_parts_kernel adds up; work items are
//        remapped so that each XCD's L2 sees one contiguous range of tiles.
//    Taps outside the image (and pixels with NaN statistics or a non-finite mean) are staged with a
//    NaN corrected mean, which fails every comparison, so clipping costs nothing in the inner loop.
//
//  * window_filter_generic<C> -- any radius, any G-buffer set, T = float or float3, one lane
//    per pixel straight from global memory.  Correctness path for configurations the hot
//    kernel does not cover (float G-buffers, radius > 20, other G-buffer sets).

#include <math.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <set>
#include <utility>

#include <stdio.h>

#include "statmc_device.h"
#include "statmc_filter_common.h"

namespace statmc {

// (dispatch overrides -- forced kernel variant, pinned window-sweep split -- are per-device state of the C-ABI layer
// and reach this file in FilterArgs::force_variant / force_parts)


// ====================================================================== generic kernel
// Every option of statmc_filter_spec, one lane per pixel, straight from global memory; the statements
// follow oracle/statmc_oracle.c:oracle_filter_spec_run line by line (same operation order, no contraction).
// (mc / dc / col: the buffer's corrected mean, discriminator and colour images -- a.mean_corr / a.disc / a.colour, or
// one of the float buffers of a multi-buffer launch)
// (S: floats between two pixels of an image -- C, or the channel count of a block + halo image the three are views of)
template <int C>
__device__ __forceinline__ bool pixel_valid(const float *mc, const float *dc, const float *col, long long p, int S = C) {
    bool v = true;
#pragma unroll
    for (int c = 0; c < C; c++) {
        v = v && __builtin_isfinite(mc[p * S + c]);
        v = v && !__builtin_isnan(dc[p * S + c]);
        v = v && __builtin_isfinite(col[p * S + c]);
    }
    return v;
}
// ... and (spec v2.1) every G-buffer value of the pixel is finite
__device__ __forceinline__ bool features_valid(const FilterArgs &a, long long p) {
    bool v = true;
    for (int g = 0; g < a.n_g; g++)
        for (int c = 0; c < a.g[g].channels; c++) v = v && __builtin_isfinite(a.g[g].data[p * a.g[g].channels + c]);
    return v;
}
template <int C>
__device__ __forceinline__ bool pixel_valid(const FilterArgs &a, long long p) {
    return pixel_valid<C>(a.mean_corr, a.disc, a.colour, p) && features_valid(a, p);
}

// (S, NS: floats / int32 between two pixels of the images and of the counts: C and 1, or a block + halo image's channel count)
template <int C>
__device__ __forceinline__ bool pair_member(const FilterArgs &a, const float *mc, const float *dc, const float *pc, const float *pd,
                                            long long p, long long q, const int32_t *n = nullptr, int S = C, int NS = 1) {
    if (n == nullptr) n = a.n;   // (the buffer's own counts: filter<float> with several buffers per launch)
    bool all = true;
    float lhs_sum = 0.f, rhs_sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float d = pc[c] - mc[q * S + c];
        const float Dp = pd[c], Dq = dc[q * S + c];
        float lhs, rhs;
        if (a.dof == STATMC_DOF_WELCH) {
            const float s = Dp + Dq;
            float Dsum = s;
            if (s > 0.f && __builtin_isfinite(s)) {
                const float nu = (s * s) / (Dp * Dp / ((float)n[p * NS] - 1.f) + Dq * Dq / ((float)n[q * NS] - 1.f));
                const int dof = nu >= 1.f ? (nu < 4096.f ? (int)nu : 4096) : 1;
                const float t = a.tq[dof - 1];
                Dsum = (t * t) * s;
            }
            lhs = __builtin_fmaf(d, d, -Dsum);
            rhs = 0.f;
        } else if (a.gate == STATMC_GATE_CENTRE) {
            lhs = d * d;
            rhs = Dp;
        } else if (a.gate == STATMC_GATE_ASYMMETRIC) {
            lhs = __builtin_fmaf(d, d, -Dq);
            rhs = Dp;
        } else {
            lhs = __builtin_fmaf(d, d, -(Dp + Dq));
            rhs = 0.f;
        }
        all = all & (lhs <= rhs);
        lhs_sum = c == 0 ? lhs : lhs_sum + lhs;
        rhs_sum = c == 0 ? rhs : rhs_sum + rhs;
    }
    return a.channel_rule == STATMC_CHANNELS_JOINT ? (lhs_sum <= rhs_sum) : all;
}
template <int C>
__device__ __forceinline__ bool pair_member(const FilterArgs &a, const float *pc, const float *pd, long long p, long long q) {
    return pair_member<C>(a, a.mean_corr, a.disc, pc, pd, p, q);
}

template <int C>
__global__ __launch_bounds__(256) void window_filter_generic(FilterArgs a) {
    const int x = a.rx0 + blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = a.ry0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.rx1 || y >= a.ry1) return;
    const long long p = (long long)y * a.width + x;
    float pc[C], pd[C], acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        pc[c] = a.mean_corr[p * C + c];
        pd[c] = a.disc[p * C + c];
        acc[c] = 0.f;
    }
    float sum_w = 0.f;
    const int r = a.radius;
    const bool clamp = a.border == STATMC_BORDER_CLAMP;
    // a pixel whose statistics or colour are not finite takes no part (spec: DESIGN.md "Filter spec")
    const bool p_valid = pixel_valid<C>(a, p);
    for (int dy = -r; p_valid && dy <= r; dy++) {
        int qy = y + dy;
        if (qy < 0 || qy >= a.height) {
            if (!clamp) continue;
            qy = qy < 0 ? 0 : a.height - 1;
        }
        for (int dx = -r; dx <= r; dx++) {
            int qx = x + dx;
            if (qx < 0 || qx >= a.width) {
                if (!clamp) continue;
                qx = qx < 0 ? 0 : a.width - 1;
            }
            const long long q = (long long)qy * a.width + qx;
            if (!pixel_valid<C>(a, q)) continue;
            if (!pair_member<C>(a, pc, pd, p, q)) continue;
            float e = a.ds * (float)(dx * dx + dy * dy);
            for (int g = 0; g < a.n_g; g++) {
                const int gc = a.g[g].channels;
                const float *G = a.g[g].data;
                const float d0 = G[p * gc] - G[q * gc];
                float dist2 = d0 * d0;
                for (int c = 1; c < gc; c++) {
                    const float dc = G[p * gc + c] - G[q * gc + c];
                    dist2 = __builtin_fmaf(dc, dc, dist2);
                }
                e = __builtin_fmaf(a.g[g].dr, dist2, e);
            }
            const float w = __builtin_amdgcn_exp2f(e * kLog2e);
            sum_w += w;
#pragma unroll
            for (int c = 0; c < C; c++) acc[c] = __builtin_fmaf(w, a.colour[q * C + c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) a.out[p * C + c] = sum_w > 0.f ? acc[c] / sum_w : a.colour[p * C + c];
}

// Border rule "clamp" on the pair-symmetric kernel: that kernel sums the taps inside the image; the taps beyond the
// image, which the clamped border maps onto the edge pixels, exist for the pixels within r of an edge only (6 % of a
// 1080p film) and have no mirror pair.  This kernel adds exactly those taps into one float4 per pixel
// (sum w * colour, sum w) that combine_sym_kernel adds to the pixel's patches.  All the taps of one window column
// that lie above the image land on the same pixel of row 0 with the same membership and range weight; they differ in
// their spatial weight only, and that factorises: sum_dy exp(ds (dx^2 + dy^2)) = exp(ds dx^2) * sum_dy exp(ds dy^2).
// So a pixel evaluates at most 4 x 41 (pixel, edge pixel) pairs instead of up to 1200 taps.
// NB = 0: one RGB buffer (a.mean_corr / a.disc / a.colour).  NB = 2: the two float buffers of a filter<float> launch
// (a.f_mean_corr[b] ...; the sums go to (x, z) and (y, w) of the float4, as combine_sym_kernel expects them).
// PACKED (NB = 0): the inputs are the channels of a block + halo image (a.packed: mean 0..2, discriminator 3..5, colour 6..8,
// the RGB G-buffers of the argument list from 9, its 1-channel ones from 15, the count's bits in the last channel of a 16- /
// 18-channel image).  The local image's edges that are film edges have no halo, its other edges one of >= r pixels: a pixel of
// the ROI has taps beyond the local image exactly where it has taps beyond the film.
template <int NB, bool PACKED = false>
__global__ __launch_bounds__(256) void border_virtual_kernel(FilterArgs a) {
    constexpr int C = NB == 0 ? 3 : 1;
    static_assert(!PACKED || NB == 0, "a block + halo image holds one RGB buffer");
    const int S = PACKED ? a.packed_ch : C;
    // the G-buffers as (first float, channels, factor) views: the argument list's images, or their slots in the packed image
    const float *g_data[STATMC_MAX_GBUFFERS];
    int g_stride[STATMC_MAX_GBUFFERS];
    {
        int n_rgb = 0, n_sc = 0;
        for (int g = 0; g < a.n_g; g++) {
            if constexpr (PACKED) {
                g_data[g] = a.g[g].channels == 3 ? a.packed + 9 + 3 * n_rgb++ : a.packed + 15 + n_sc++;
                g_stride[g] = S;
            } else {
                g_data[g] = a.g[g].data;
                g_stride[g] = a.g[g].channels;
            }
        }
    }
    auto feats_valid = [&](long long p) {
        bool v = true;
        for (int g = 0; g < a.n_g; g++)
            for (int c = 0; c < a.g[g].channels; c++) v = v && __builtin_isfinite(g_data[g][p * g_stride[g] + c]);
        return v;
    };
    const int x = a.rx0 + blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = a.ry0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.rx1 || y >= a.ry1) return;
    const int r = a.radius, W = a.width, H = a.height;
    if (x >= r && x < W - r && y >= r && y < H - r) return;   // every tap of this pixel is inside the image
    const long long p = (long long)y * W + x;
    // summed spatial weights of the window rows above / below and of the window columns left / right of the image
    float t_top = 0.f, t_bot = 0.f, t_left = 0.f, t_right = 0.f;
    for (int d = 1; d <= r; d++) {
        const float sw = __builtin_amdgcn_exp2f(a.ds * (float)(d * d) * kLog2e);
        if (d > y) t_top += sw;
        if (d > H - 1 - y) t_bot += sw;
        if (d > x) t_left += sw;
        if (d > W - 1 - x) t_right += sw;
    }
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int kBuffers = NB == 0 ? 1 : NB;
#pragma unroll
    for (int b = 0; b < kBuffers; b++) {
        if (NB != 0 && b >= a.f_active) break;
        const float *mc = PACKED ? a.packed : NB == 0 ? a.mean_corr : a.f_mean_corr[b];
        const float *dc = PACKED ? a.packed + 3 : NB == 0 ? a.disc : a.f_disc[b];
        const float *col = PACKED ? a.packed + 6 : NB == 0 ? a.colour : a.f_colour[b];
        const int32_t *cnt = PACKED ? reinterpret_cast<const int32_t *>(a.packed) + (a.packed_ch == 18 ? 17 : 15) : NB == 0 ? a.n : a.f_n[b];
        float pc[C], pd[C], acc[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            pc[c] = mc[p * S + c];
            pd[c] = dc[p * S + c];
            acc[c] = 0.f;
        }
        float sum_w = 0.f;
        if (pixel_valid<C>(mc, dc, col, p, S) && feats_valid(p)) {
            // one (pixel, edge pixel) pair: membership, range weight, times the summed spatial weight of the taps it stands for
            auto pair = [&](int qx, int qy, int d_along, float t_across) {
                const long long q = (long long)qy * W + qx;
                if (!pixel_valid<C>(mc, dc, col, q, S) || !feats_valid(q)) return;
                if (!pair_member<C>(a, mc, dc, pc, pd, p, q, cnt, S, PACKED ? S : 1)) return;
                float e = a.ds * (float)(d_along * d_along);
                for (int g = 0; g < a.n_g; g++) {
                    const int gc = a.g[g].channels, gs = g_stride[g];
                    const float *G = g_data[g];
                    const float d0 = G[p * gs] - G[q * gs];
                    float dist2 = d0 * d0;
                    for (int c = 1; c < gc; c++) {
                        const float dcc = G[p * gs + c] - G[q * gs + c];
                        dist2 = __builtin_fmaf(dcc, dcc, dist2);
                    }
                    e = __builtin_fmaf(a.g[g].dr, dist2, e);
                }
                const float w = __builtin_amdgcn_exp2f(e * kLog2e) * t_across;
                sum_w += w;
#pragma unroll
                for (int c = 0; c < C; c++) acc[c] = __builtin_fmaf(w, col[q * S + c], acc[c]);
            };
            for (int dx = -r; dx <= r; dx++) {   // rows beyond the image (corners included): edge rows, clamped column
                const int tx = x + dx, qx = tx < 0 ? 0 : tx >= W ? W - 1 : tx;
                if (t_top > 0.f) pair(qx, 0, dx, t_top);
                if (t_bot > 0.f) pair(qx, H - 1, dx, t_bot);
            }
            for (int dy = -r; dy <= r; dy++) {   // columns beyond the image, rows inside it: edge columns
                const int ty = y + dy;
                if (ty < 0 || ty >= H) continue;
                if (t_left > 0.f) pair(0, ty, dy, t_left);
                if (t_right > 0.f) pair(W - 1, ty, dy, t_right);
            }
        }
        if (NB == 0) {
            out[0] = acc[0]; out[1] = acc[C - 1 >= 1 ? 1 : 0]; out[2] = acc[C - 1 >= 2 ? 2 : 0]; out[3] = sum_w;
        } else {
            out[b] = acc[0];
            out[2 + b] = sum_w;
        }
    }
    a.sym.border_extra[p] = make_float4(out[0], out[1], out[2], out[3]);
}

hipError_t launch_border_virtual(const FilterArgs &a, hipStream_t s) {
    const dim3 grid((a.rx1 - a.rx0 + 31) / 32, (a.ry1 - a.ry0 + 7) / 8);
    if (a.packed && a.sym.pair) return hipErrorInvalidValue;
    if (a.packed) hipLaunchKernelGGL((border_virtual_kernel<0, true>), grid, dim3(256), 0, s, a);
    else if (a.sym.pair) hipLaunchKernelGGL(border_virtual_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(border_virtual_kernel<0>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ====================================================================== LDS kernel
constexpr int kPx = 4;                 // pixels per lane
constexpr int kTileW = 64 * kPx;       // 256 columns per wave-row
constexpr int kTileH = 8;              // rows per tile = waves per workgroup
constexpr int kThreads = 64 * kTileH;  // 512
constexpr int kSlots = kTileH + 1;     // LDS row ring
constexpr int kCh = 15;                // floats staged per pixel
constexpr int kMaxR = 20;

// Tile geometry of the LDS kernel.  Regular: 256 x 8 pixels, wave = one row.  DUAL: 128 x 15 pixels
// for the last tile column of images whose width leaves at most half a tile over (1920 = 7.5 x 256):
// the lower and upper half of every wave own different rows (w and w + 8), so that column costs 72
// workgroups at 1080p instead of 135 half-empty ones; the 16-slot ring of 168-column rows fits the
// same 160 KB of LDS.
template <bool DUAL>
struct Geo {
    static constexpr int W = DUAL ? 32 * kPx : kTileW;     // tile width in pixels
    static constexpr int ROWS = DUAL ? 2 * kTileH - 1 : kTileH;
    static constexpr int SLOTS = ROWS + 1;                 // LDS row ring
};

// LDS row layout (floats; P = pitch = columns staged per row, a multiple of 4): 15 planes of P
// floats, one per channel -- scaled normal 0..2, scaled albedo 3..5, corrected mean 6..8,
// -discriminator 9..11, colour 12..14.  A lane's ds_read_b128 of a plane returns one channel of 4
// adjacent taps; consecutive lanes read consecutive 16 B (conflict-free).  The inner loop works on
// TAP PAIRS with packed fp32 (v_pk_add/mul/fma_f32: both taps of a pair in one instruction, the
// pixel's own value broadcast through op_sel): the kernel is bound by VALU instruction issue, and
// pairing taps -- rather than channels -- needs no cross-half adds and also packs the
// single-channel work (17 instructions per (tap, pixel) pair instead of 21).

__host__ __device__ constexpr int round_up4(int r) { return (r + 3) & ~3; }
__host__ __device__ constexpr int tab_width(int rp) { return 2 * rp + 7; }

size_t spatial_table_floats(int radius) {
    if (radius < 0 || radius > kMaxR) return 0;
    return (size_t)(2 * radius + 1) * tab_width(round_up4(radius));
}

// tab[dy + r][dx + rp + 3] = log2(e) * ds * (dx^2 + dy^2) for |dx| <= r, -inf otherwise.
void fill_spatial_table(float *tab, int radius, float ds) {
    const int rp = round_up4(radius), tw = tab_width(rp);
    for (int dy = -radius; dy <= radius; dy++)
        for (int i = 0; i < tw; i++) {
            const int dx = i - rp - 3;
            const float e = ds * (float)(dx * dx + dy * dy);  // same product as the oracle
            tab[(dy + radius) * tw + i] = (dx >= -radius && dx <= radius) ? e * kLog2e : -INFINITY;
        }
}

// The pair-symmetric kernel's runtime-radius build keeps the r = 20 staging geometry: tab[dy][dx + 23] for dy = 0 .. r,
// 47 entries per row, -inf for |dx| > r (those taps get weight 0).
size_t sym_rt_table_floats(int radius) { return radius < 1 || radius > kMaxR ? 0 : (size_t)(radius + 1) * tab_width(kMaxR); }
void fill_sym_rt_table(float *tab, int radius, float ds) {
    const int tw = tab_width(kMaxR);
    for (int dy = 0; dy <= radius; dy++)
        for (int i = 0; i < tw; i++) {
            const int dx = i - kMaxR - 3;
            const float e = ds * (float)(dx * dx + dy * dy);  // same product as the oracle
            tab[dy * tw + i] = (dx >= -radius && dx <= radius) ? e * kLog2e : -INFINITY;
        }
}

template <int J, int RT>
struct ChunkMask {
    // bit (i*4+k) set when tap i of chunk J is inside the window of pixel k (compile-time R)
    static constexpr unsigned value() {
        unsigned m = 0;
        constexpr int rp = round_up4(RT);
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) {
                const int dx = 4 * J + i - k - rp;
                if (dx >= -RT && dx <= RT) m |= 1u << (i * 4 + k);
            }
        return m;
    }
};

// The lane's own 4 pixels.  Accumulators are 2-wide: .x collects the even taps of every read
// group, .y the odd ones; the epilogue adds the halves.
struct LaneState {
    float pg[kPx][6];            // scaled normal, scaled albedo
    float pmc[kPx][3], pd[kPx][3];
    v2f acc[kPx][3];
    v2f sw[kPx][3];              // RGB uses sw[k][0] only; float mode has one weight sum per buffer
};

// Operands of half a read group: taps 2H, 2H+1 (columns 4*(lane+j) + 2H, +1 of the staged row) of
// all 15 channels, and the spatial exponents of the 4 (tap pair, pixel) combinations.  The LDS
// copy of the window row's table holds PAIRS, tabp[t] = (tab[t], tab[t+1]), so the pair for pixel k
// -- taps at dx and dx+1 -- is one aligned 8-byte broadcast read and needs no register shuffling.
struct HalfChunk {
    v2f q[kCh];
    v2f tp[kPx];  // tp[k] = (tab[4j + 2H - k + 3], tab[4j + 2H - k + 4])
};

template <int H>
__device__ __forceinline__ void load_half(HalfChunk &c, const float *__restrict__ row, int pitch,
                                          const float *__restrict__ tab, int j) {
#pragma unroll
    for (int ch = 0; ch < kCh; ch++) c.q[ch] = *reinterpret_cast<const v2f *>(row + ch * pitch + 4 * j + 2 * H);
#pragma unroll
    for (int k = 0; k < kPx; k++) c.tp[k] = *reinterpret_cast<const v2f *>(tab + 2 * (4 * j + 2 * H - k + 3));
}

// Taps 2H and 2H+1 of the read group against the lane's 4 pixels: 4 tap-pair evaluations, written
// stage by stage across the pixels so every dependent step is followed by independent work.
// MASK: bit (i*4+k) set when tap i lies inside the window of pixel k (all 16 bits in the
// runtime-radius variant, where the table holds -inf beyond the radius); a pair is evaluated when
// either of its taps is inside, the outside tap of a partial pair is switched off by the table
// (static variant: by a compile-time zero weight).
// RGB = true: filter<float3> -- one buffer, membership is the AND over its three channels.
// RGB = false: filter<float> -- the three "channels" are three independent 1-channel buffers
// (ACRR bounces / SMIS win rates) that share the range weight but gate and normalise separately.
template <int H, unsigned MASK>
struct TapMask {
    static constexpr int i0 = 2 * H;
    static constexpr bool in0(int k) { return (MASK & (1u << (i0 * 4 + k))) != 0; }
    static constexpr bool in1(int k) { return (MASK & (1u << ((i0 + 1) * 4 + k))) != 0; }
    static constexpr bool on(int k) { return in0(k) || in1(k); }
};

// range weight exponent of a tap pair against the lane's 4 pixels: tab - |k_n dn|^2 - |k_a da|^2
template <int H, unsigned MASK>
__device__ __forceinline__ void range_exponent(const LaneState &st, const v2f *g, const v2f *tp, v2f (&e)[kPx]) {
    using M = TapMask<H, MASK>;
#pragma unroll
    for (int k = 0; k < kPx; k++) if (M::on(k)) { const v2f d = st.pg[k][0] - g[0]; e[k] = -d * d; }
#pragma unroll
    for (int ch = 1; ch < 6; ch++) {
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) { const v2f d = st.pg[k][ch] - g[ch]; e[k] = __builtin_elementwise_fma(-d, d, e[k]); }
    }
#pragma unroll
    for (int k = 0; k < kPx; k++) if (M::on(k)) e[k] += tp[k];
}

// membership gate, weight and accumulation of a tap pair (mc / nd / col: corrected mean,
// -discriminator, colour of the two taps, 3 channels each)
// SPEC: the membership test of the filter spec (statmc_filter_spec, oracle pair_member).  0 = the default (symmetric
// gate, every channel passes); kSpecAsym = one-sided gate fma(d, d, -D_q) <= D_p; kSpecJoint = channels pooled,
// sum_c lhs_c <= sum_c rhs_c (an RGB buffer only: a float buffer has one channel, pooled == per channel).
constexpr int kSpecAsym = 1, kSpecJoint = 2;

template <int H, unsigned MASK, int K, int SPEC>
__device__ __forceinline__ void gate_accumulate(LaneState &st, const v2f (&e)[kPx], const v2f *mc, const v2f *nd, const v2f *col) {
    using M = TapMask<H, MASK>;
    constexpr bool RGB = K == 0;
    constexpr int NB = RGB ? 3 : K;  // float mode: only the K real buffers of the launch are gated and summed
    constexpr bool ASYM = (SPEC & kSpecAsym) != 0, JOINT = RGB && (SPEC & kSpecJoint) != 0;
    v2f u[kPx][3], w[kPx];
    // membership statistic per channel: t_c = fma(d_c, d_c, -(D_p,c + D_q,c))  (the oracle's expression; nd = -D_q);
    // one-sided gate: fma(d_c, d_c, -D_q,c), compared with D_p,c below
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) {
            const v2f d = st.pmc[k][ch] - mc[ch];
            const v2f s = ASYM ? nd[ch] : nd[ch] - st.pd[k][ch];
            u[k][ch] = __builtin_elementwise_fma(d, d, s);
        }
    }
#pragma unroll
    for (int k = 0; k < kPx; k++) if (M::on(k)) w[k] = v2f{__builtin_amdgcn_exp2f(e[k].x), __builtin_amdgcn_exp2f(e[k].y)};
    if constexpr (RGB && SPEC != 0) {
        // the non-default tests as the oracle writes them: plain compares (a NaN statistic fails every one)
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) {
            bool m0, m1;
            if constexpr (JOINT) {
                const v2f lhs = (u[k][0] + u[k][1]) + u[k][2];
                const float rhs = ASYM ? (st.pd[k][0] + st.pd[k][1]) + st.pd[k][2] : 0.f;
                m0 = lhs.x <= rhs;
                m1 = lhs.y <= rhs;
            } else {
                m0 = u[k][0].x <= st.pd[k][0] && u[k][1].x <= st.pd[k][1] && u[k][2].x <= st.pd[k][2];
                m1 = u[k][0].y <= st.pd[k][0] && u[k][1].y <= st.pd[k][1] && u[k][2].y <= st.pd[k][2];
            }
            w[k] = v2f{M::in0(k) && m0 ? w[k].x : 0.f, M::in1(k) && m1 ? w[k].y : 0.f};
        }
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) st.sw[k][0] += w[k];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
#pragma unroll
            for (int k = 0; k < kPx; k++) if (M::on(k)) st.acc[k][ch] = __builtin_elementwise_fma(w[k], col[ch], st.acc[k][ch]);
        }
    } else if constexpr (RGB) {
        // all three channels pass  <=>  max_c t_c <= 0; max3 + compare + select stays on the VALU, where 3
        // compares + 2 scalar ANDs send every pair through the scalar unit.  v_max3 drops NaN operands,
        // so a pixel that takes no part (NaN statistic, non-finite mean or colour) is staged with NaN in
        // all three channels of its mean (canonical_mean).
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) {
            const float m0 = __builtin_fmaxf(__builtin_fmaxf(u[k][0].x, u[k][1].x), u[k][2].x);
            const float m1 = __builtin_fmaxf(__builtin_fmaxf(u[k][0].y, u[k][1].y), u[k][2].y);
            w[k] = v2f{M::in0(k) && m0 <= 0.f ? w[k].x : 0.f, M::in1(k) && m1 <= 0.f ? w[k].y : 0.f};
        }
#pragma unroll
        for (int k = 0; k < kPx; k++) if (M::on(k)) st.sw[k][0] += w[k];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
#pragma unroll
            for (int k = 0; k < kPx; k++) if (M::on(k)) st.acc[k][ch] = __builtin_elementwise_fma(w[k], col[ch], st.acc[k][ch]);
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < NB; ch++) {
#pragma unroll
            for (int k = 0; k < kPx; k++) if (M::on(k)) {
                const float rhs = ASYM ? st.pd[k][ch] : 0.f;
                const v2f wc = v2f{M::in0(k) && u[k][ch].x <= rhs ? w[k].x : 0.f,
                                   M::in1(k) && u[k][ch].y <= rhs ? w[k].y : 0.f};
                st.sw[k][ch] += wc;
                st.acc[k][ch] = __builtin_elementwise_fma(wc, col[ch], st.acc[k][ch]);
            }
        }
    }
}

template <int H, unsigned MASK, int K, int SPEC>
__device__ __forceinline__ void compute_half(LaneState &st, const HalfChunk &c) {
    v2f e[kPx];
    range_exponent<H, MASK>(st, c.q + C_G0, c.tp, e);
    gate_accumulate<H, MASK, K, SPEC>(st, e, c.q + C_MC, c.q + C_ND, c.q + C_COL);
}

// Sweep one window row: 2*rp/4 + 1 read groups.  RT > 0 (compile-time radius, a multiple of 4):
// the first and last groups hold (tap, pixel) pairs outside the window and get their static
// masks; every group between is full and runs as a rolled loop.
template <int RT, int K, int SPEC>
__device__ __forceinline__ void eval_row(LaneState &st, const float *row, int pitch, const float *tab, int n_chunks) {
    HalfChunk c;
    constexpr unsigned kFull = 0xFFFFu;
    if constexpr (RT > 0) {
        constexpr int n = 2 * round_up4(RT) / 4 + 1;
        static_assert(RT % 4 == 0 && n >= 3, "static variant: radius multiple of 4");
        static_assert(ChunkMask<1, RT>::value() == kFull && ChunkMask<n - 2, RT>::value() == kFull, "");
        load_half<0>(c, row, pitch, tab, 0);
        compute_half<0, ChunkMask<0, RT>::value(), K, SPEC>(st, c);
        load_half<1>(c, row, pitch, tab, 0);
        compute_half<1, ChunkMask<0, RT>::value(), K, SPEC>(st, c);
#pragma unroll 1
        for (int j = 1; j < n - 1; j++) {
            load_half<0>(c, row, pitch, tab, j);
            compute_half<0, kFull, K, SPEC>(st, c);
            load_half<1>(c, row, pitch, tab, j);
            compute_half<1, kFull, K, SPEC>(st, c);
        }
        load_half<0>(c, row, pitch, tab, n - 1);
        compute_half<0, ChunkMask<n - 1, RT>::value(), K, SPEC>(st, c);
        load_half<1>(c, row, pitch, tab, n - 1);
        compute_half<1, ChunkMask<n - 1, RT>::value(), K, SPEC>(st, c);
    } else {
#pragma unroll 1
        for (int j = 0; j < n_chunks; j++) {
            load_half<0>(c, row, pitch, tab, j);
            compute_half<0, kFull, K, SPEC>(st, c);
            load_half<1>(c, row, pitch, tab, j);
            compute_half<1, kFull, K, SPEC>(st, c);
        }
    }
}
// RT > 0: compile-time radius (window edges resolved statically); RT == 0: runtime radius
// a.radius <= 20, every pair of every read group evaluated, the table masks taps beyond r.
// One work item: `part` of tile `tile` of the tile grid laid over columns [cx0, cx1) of the ROI.
template <int RT, int K, int SPEC, bool DUAL>
__device__ __forceinline__ void filter_tile(const FilterArgs &a, float *lds, int u, int cx0, int cx1) {
    using G = Geo<DUAL>;
    constexpr bool RGB = K == 0;  // K = 0: filter<float3>; K = 1..3: filter<float> with K real buffers in this launch
    const int r = RT > 0 ? RT : a.radius;
    const int rp = RT > 0 ? round_up4(RT) : round_up4(a.radius);
    const int pitch = G::W + 2 * rp;
    const int n_chunks = 2 * rp / 4 + 1;
    const int tw = tab_width(rp);
    const int tw_pad = 2 * (tw + 1);  // pairs (tab[t], tab[t+1]) for t = 0 .. tw-1, + one pad pair
    const int slot_floats = kCh * pitch;
    float *tab_lds = lds + G::SLOTS * slot_floats;  // two buffers: this window row's exponents / the next one's

    const int wave = threadIdx.x >> 6;
    // lane -> (row of the tile, 4-pixel column group): regular tiles give a wave one row; DUAL tiles give
    // its lower half row `wave` and its upper half row `wave + 8` (row 15 of a 15-row tile does not exist:
    // those 32 lanes compute a clamped copy and store nothing)
    const int lane = DUAL ? (int)(threadIdx.x & 31) : (int)(threadIdx.x & 63);
    const int trow = DUAL ? wave + kTileH * (int)((threadIdx.x >> 5) & 1) : wave;
    const int part = u % a.n_parts, tile = u / a.n_parts;
    const int tiles_x = (cx1 - cx0 + G::W - 1) / G::W;
    const int x0 = cx0 + (tile % tiles_x) * G::W;
    const int y0 = a.ry0 + (tile / tiles_x) * G::ROWS;
    const float k0 = a.gscale0, k1 = a.gscale1;
    // window rows [s0, s1) of the 2r+1 belong to this workgroup (`part`): splitting the
    // window sweep over several workgroups per tile evens out the last round of the 1-WG-per-CU
    // grid (1080 tiles on 256 CUs would otherwise leave 200 CUs idle for a fifth of the run).
    const int n_rows = 2 * r + 1;
    // ... clipped to the window rows that reach the image for at least one row of the tile (tiles at
    // the top and bottom of the film would otherwise sweep up to 13 rows of nothing but invalid taps)
    // (border rule "clamp": rows beyond the image repeat its first / last row, nothing to clip)
    const bool clamp = a.border == STATMC_BORDER_CLAMP;
    const int s0 = clamp ? (n_rows * part) / a.n_parts : max((n_rows * part) / a.n_parts, r - y0 - (G::ROWS - 1));
    const int s1 = clamp ? (n_rows * (part + 1)) / a.n_parts : min((n_rows * (part + 1)) / a.n_parts, a.height + r - y0);

    // ---- the lane's own 4 pixels (clamped into the image so the loads stay in bounds)
    LaneState st;
    const int py = min(y0 + trow, a.height - 1);
#pragma unroll
    for (int k = 0; k < kPx; k++) {
        const int px = min(x0 + kPx * lane + k, a.width - 1);
        const long long p = (long long)py * a.width + px;
        f3 mc, d, g0, g1, col;
        if (RGB && a.packed) {
            const f3 *px = reinterpret_cast<const f3 *>(a.packed + p * 15);
            mc = px[0]; d = px[1]; col = px[2]; g0 = px[3]; g1 = px[4];
        } else if constexpr (RGB) {
            mc = reinterpret_cast<const f3 *>(a.mean_corr)[p];
            d = reinterpret_cast<const f3 *>(a.disc)[p];
            col = reinterpret_cast<const f3 *>(a.colour)[p];
        } else {
            mc = f3{a.f_mean_corr[0][p], a.f_mean_corr[1][p], a.f_mean_corr[2][p]};
            d = f3{a.f_disc[0][p], a.f_disc[1][p], a.f_disc[2][p]};
            col = f3{a.f_colour[0][p], a.f_colour[1][p], a.f_colour[2][p]};
        }
        if (!(RGB && a.packed)) load_features(a, p, g0, g1);
        bool fin = features_finite(g0, g1);   // spec v2.1: a pixel with a non-finite feature takes no part
        if (!fin) g0 = g1 = f3{0.f, 0.f, 0.f};
        mc = canonical_mean(mc, pixel_validity(mc, d, col, fin, RGB));  // the same rule as for the staged taps
        st.pg[k][0] = g0.x * k0; st.pg[k][1] = g0.y * k0; st.pg[k][2] = g0.z * k0;
        st.pg[k][3] = g1.x * k1; st.pg[k][4] = g1.y * k1; st.pg[k][5] = g1.z * k1;
        st.pmc[k][0] = mc.x; st.pmc[k][1] = mc.y; st.pmc[k][2] = mc.z;
        st.pd[k][0] = d.x; st.pd[k][1] = d.y; st.pd[k][2] = d.z;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            st.acc[k][ch] = v2f{0.f, 0.f};
            st.sw[k][ch] = v2f{0.f, 0.f};
        }
    }

    // ---- prologue: window rows rel = s0 .. s0+ROWS-1 (image rows y0-r+rel) into slots 0..ROWS-1
    for (int idx = threadIdx.x; idx < G::ROWS * pitch; idx += kThreads) {
        const int rel = idx / pitch, i = idx - rel * pitch;
        const StagedPixel s = load_pixel<RGB>(a, x0 - rp + i, y0 - r + s0 + rel);
        store_pixel(lds + rel * slot_floats, pitch, i, s, k0, k1, RGB, a.gate == STATMC_GATE_CENTRE);
    }
    if ((int)threadIdx.x < tw) {
        const float *t = a.spatial_tab + s0 * tw + threadIdx.x;
        *reinterpret_cast<v2f *>(tab_lds + 2 * threadIdx.x) = v2f{t[0], (int)threadIdx.x + 1 < tw ? t[1] : 0.f};
    }
    __syncthreads();

    // ---- sweep the window rows; tile row t works on staged row rel = t + step
    int slot = trow;     // (trow + step - s0) % SLOTS
    int fill = G::ROWS;  // (step - s0 + ROWS) % SLOTS: the slot the next row is staged into
    for (int step = s0; step < s1; step++) {
        // issue the global loads of the row needed by the next step early
        const bool stage = step + 1 < s1 && (int)threadIdx.x < pitch;
        StagedPixel nxt;
        nxt.valid = false;
        if (stage) nxt = load_pixel<RGB>(a, x0 - rp + (int)threadIdx.x, y0 - r + step + G::ROWS);

        // the spatial exponents of the next window row go to the other table buffer (last wave's lanes)
        const int ti = (int)threadIdx.x - (kThreads - 64);
        const bool tstage = step + 1 < s1 && ti >= 0 && ti < tw;
        v2f tnext = v2f{0.f, 0.f};
        if (tstage) {
            const float *t = a.spatial_tab + (step + 1) * tw + ti;
            tnext = v2f{t[0], ti + 1 < tw ? t[1] : 0.f};
        }

        const float *row = lds + slot * slot_floats + kPx * lane;
        eval_row<RT, K, SPEC>(st, row, pitch, tab_lds + ((step - s0) & 1) * tw_pad, n_chunks);

        if (tstage) *reinterpret_cast<v2f *>(tab_lds + ((step - s0 + 1) & 1) * tw_pad + 2 * ti) = tnext;
        if (stage) store_pixel(lds + fill * slot_floats, pitch, threadIdx.x, nxt, k0, k1, RGB, a.gate == STATMC_GATE_CENTRE);
        __syncthreads();
        slot = slot + 1 == G::SLOTS ? 0 : slot + 1;
        fill = fill + 1 == G::SLOTS ? 0 : fill + 1;
    }

    // ---- epilogue
    const int oy = y0 + trow;
    if (trow < G::ROWS && oy < a.ry1) {
#pragma unroll
        for (int k = 0; k < kPx; k++) {
            const int ox = x0 + kPx * lane + k;
            if (ox < cx1) {
                const long long p = (long long)oy * a.width + ox;
                float acc[3], sw[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    acc[ch] = st.acc[k][ch].x + st.acc[k][ch].y;
                    sw[ch] = st.sw[k][ch].x + st.sw[k][ch].y;
                }
                if constexpr (RGB) {
                    if (a.n_parts > 1) {  // partial sums; combine_parts_kernel finishes the pixel
                        reinterpret_cast<float4 *>(a.partial)[(long long)part * a.width * a.height + p] =
                            make_float4(acc[0], acc[1], acc[2], sw[0]);
                        continue;
                    }
                    f3 o;
                    if (sw[0] > 0.f) {
                        o.x = acc[0] / sw[0];
                        o.y = acc[1] / sw[0];
                        o.z = acc[2] / sw[0];
                    } else {
                        o = a.packed ? reinterpret_cast<const f3 *>(a.packed + p * 15)[2] : reinterpret_cast<const f3 *>(a.colour)[p];
                    }
                    reinterpret_cast<f3 *>(a.out)[p] = o;
                } else {
                    if (a.n_parts > 1) {  // two float4 per (part, pixel): sums, then weights
                        float4 *dst = reinterpret_cast<float4 *>(a.partial) + 2 * ((long long)part * a.width * a.height + p);
                        dst[0] = make_float4(acc[0], acc[1], acc[2], 0.f);
                        dst[1] = make_float4(sw[0], sw[1], sw[2], 0.f);
                        continue;
                    }
#pragma unroll
                    for (int b = 0; b < 3; b++)
                        if (b < a.f_active) a.f_out[b][p] = sw[b] > 0.f ? acc[b] / sw[b] : a.f_colour[b][p];
                }
            }
        }
    }
}

// The kernel: regular tiles over columns [rx0, rx_split), DUAL tiles over [rx_split, rx1), one grid.
// The DUAL items come last in item order, i.e. they are dispatched last and fill the final round.
// XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB
// L2), so workgroup b is given work item u such that every XCD walks one contiguous range of items:
// the parts of a tile and vertically adjacent tiles -- which stage the same image rows -- then hit in
// the same L2 instead of each fetching its own copy over the fabric.  Placement only affects speed;
// the remap is a bijection for any grid size.
template <int RT, int K, int SPEC>
__global__ __launch_bounds__(kThreads, 2) void window_filter_lds(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n_items = gridDim.x, b = blockIdx.x;
    const int q8 = n_items >> 3, rem8 = n_items & 7, xcd = b & 7, idx = b >> 3;
    const int u = xcd < rem8 ? xcd * (q8 + 1) + idx : rem8 * (q8 + 1) + (xcd - rem8) * q8 + idx;
    if (u < a.n_main_items) filter_tile<RT, K, SPEC, false>(a, lds, u, a.rx0, a.rx_split);
    else filter_tile<RT, K, SPEC, true>(a, lds, u - a.n_main_items, a.rx_split, a.rx1);
}

// Sums the per-part partial (acc, sum_w) of every ROI pixel in part order and normalises.
template <bool RGB>
__global__ __launch_bounds__(256) void combine_parts_kernel(FilterArgs a) {
    const int x = a.rx0 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.ry0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.rx1 || y >= a.ry1) return;
    const long long p = (long long)y * a.width + x, plane = (long long)a.width * a.height;
    if constexpr (RGB) {
        float4 t = reinterpret_cast<const float4 *>(a.partial)[p];
        for (int k = 1; k < a.n_parts; k++) {
            const float4 u = reinterpret_cast<const float4 *>(a.partial)[k * plane + p];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        f3 o;
        if (t.w > 0.f) {
            o.x = t.x / t.w; o.y = t.y / t.w; o.z = t.z / t.w;
        } else {
            o = a.packed ? reinterpret_cast<const f3 *>(a.packed + p * 15)[2] : reinterpret_cast<const f3 *>(a.colour)[p];
        }
        reinterpret_cast<f3 *>(a.out)[p] = o;
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(a.partial);
        float4 acc = src[2 * p], sw = src[2 * p + 1];
        for (int k = 1; k < a.n_parts; k++) {
            const float4 u = src[2 * (k * plane + p)], v = src[2 * (k * plane + p) + 1];
            acc.x += u.x; acc.y += u.y; acc.z += u.z;
            sw.x += v.x; sw.y += v.y; sw.z += v.z;
        }
        const float accs[3] = {acc.x, acc.y, acc.z}, sws[3] = {sw.x, sw.y, sw.z};
#pragma unroll
        for (int b = 0; b < 3; b++)
            if (b < a.f_active) a.f_out[b][p] = sws[b] > 0.f ? accs[b] / sws[b] : a.f_colour[b][p];
    }
}

// Owned block of the five filter inputs (+ sample counts / + two 1-channel G-buffers / + both) -> 15 / 16 / 17 / 18-channel block + halo image (one pass).
__global__ __launch_bounds__(256) void pack_inputs_kernel(PackArgs a) {
    const long long n = (long long)a.src_w * a.src_h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / a.src_w), x = (int)(i - (long long)y * a.src_w);
        float *px = a.packed + ((long long)(y + a.dst_y0) * a.dst_w + (x + a.dst_x0)) * a.ch;
        f3 *dst = reinterpret_cast<f3 *>(px);
        dst[0] = reinterpret_cast<const f3 *>(a.mean_corr)[i];
        dst[1] = reinterpret_cast<const f3 *>(a.disc)[i];
        dst[2] = reinterpret_cast<const f3 *>(a.colour)[i];
        dst[3] = a.g0 ? reinterpret_cast<const f3 *>(a.g0)[i] : f3{0.f, 0.f, 0.f};
        dst[4] = a.g1 ? reinterpret_cast<const f3 *>(a.g1)[i] : f3{0.f, 0.f, 0.f};
        if (a.ch >= 17) {
            px[15] = a.s0 ? a.s0[i] : 0.f;
            px[16] = a.s1 ? a.s1[i] : 0.f;
            if (a.ch == 18) px[17] = __int_as_float(a.n[i]);
        } else if (a.ch == 16) {
            px[15] = __int_as_float(a.n[i]);
        }
    }
}

hipError_t launch_pack_inputs(const PackArgs &a, hipStream_t s) {
    const long long n = (long long)a.src_w * a.src_h;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// The LDS kernel covers T = float3 (one RGB buffer) and T = float (three buffers per launch)
// under G-buffers of 1 or 3 channels whose channel counts add up to at most six (the kernel's feature
// slots; written for two RGB images, an absent one is a slot with factor 0 that is never read), finite
// non-positive DR factors and radius 1..20.
bool fast_path_eligible(const FilterArgs &a, int channels) {
    if (channels != 3 && channels != 1) return false;
    // the LDS kernels know both gates, both channel rules and both border rules; a per-pair Welch lookup runs the
    // general kernel
    if (a.dof != STATMC_DOF_PIXEL) return false;
    int slots = 0;
    for (int g = 0; g < a.n_g; g++) {
        if (a.g[g].channels != 1 && a.g[g].channels != 3) return false;
        if (!(a.g[g].dr <= 0.f) || !isfinite(a.g[g].dr)) return false;
        slots += a.g[g].channels;
    }
    if (slots > 6) return false;
    if (a.radius < 1 || a.radius > kMaxR) return false;
    return true;
}

// Feature layout of the LDS kernel for an eligible G-buffer set: up to two RGB images keep the vector
// loads (k0, k1 per image); anything else is spread over the six slots, one channel each.
void set_feature_layout(FilterArgs &a) {
    constexpr float kL2e = 1.44269504088896340736f;
    bool rgb_pair = a.n_g <= 2;
    for (int g = 0; g < a.n_g; g++) rgb_pair = rgb_pair && a.g[g].channels == 3;
    for (int f = 0; f < 6; f++) a.feat[f] = FilterArgs::FeatSlot{nullptr, 0, 0, 0.f};
    if (rgb_pair) {
        a.feat_generic = 0;
        a.gscale0 = a.n_g > 0 ? sqrtf(-a.g[0].dr * kL2e) : 0.f;
        a.gscale1 = a.n_g > 1 ? sqrtf(-a.g[1].dr * kL2e) : 0.f;
        return;
    }
    a.feat_generic = 1;
    a.gscale0 = a.gscale1 = 1.f;
    int f = 0;
    for (int g = 0; g < a.n_g; g++)
        for (int c = 0; c < a.g[g].channels; c++, f++)
            a.feat[f] = FilterArgs::FeatSlot{a.g[g].data, a.g[g].channels, c, sqrtf(-a.g[g].dr * kL2e)};
}

// Number of window-sweep parts per tile: the grid runs one workgroup per CU (LDS-bound), so its
// makespan is ceil(tiles*parts / CUs) rounds of 1/parts tile-time each (+ a prologue per part
// that stages 8 rows without compute to overlap, measured at about one window row of time).
// Pick the part count with the smallest estimate.
int choose_parts(int tiles, int n_rows, int n_cus) {
    int best = 1;
    double best_cost = 1e30;
    for (int k = 1; k <= 8 && k <= n_rows; k++) {
        const long long wgs = (long long)tiles * k;
        const double rounds = (double)((wgs + n_cus - 1) / n_cus);
        const double cost = rounds * ((double)n_rows / k + 1.0);
        if (cost < best_cost * 0.995) {
            best_cost = cost;
            best = k;
        }
    }
    return best;
}

bool sym_path_selected(const FilterArgs &a, int channels) {
    return sym_eligible(a, channels) && a.spatial_tab != nullptr && a.force_variant == 0;
}
int sym_filter_parts(const FilterArgs &a, int n_cus) {
    const int forced = a.force_parts;
    if (forced > 0) return forced < a.radius + 1 ? forced : a.radius + 1;
    return sym_choose_parts(sym_tiles(a), n_cus, a.radius + 1);
}

// Where the ROI is cut: columns [rx0, split) go to regular 256-wide tiles, [split, rx1) -- at most half
// a tile -- to the DUAL tiles (split == rx1: no DUAL column).
static int dual_split(const FilterArgs &a) {
    const int rem = (a.rx1 - a.rx0) % kTileW;
    return rem > 0 && rem <= Geo<true>::W ? a.rx1 - rem : a.rx1;
}
static int lds_tiles(const FilterArgs &a) {
    const int split = dual_split(a), h = a.ry1 - a.ry0;
    int tiles = ((split - a.rx0 + kTileW - 1) / kTileW) * ((h + Geo<false>::ROWS - 1) / Geo<false>::ROWS);
    if (split < a.rx1) tiles += (h + Geo<true>::ROWS - 1) / Geo<true>::ROWS;
    return tiles;
}

template <bool DUAL>
static size_t lds_bytes_for(int rp) {
    using G = Geo<DUAL>;
    return ((size_t)G::SLOTS * kCh * (G::W + 2 * rp) + 4 * (2 * rp + 8)) * sizeof(float);
}

// The 160 KB dynamic-LDS attribute is a property of (kernel, device): set once for each pair.
static hipError_t allow_full_lds(const void *kernel) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess)
        return e;
    done.insert({dev, kernel});
    return hipSuccess;
}

template <int RT, int K, int SPEC = 0>
static hipError_t launch_lds(FilterArgs a, hipStream_t s) {
    if (a.partial == nullptr) a.n_parts = 1;
    if (a.rx1 <= a.rx0 || a.ry1 <= a.ry0) return hipSuccess;
    const int rp = RT > 0 ? round_up4(RT) : round_up4(a.radius);
    a.rx_split = dual_split(a);
    const int h = a.ry1 - a.ry0;
    const int main_tiles = ((a.rx_split - a.rx0 + kTileW - 1) / kTileW) * ((h + Geo<false>::ROWS - 1) / Geo<false>::ROWS);
    const int dual_tiles = a.rx_split < a.rx1 ? (h + Geo<true>::ROWS - 1) / Geo<true>::ROWS : 0;
    a.n_main_items = main_tiles * a.n_parts;
    const size_t lds_bytes = std::max(main_tiles ? lds_bytes_for<false>(rp) : 0, dual_tiles ? lds_bytes_for<true>(rp) : 0);
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void *>(&window_filter_lds<RT, K, SPEC>)); e != hipSuccess) return e;
    hipLaunchKernelGGL((window_filter_lds<RT, K, SPEC>), dim3((main_tiles + dual_tiles) * a.n_parts), dim3(kThreads), lds_bytes, s, a);
    if (a.n_parts > 1) {
        const dim3 cgrid((a.rx1 - a.rx0 + 63) / 64, (a.ry1 - a.ry0 + 3) / 4);
        hipLaunchKernelGGL(combine_parts_kernel<K == 0>, cgrid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

int lds_filter_parts(const FilterArgs &a, int n_cus) {
    const int forced = a.force_parts;
    if (forced > 0) return forced < 2 * a.radius + 1 ? forced : 2 * a.radius + 1;
    return choose_parts(lds_tiles(a), 2 * a.radius + 1, n_cus);
}

// True when launch_window_filter will run the LDS kernel for these arguments (the C-ABI layer
// groups float buffers three per launch only then).
bool lds_path_selected(const FilterArgs &a, int channels) {
    return fast_path_eligible(a, channels) && a.spatial_tab != nullptr && a.force_variant != 1;
}

// "sym_r20|sym_rt[_f][_g8][_asym][_joint][_clamp]": compile-time / runtime radius, float buffers, eight feature planes, then the spec's non-default choices
static const char *sym_variant_name(const FilterArgs &a, int channels) {
    static thread_local char name[64];
    const bool joint = a.channel_rule == STATMC_CHANNELS_JOINT && channels == 3;
    const bool welch = a.dof == STATMC_DOF_WELCH;   // (one build for every radius; the gate field has no meaning under Welch)
    snprintf(name, sizeof(name), "%s%s%s%s%s%s", welch ? "sym_welch" : a.radius == 20 ? "sym_r20" : "sym_rt", channels == 1 ? "_f" : "", a.sym.g8 ? "_g8" : "",
             welch ? "" : a.gate == STATMC_GATE_ASYMMETRIC ? "_asym" : a.gate == STATMC_GATE_CENTRE ? "_centre" : "", joint ? "_joint" : "",
             a.border == STATMC_BORDER_CLAMP ? "_clamp" : "");
    return name;
}

// Launch of the one-sided LDS kernel for the arguments' spec.  The default membership test has a compile-time-radius
// build for r = 20; the other tests (one-sided gate, pooled channels) run the runtime-radius build.
static int lds_spec_of(const FilterArgs &a, bool rgb) {
    // (STATMC_GATE_CENTRE runs the one-sided-gate build on rows staged with -D_q = 0)
    return (a.gate != STATMC_GATE_SYMMETRIC ? kSpecAsym : 0) | (rgb && a.channel_rule == STATMC_CHANNELS_JOINT ? kSpecJoint : 0);
}

template <int K>
static hipError_t launch_lds_spec(const FilterArgs &a, hipStream_t s, const char **variant) {
    constexpr bool rgb = K == 0;
    switch (lds_spec_of(a, rgb)) {
    case 0:
        if (a.radius == 20 && a.force_variant != 2) {
            *variant = rgb ? "lds_r20" : "lds_r20_f";
            return launch_lds<20, K>(a, s);
        }
        *variant = rgb ? "lds_rt" : "lds_rt_f";
        return launch_lds<0, K>(a, s);
    case kSpecAsym:
        *variant = a.gate == STATMC_GATE_CENTRE ? (rgb ? "lds_rt_centre" : "lds_rt_f_centre") : (rgb ? "lds_rt_asym" : "lds_rt_f_asym");
        return launch_lds<0, K, kSpecAsym>(a, s);
    case kSpecJoint:
        *variant = "lds_rt_joint";
        return launch_lds<0, K, rgb ? kSpecJoint : 0>(a, s);
    default:
        *variant = a.gate == STATMC_GATE_CENTRE ? "lds_rt_centre_joint" : "lds_rt_asym_joint";
        return launch_lds<0, K, rgb ? (kSpecAsym | kSpecJoint) : kSpecAsym>(a, s);
    }
}

// Window filter reading the 15-channel block + halo image (multi-GPU path): LDS kernels only.
hipError_t launch_lds_packed(const FilterArgs &a, hipStream_t s, const char **variant) {
    if (a.sym.patch != nullptr) {
        *variant = sym_variant_name(a, 3);
        return launch_sym(a, s);
    }
    return launch_lds_spec<0>(a, s, variant);
}

hipError_t launch_window_filter(const FilterArgs &a, int channels, hipStream_t s, const char **variant) {
    if (a.sym.patch != nullptr) {   // the C-ABI layer prepared the pair-symmetric kernel's workspace: that kernel was chosen
        *variant = sym_variant_name(a, channels);
        return launch_sym(a, s);
    }
    const bool fast = lds_path_selected(a, channels);
    if (fast) {
        if (channels == 3) return launch_lds_spec<0>(a, s, variant);
        return a.f_active >= 3 ? launch_lds_spec<3>(a, s, variant) : a.f_active == 2 ? launch_lds_spec<2>(a, s, variant) : launch_lds_spec<1>(a, s, variant);
    }
    *variant = "generic";
    const dim3 grid((a.rx1 - a.rx0 + 31) / 32, (a.ry1 - a.ry0 + 7) / 8);
    if (channels == 3)
        hipLaunchKernelGGL(window_filter_generic<3>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(window_filter_generic<1>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace statmc
