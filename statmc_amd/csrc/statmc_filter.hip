// statmc_filter.hip -- the statistics-gated cross-bilateral window filter (gfx950).
//
// Replaces the window part of cv::cuda::stat_denoiser::filter<T> (call sites
// src/statistics/estimator.cpp:437-487 of the reference; its CUDA source is not in the tree,
// the arithmetic is this build's own frozen spec: DESIGN.md "Filter spec v1", restated on the
// CPU in oracle/statmc_oracle.c:oracle_filter).
//
// Two implementations:
//
//  * window_filter_lds<R>  -- the hot kernel.  T = float3 with two 3-channel G-buffers (the
//    shipped configuration: radiance filtered under normal + albedo), radius <= 20.
//    This is a stencil, not a contraction: 1681 taps x ~28 fp32 VALU ops per pixel against
//    72 B of compulsory HBM traffic, so the design goal is VALU issue rate, with LDS as the
//    operand feed:
//      - a 512-thread workgroup (8 waves = 2 per SIMD, which is what a gfx950 SIMD needs to
//        issue a VALU op every 2 cycles) owns a 256 x 8 output tile; each wave owns one row,
//        each lane 4 adjacent pixels of it;
//      - the 15 per-pixel floats a tap needs (corrected mean 3, -discriminator 3, scaled
//        normal 3, scaled albedo 3, colour 3) are staged as SoA planes in LDS, one image row
//        at a time, in a ring of 9 rows (8 live + 1 being filled) -- the full (256+40) x
//        (8+40) halo would need 850 KB, the ring needs 156 KB of the CU's 160 KB;
//      - per window row a lane reads its 44-column span with 11 x 15 ds_read_b128
//        (conflict-free: consecutive lanes read consecutive 16 B) and evaluates 4 taps x 4
//        pixels per read group in registers, so every LDS value is used 4 times;
//      - G-buffers are pre-multiplied by sqrt(-DR_g * log2 e) when staged and the spatial
//        term comes from a per-(dy,dx) table read through the scalar cache, so the range
//        weight is 6 x (sub, fma) + one v_exp_f32;
//      - membership is 3 x (sub, fma, cmp) + one cndmask: fma(d, d, -D_q) <= D_p, the same
//        expression, bit for bit, as the oracle.
//    Taps outside the image are staged with a NaN corrected mean, which fails every
//    comparison, so clipping costs nothing in the inner loop.
//
//  * window_filter_generic<C> -- any radius, any G-buffer set, T = float or float3, one lane
//    per pixel straight from global memory.  Correctness path for configurations the hot
//    kernel does not cover (float G-buffers, radius > 20, filter<float>).

#include <math.h>

#include <utility>

#include "statmc_device.h"

namespace statmc {

static int g_variant_override = 0;
void set_filter_variant_override(int v) { g_variant_override = v; }

constexpr float kLog2e = 1.44269504088896340736f;

// ====================================================================== generic kernel
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
    for (int dy = -r; dy <= r; dy++) {
        const int qy = y + dy;
        if (qy < 0 || qy >= a.height) continue;
        for (int dx = -r; dx <= r; dx++) {
            const int qx = x + dx;
            if (qx < 0 || qx >= a.width) continue;
            const long long q = (long long)qy * a.width + qx;
            bool member = true;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float d = pc[c] - a.mean_corr[q * C + c];
                member = member & (__builtin_fmaf(d, d, -a.disc[q * C + c]) <= pd[c]);
            }
            if (!member) continue;
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

// ====================================================================== LDS kernel
constexpr int kPx = 4;                 // pixels per lane
constexpr int kTileW = 64 * kPx;       // 256 columns per wave-row
constexpr int kTileH = 8;              // rows per tile = waves per workgroup
constexpr int kThreads = 64 * kTileH;  // 512
constexpr int kSlots = kTileH + 1;     // LDS row ring
constexpr int kCh = 15;                // floats staged per pixel
constexpr int kMaxR = 20;

// plane order inside an LDS row
enum { P_MC = 0, P_ND = 3, P_G0 = 6, P_G1 = 9, P_COL = 12 };

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

struct f3 {
    float x, y, z;
};

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

struct LaneState {
    float pc[kPx][3], pd[kPx][3], pg0[kPx][3], pg1[kPx][3];
    float acc[kPx][3], sw[kPx];
};

// One read group: 4 taps (columns 4*lane + 4*j .. +3 of the staged row) against the lane's
// 4 pixels.  `tabrow` points at the spatial exponents of this window row; MASK selects the
// (tap, pixel) pairs that lie inside the window (all 16 in the runtime-radius variant, where
// the table holds -inf outside the radius).
template <unsigned MASK>
__device__ __forceinline__ void eval_chunk(LaneState &st, const float *__restrict__ row, int pitch, int j,
                                           const float *__restrict__ tabrow) {
    float4 q[kCh];
#pragma unroll
    for (int ch = 0; ch < kCh; ch++) q[ch] = *reinterpret_cast<const float4 *>(row + ch * pitch + 4 * j);
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int k = 0; k < kPx; k++) {
            if (!(MASK & (1u << (i * 4 + k)))) continue;
            float e = tabrow[4 * j + i - k + 3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float d = st.pg0[k][c] - (&q[P_G0 + c].x)[i];
                e = __builtin_fmaf(-d, d, e);
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float d = st.pg1[k][c] - (&q[P_G1 + c].x)[i];
                e = __builtin_fmaf(-d, d, e);
            }
            float w = __builtin_amdgcn_exp2f(e);
            bool member = true;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float d = st.pc[k][c] - (&q[P_MC + c].x)[i];
                member = member & (__builtin_fmaf(d, d, (&q[P_ND + c].x)[i]) <= st.pd[k][c]);
            }
            w = member ? w : 0.f;
            st.sw[k] += w;
#pragma unroll
            for (int c = 0; c < 3; c++) st.acc[k][c] = __builtin_fmaf(w, (&q[P_COL + c].x)[i], st.acc[k][c]);
        }
    }
}

// Compile-time radius: the read groups at the two ends of the span have (tap, pixel) pairs
// outside the window and are peeled with their static masks; the groups in between are all
// inside and run as a rolled loop (keeps the live set to one read group: 60 VGPRs of taps).
template <int RT>
struct StaticSpan {
    static constexpr int n_chunks = 2 * round_up4(RT) / 4 + 1;
    template <int J>
    static constexpr bool full() { return ChunkMask<J, RT>::value() == 0xFFFFu; }
};

template <int RT, int J, int JEND>
__device__ __forceinline__ void eval_peeled(LaneState &st, const float *row, int pitch, const float *tabrow) {
    if constexpr (J < JEND) {
        eval_chunk<ChunkMask<J, RT>::value()>(st, row, pitch, J, tabrow);
        eval_peeled<RT, J + 1, JEND>(st, row, pitch, tabrow);
    }
}

template <int RT, int J = 0>
constexpr int first_full() {
    if constexpr (J >= StaticSpan<RT>::n_chunks) return J;
    else if constexpr (StaticSpan<RT>::template full<J>()) return J;
    else return first_full<RT, J + 1>();
}
template <int RT, int J>
constexpr int end_full() {  // first non-full group at or after J
    if constexpr (J >= StaticSpan<RT>::n_chunks) return J;
    else if constexpr (!StaticSpan<RT>::template full<J>()) return J;
    else return end_full<RT, J + 1>();
}

template <int RT>
__device__ __forceinline__ void eval_row_static(LaneState &st, const float *row, int pitch, const float *tabrow) {
    constexpr int f0 = first_full<RT>();
    constexpr int f1 = end_full<RT, f0>();
    eval_peeled<RT, 0, f0>(st, row, pitch, tabrow);
#pragma unroll 1
    for (int j = f0; j < f1; j++) eval_chunk<0xFFFFu>(st, row, pitch, j, tabrow);
    eval_peeled<RT, f1, StaticSpan<RT>::n_chunks>(st, row, pitch, tabrow);
}

// Stage one image row (image row yrow, columns x0-rp .. x0-rp+pitch) into an LDS ring slot.
struct StagedPixel {
    f3 mc, d, g0, g1, col;
    bool valid;
};

__device__ __forceinline__ StagedPixel load_pixel(const FilterArgs &a, int x, int yrow) {
    StagedPixel s;
    s.valid = x >= 0 && x < a.width && yrow >= 0 && yrow < a.height;
    if (s.valid) {
        const long long q = (long long)yrow * a.width + x;
        s.mc = reinterpret_cast<const f3 *>(a.mean_corr)[q];
        s.d = reinterpret_cast<const f3 *>(a.disc)[q];
        s.g0 = reinterpret_cast<const f3 *>(a.g[0].data)[q];
        s.g1 = reinterpret_cast<const f3 *>(a.g[1].data)[q];
        s.col = reinterpret_cast<const f3 *>(a.colour)[q];
    }
    return s;
}

__device__ __forceinline__ void store_pixel(float *slot, int pitch, int i, const StagedPixel &s, float k0, float k1) {
    const float nan = __builtin_nanf("");
    slot[(P_MC + 0) * pitch + i] = s.valid ? s.mc.x : nan;
    slot[(P_MC + 1) * pitch + i] = s.valid ? s.mc.y : nan;
    slot[(P_MC + 2) * pitch + i] = s.valid ? s.mc.z : nan;
    slot[(P_ND + 0) * pitch + i] = s.valid ? -s.d.x : 0.f;
    slot[(P_ND + 1) * pitch + i] = s.valid ? -s.d.y : 0.f;
    slot[(P_ND + 2) * pitch + i] = s.valid ? -s.d.z : 0.f;
    slot[(P_G0 + 0) * pitch + i] = s.valid ? s.g0.x * k0 : 0.f;
    slot[(P_G0 + 1) * pitch + i] = s.valid ? s.g0.y * k0 : 0.f;
    slot[(P_G0 + 2) * pitch + i] = s.valid ? s.g0.z * k0 : 0.f;
    slot[(P_G1 + 0) * pitch + i] = s.valid ? s.g1.x * k1 : 0.f;
    slot[(P_G1 + 1) * pitch + i] = s.valid ? s.g1.y * k1 : 0.f;
    slot[(P_G1 + 2) * pitch + i] = s.valid ? s.g1.z * k1 : 0.f;
    slot[(P_COL + 0) * pitch + i] = s.valid ? s.col.x : 0.f;
    slot[(P_COL + 1) * pitch + i] = s.valid ? s.col.y : 0.f;
    slot[(P_COL + 2) * pitch + i] = s.valid ? s.col.z : 0.f;
}

// RT > 0: compile-time radius (window edges resolved statically); RT == 0: runtime radius
// a.radius <= 20, every pair of every read group evaluated, the table masks taps beyond r.
template <int RT>
__global__ __launch_bounds__(kThreads, 2) void window_filter_lds(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int r = RT > 0 ? RT : a.radius;
    const int rp = RT > 0 ? round_up4(RT) : round_up4(a.radius);
    const int pitch = kTileW + 2 * rp;
    const int n_chunks = 2 * rp / 4 + 1;
    const int tw = tab_width(rp);
    const int slot_floats = kCh * pitch;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x0 = a.rx0 + blockIdx.x * kTileW;
    const int y0 = a.ry0 + blockIdx.y * kTileH;
    const float k0 = a.gscale0, k1 = a.gscale1;

    // ---- the lane's own 4 pixels (clamped into the image so the loads stay in bounds)
    LaneState st;
    const int py = min(y0 + wave, a.height - 1);
#pragma unroll
    for (int k = 0; k < kPx; k++) {
        const int px = min(x0 + kPx * lane + k, a.width - 1);
        const long long p = (long long)py * a.width + px;
        const f3 mc = reinterpret_cast<const f3 *>(a.mean_corr)[p];
        const f3 d = reinterpret_cast<const f3 *>(a.disc)[p];
        const f3 g0 = reinterpret_cast<const f3 *>(a.g[0].data)[p];
        const f3 g1 = reinterpret_cast<const f3 *>(a.g[1].data)[p];
        st.pc[k][0] = mc.x; st.pc[k][1] = mc.y; st.pc[k][2] = mc.z;
        st.pd[k][0] = d.x; st.pd[k][1] = d.y; st.pd[k][2] = d.z;
        st.pg0[k][0] = g0.x * k0; st.pg0[k][1] = g0.y * k0; st.pg0[k][2] = g0.z * k0;
        st.pg1[k][0] = g1.x * k1; st.pg1[k][1] = g1.y * k1; st.pg1[k][2] = g1.z * k1;
        st.sw[k] = 0.f;
        st.acc[k][0] = st.acc[k][1] = st.acc[k][2] = 0.f;
    }

    // ---- prologue: window rows rel = 0 .. kTileH-1 (image rows y0-r+rel) into slots 0..7
    for (int idx = threadIdx.x; idx < kTileH * pitch; idx += kThreads) {
        const int rel = idx / pitch, i = idx - rel * pitch;
        const StagedPixel s = load_pixel(a, x0 - rp + i, y0 - r + rel);
        store_pixel(lds + rel * slot_floats, pitch, i, s, k0, k1);
    }
    __syncthreads();

    // ---- sweep the 2r+1 window rows; wave w works on window row rel = w + step
    int slot = wave;  // (wave + step) % kSlots
    int fill = kTileH;  // (step + kTileH) % kSlots: the slot the next row is staged into
    for (int step = 0; step <= 2 * r; step++) {
        // issue the global loads of the row needed by the next step early
        const bool stage = step < 2 * r && (int)threadIdx.x < pitch;
        StagedPixel nxt;
        nxt.valid = false;
        if (stage) nxt = load_pixel(a, x0 - rp + (int)threadIdx.x, y0 - r + step + kTileH);

        const float *row = lds + slot * slot_floats + kPx * lane;
        const float *tabrow = a.spatial_tab + step * tw;
        if constexpr (RT > 0) {
            eval_row_static<RT>(st, row, pitch, tabrow);
        } else {
#pragma unroll 1
            for (int j = 0; j < n_chunks; j++) eval_chunk<0xFFFFu>(st, row, pitch, j, tabrow);
        }

        if (stage) store_pixel(lds + fill * slot_floats, pitch, threadIdx.x, nxt, k0, k1);
        __syncthreads();
        slot = slot + 1 == kSlots ? 0 : slot + 1;
        fill = fill + 1 == kSlots ? 0 : fill + 1;
    }

    // ---- epilogue
    const int oy = y0 + wave;
    if (oy < a.ry1) {
#pragma unroll
        for (int k = 0; k < kPx; k++) {
            const int ox = x0 + kPx * lane + k;
            if (ox < a.rx1) {
                const long long p = (long long)oy * a.width + ox;
                f3 o;
                if (st.sw[k] > 0.f) {
                    o.x = st.acc[k][0] / st.sw[k];
                    o.y = st.acc[k][1] / st.sw[k];
                    o.z = st.acc[k][2] / st.sw[k];
                } else {
                    o = reinterpret_cast<const f3 *>(a.colour)[p];
                }
                reinterpret_cast<f3 *>(a.out)[p] = o;
            }
        }
    }
}

bool fast_path_eligible(const FilterArgs &a, int channels) {
    if (channels != 3 || a.n_g != 2) return false;
    if (a.g[0].channels != 3 || a.g[1].channels != 3) return false;
    if (a.radius < 1 || a.radius > kMaxR) return false;
    if (!(a.g[0].dr <= 0.f) || !(a.g[1].dr <= 0.f)) return false;
    if (!isfinite(a.g[0].dr) || !isfinite(a.g[1].dr)) return false;
    return true;
}

template <int RT>
static hipError_t launch_lds(const FilterArgs &a, hipStream_t s) {
    const int rp = RT > 0 ? round_up4(RT) : round_up4(a.radius);
    const size_t lds_bytes = (size_t)kSlots * kCh * (kTileW + 2 * rp) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&window_filter_lds<RT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const dim3 grid((a.rx1 - a.rx0 + kTileW - 1) / kTileW, (a.ry1 - a.ry0 + kTileH - 1) / kTileH);
    hipLaunchKernelGGL(window_filter_lds<RT>, grid, dim3(kThreads), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_window_filter(const FilterArgs &a, int channels, hipStream_t s, const char **variant) {
    const bool fast = fast_path_eligible(a, channels) && a.spatial_tab != nullptr && g_variant_override != 1;
    if (fast) {
        if (a.radius == 20 && g_variant_override != 2) {
            *variant = "lds_r20";
            return launch_lds<20>(a, s);
        }
        *variant = "lds_rt";
        return launch_lds<0>(a, s);
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
