// statmc_filter_common.h -- device helpers shared by the window-filter translation units
// (statmc_filter.hip: general + row-per-wave LDS kernels; statmc_filter_sym.hip: pair-symmetric LDS kernel):
// the 15-float staging of a pixel, which pixels take part in windows (spec v2), the LDS channel planes.
#pragma once

#include "statmc_device.h"

namespace statmc {

constexpr float kLog2e = 1.44269504088896340736f;

// LDS row layout of the LDS kernels: 15 channel planes of `pitch` floats.
enum { C_G0 = 0, C_G1 = 3, C_MC = 6, C_ND = 9, C_COL = 12 };


typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct f3 {
    float x, y, z;
};


// One pixel as the LDS kernels stage it (image row yrow, column x).
struct StagedPixel {
    f3 mc, d, g0, g1, col;
    bool valid;
};

// The six feature values of pixel q (before the k0 / k1 scaling of the two-RGB-buffer layout; already
// scaled, with k0 = k1 = 1, in the generic slot layout).  A slot or buffer with factor 0 is never read.
__device__ __forceinline__ void load_features(const FilterArgs &a, long long q, f3 &g0, f3 &g1) {
    if (a.feat_generic) {
        float v[6];
#pragma unroll
        for (int f = 0; f < 6; f++)
            v[f] = a.feat[f].scale != 0.f ? a.feat[f].data[q * a.feat[f].stride + a.feat[f].offset] * a.feat[f].scale : 0.f;
        g0 = f3{v[0], v[1], v[2]};
        g1 = f3{v[3], v[4], v[5]};
        return;
    }
    g0 = a.gscale0 != 0.f ? reinterpret_cast<const f3 *>(a.g[0].data)[q] : f3{0.f, 0.f, 0.f};
    g1 = a.gscale1 != 0.f ? reinterpret_cast<const f3 *>(a.g[1].data)[q] : f3{0.f, 0.f, 0.f};
}

template <bool RGB>
__device__ __forceinline__ StagedPixel load_pixel(const FilterArgs &a, int x, int yrow) {
    StagedPixel s;
    if (a.border == STATMC_BORDER_CLAMP) {  // taps beyond the image repeat its edge pixels
        x = min(max(x, 0), a.width - 1);
        yrow = min(max(yrow, 0), a.height - 1);
    }
    s.valid = x >= 0 && x < a.width && yrow >= 0 && yrow < a.height;
    if (s.valid) {
        const long long q = (long long)yrow * a.width + x;
        if (RGB && a.packed) {  // block + halo image of the multi-GPU path: 15 contiguous floats per pixel
            const f3 *px = reinterpret_cast<const f3 *>(a.packed + q * 15);
            s.mc = px[0]; s.d = px[1]; s.col = px[2]; s.g0 = px[3]; s.g1 = px[4];
            return s;
        }
        if constexpr (RGB) {
            s.mc = reinterpret_cast<const f3 *>(a.mean_corr)[q];
            s.d = reinterpret_cast<const f3 *>(a.disc)[q];
            s.col = reinterpret_cast<const f3 *>(a.colour)[q];
        } else {
            s.mc = f3{a.f_mean_corr[0][q], a.f_mean_corr[1][q], a.f_mean_corr[2][q]};
            s.d = f3{a.f_disc[0][q], a.f_disc[1][q], a.f_disc[2][q]};
            s.col = f3{a.f_colour[0][q], a.f_colour[1][q], a.f_colour[2][q]};
        }
        load_features(a, q, s.g0, s.g1);
    }
    return s;
}

// Which pixels take part in windows (spec v2, oracle pixel_valid): corrected mean finite, discriminator
// not NaN, colour finite -- per pixel for an RGB buffer (all three channels), per buffer in float mode
// (three independent 1-channel buffers); pixels outside the image never do.  vx/vy/vz: the verdict per
// channel (RGB: all three equal).
struct Validity {
    bool x, y, z;
};
__device__ __forceinline__ Validity pixel_validity(const f3 &mc, const f3 &d, const f3 &col, bool in_image, bool rgb) {
    const bool vx = in_image && __builtin_isfinite(mc.x) && d.x == d.x && __builtin_isfinite(col.x);
    const bool vy = in_image && __builtin_isfinite(mc.y) && d.y == d.y && __builtin_isfinite(col.y);
    const bool vz = in_image && __builtin_isfinite(mc.z) && d.z == d.z && __builtin_isfinite(col.z);
    if (rgb) {
        const bool v = vx && vy && vz;
        return Validity{v, v, v};
    }
    return Validity{vx, vy, vz};
}
// Spec v2.1 (round 4): a pixel with a non-finite G-buffer value takes no part either -- its range weights would be NaN for
// every pair (and, in the runtime-radius builds of the pair-symmetric kernel, for taps just beyond a small radius, whose
// exponent is -inf only as long as the feature term is a number).  Callers pass `in_image && features_finite(...)` where
// pixel_validity asks whether the pixel exists, and stage such a pixel's features as 0.
__device__ __forceinline__ bool features_finite(const f3 &g0, const f3 &g1, float s0 = 0.f, float s1 = 0.f) {
    return __builtin_isfinite(g0.x) && __builtin_isfinite(g0.y) && __builtin_isfinite(g0.z) && __builtin_isfinite(g1.x) &&
           __builtin_isfinite(g1.y) && __builtin_isfinite(g1.z) && __builtin_isfinite(s0) && __builtin_isfinite(s1);
}
// Staging rule for the corrected mean: it is what switches a tap off in the inner loop -- NaN there
// fails every comparison (and v_max3 drops a NaN in a single channel, hence all three for RGB).
__device__ __forceinline__ f3 canonical_mean(const f3 &mc, const Validity &v) {
    const float nan = __builtin_nanf("");
    return f3{v.x ? mc.x : nan, v.y ? mc.y : nan, v.z ? mc.z : nan};
}

// zero_nd (one-sided kernel under STATMC_GATE_CENTRE): the staged -D_q is 0, so that the one-sided test
// fma(d, d, -D_q) <= D_p of that kernel reads d * d <= D_p -- the tap's own interval plays no part
__device__ __forceinline__ void store_pixel(float *slot, int pitch, int i, const StagedPixel &s, float k0, float k1,
                                            bool rgb, bool zero_nd = false) {
    const bool v = s.valid && features_finite(s.g0, s.g1);
    const Validity ok = pixel_validity(s.mc, s.d, s.col, v, rgb);
    const f3 mc = canonical_mean(s.mc, ok);
    float *p = slot + i;
    p[(C_G0 + 0) * pitch] = v ? s.g0.x * k0 : 0.f;
    p[(C_G0 + 1) * pitch] = v ? s.g0.y * k0 : 0.f;
    p[(C_G0 + 2) * pitch] = v ? s.g0.z * k0 : 0.f;
    p[(C_G1 + 0) * pitch] = v ? s.g1.x * k1 : 0.f;
    p[(C_G1 + 1) * pitch] = v ? s.g1.y * k1 : 0.f;
    p[(C_G1 + 2) * pitch] = v ? s.g1.z * k1 : 0.f;
    p[(C_MC + 0) * pitch] = mc.x;
    p[(C_MC + 1) * pitch] = mc.y;
    p[(C_MC + 2) * pitch] = mc.z;
    p[(C_ND + 0) * pitch] = ok.x && !zero_nd ? -s.d.x : 0.f;
    p[(C_ND + 1) * pitch] = ok.y && !zero_nd ? -s.d.y : 0.f;
    p[(C_ND + 2) * pitch] = ok.z && !zero_nd ? -s.d.z : 0.f;
    // the colour of a pixel that takes no part is staged as 0: its weight is 0, and 0 * NaN would
    // otherwise poison the sums of every window that covers it
    p[(C_COL + 0) * pitch] = ok.x ? s.col.x : 0.f;
    p[(C_COL + 1) * pitch] = ok.y ? s.col.y : 0.f;
    p[(C_COL + 2) * pitch] = ok.z ? s.col.z : 0.f;
}


}  // namespace statmc
