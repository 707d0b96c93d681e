// statmc_pointwise.hip -- the HBM-bound kernels of the statistics path (gfx950).
//
//   prepass        (n, mean, m2, m3) -> Johnson-corrected mean + discriminator
//   mean_vars      film_m2 / ((n-1) n)            (Estimator::CalculateMeanVars)
//   accumulate     sample stream -> running moments (StatTile::Add*Sample* + Merge*Tile)
//   merge_tiles    reference-layout AoS tiles -> planar images (Estimator::Merge*Tile)
//   film_update    Film::Pixel AoS -> interleaved RGB "film" image (Film::UpdateImage)
//   tile_moments   tile-local pooled moments by wavefront-level Welford/Chan merges
//
// Layout choice.  The reference images are interleaved (cv::Mat of Vec3f) and stay that way.
// Every statistic on this path is element-wise per channel, with only the sample count n shared
// by the channels of a pixel, so the kernels walk an image as a flat array of
// width*height*channels scalars and give each lane 4 consecutive PIXELS (channels x float4 per
// plane, one int4 of counts): all traffic is 16 B per lane, whatever the channel count (no
// float3 gathers), and n never crosses lanes.
//
// Arithmetic is written in the exact operation order of oracle/statmc_oracle.c and the file is
// compiled with -ffp-contract=off, so everything except sqrt-vs-pow in the Box-Cox transform
// rounds identically to the CPU restatement.

#include <type_traits>

#include "statmc_device.h"

// sample rows in flight per lane (x2: current + next group) per kind of stat type; the defaults are
// the measured optimum (tools/experiments/build_variant.sh sweeps them)
#ifndef STATMC_ACC_U_RGB_T
#define STATMC_ACC_U_RGB_T 3
#endif
#ifndef STATMC_ACC_U_RGB
#define STATMC_ACC_U_RGB 3
#endif
#ifndef STATMC_ACC_U_F
#define STATMC_ACC_U_F 6
#endif
// RGB sample planes stream through LDS-DMA (global_load_lds_dwordx4, non-temporal): a wave's sample row is 3 KiB of
// contiguous memory, three 1-KiB transfers land it in a wave-private ring of STATMC_ACC_DMA_D rows, every lane reads its own
// 48 B back.  tools/microbench/hbm_read_ldsdma.hip: 7.0 TB/s against 6.26 for the same walk with loads into registers
// (which for a 48-B lane stride coalesce only through the cache and must not be non-temporal); 1-channel planes keep
// their non-temporal register loads (6.8 TB/s against 6.3 - 6.6 through LDS-DMA).  Ring depth (STATMC_ACC_DMA_D, below): 3 rows --
// 240 VGPRs, two waves per SIMD; with 5 the compiler hoists all five row reads (256 VGPRs + AGPR copies), occupancy
// drops to one wave per SIMD and every type loses.
// element pairs folded stage by stage together (RGB types: 6 pairs per lane and sample)
#ifndef STATMC_ACC_PAIR_GROUP
#define STATMC_ACC_PAIR_GROUP 3
#endif
// timing experiment only (variant builds): the walk without its state stores -- what the stores cost
#ifndef STATMC_ACC_SKIP_STORES
#define STATMC_ACC_SKIP_STORES 0
#elif defined(STATMC_PRODUCT_BUILD) && STATMC_ACC_SKIP_STORES
#error "STATMC_ACC_SKIP_STORES is a timing experiment: the results are wrong"
#endif
// the state planes are written once per launch and not read again before the next one: non-temporal stores
// (tools/microbench/acc_model.hip: the stores are 1 % of the bytes and 9 % of the time of a one-type walk, streaming ones cost a
// quarter less there; in the kernel the radiance type alone gains 3 %, the full mix nothing: off)
#ifndef STATMC_ACC_NT_STORES
#define STATMC_ACC_NT_STORES 0
#endif
#ifndef STATMC_ACC_WAVES
#define STATMC_ACC_WAVES 2   // waves per SIMD the film-major kernel is compiled for (3: 168 VGPRs, 346 spills, slower)
#endif
#ifndef STATMC_ACC_DMA_D
#define STATMC_ACC_DMA_D 3
#endif
#ifndef STATMC_ACC_TILES_DMA_D
#define STATMC_ACC_TILES_DMA_D 3     // ring depth of the tile-fed kernel
#endif
#ifndef STATMC_ACC_OCC_AB
#define STATMC_ACC_OCC_AB 0
#endif
#ifndef STATMC_ACC_DMA_DEPTHS
#define STATMC_ACC_DMA_DEPTHS 0      // 1 (experiment builds): the film-major kernel at ring depths 3 .. 6, chosen by statmc_debug_accumulate_dma
#endif
#include "t_quantiles.h"

namespace statmc {

constexpr int kAccDmaD = STATMC_ACC_DMA_D;                 // sample rows in flight per wave (RGB types): the default ring depth
constexpr int kAccTilesDmaD = STATMC_ACC_TILES_DMA_D;
constexpr int kAccDmaMaxD = 6;                              // 2 workgroups x 4 waves x D rows x 3 KiB <= 160 KiB of LDS
constexpr int acc_ring_floats(int d) { return d * 3 * 256; }                                  // per wave: D rows of 64 lanes x 12 floats
constexpr size_t acc_lds_bytes(int d) { return (size_t)4 * acc_ring_floats(d) * sizeof(float); }   // four waves per workgroup
template <int N> __device__ __forceinline__ void acc_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Student-t tables, uploaded once by statmc_setup() (hipMemcpyToSymbol).
__device__ float g_tq[STATMC_TQ_N_TABLES][STATMC_TQ_N_DOF];
// fl(t * t) of every entry: what the Welch pair test of the pair-symmetric kernel multiplies with (the oracle forms
// (t * t) * s; the product t * t rounds the same here as there).  Indexed by the degrees of freedom themselves, 0 .. 4096:
// entry 0 repeats entry 1, which is where nu < 1 and a NaN nu land (the oracle's `nu >= 1 ? ... : 1`) after the
// hardware's unsigned conversion.
__device__ float g_tq2[STATMC_TQ_N_TABLES][STATMC_TQ_N_DOF + 1];

hipError_t upload_t_table(int table, const float *host_4096) {
    if (hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_tq), host_4096, sizeof(float) * STATMC_TQ_N_DOF, sizeof(float) * STATMC_TQ_N_DOF * table); e != hipSuccess)
        return e;
    static thread_local float sq[STATMC_TQ_N_DOF + 1];
    for (int i = 0; i < STATMC_TQ_N_DOF; i++) {
        volatile float t = host_4096[i];     // one rounded product, never contracted or widened
        sq[i + 1] = t * t;
    }
    sq[0] = sq[1];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tq2), sq, sizeof(sq), sizeof(sq) * table);
}

hipError_t upload_t_tables() {
    for (int t = 0; t < STATMC_TQ_N_TABLES; t++)
        if (hipError_t e = upload_t_table(t, statmc_tq_tables[t]); e != hipSuccess) return e;
    return hipSuccess;
}
const float *t_table_sq_device_ptr(int table) {
    float *base = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void **>(&base), HIP_SYMBOL(g_tq2)) != hipSuccess) return nullptr;
    return base + (size_t)table * (STATMC_TQ_N_DOF + 1);
}
const float *t_table_device_ptr(int table) {
    float *base = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void **>(&base), HIP_SYMBOL(g_tq)) != hipSuccess) return nullptr;
    return base + (size_t)table * STATMC_TQ_N_DOF;
}

// table = alpha_index + 3 * sides (t_quantiles.h)
__device__ __forceinline__ float t_quantile(int table, int dof) {
    if (dof < 1) return __builtin_inff();
    if (dof > STATMC_TQ_N_DOF) dof = STATMC_TQ_N_DOF;
    return g_tq[table][dof - 1];
}

constexpr int kBlock = 256;
typedef float vfloat4 __attribute__((ext_vector_type(4)));

static inline int grid_for(long long work_items, int cap = 256 * 16) {
    long long b = (work_items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------ pre-pass
// t = 1 in Welch mode (the pair looks its quantile up itself); exclude: n < 2 takes the pixel out of every window
__device__ __forceinline__ void prepass_elem(int ni, float t, float mu, float s2sum, float s3sum,
                                             float &mc, float &dc, bool exclude_small_n = false) {
    const float nf = (float)ni;
    if (ni >= 2 && s2sum > 0.f) {
        const float var = s2sum / (nf - 1.f);
        const float mu3 = s3sum / nf;
        mc = mu + mu3 / (6.f * var * nf);
        dc = (t * t) * (var / nf);
    } else if (ni < 2 && exclude_small_n) {
        mc = __builtin_nanf("");
        dc = __builtin_nanf("");
    } else {
        mc = mu;
        dc = ni >= 2 ? 0.f : __builtin_inff();
    }
}

template <bool VEC4>
__global__ __launch_bounds__(kBlock) void prepass_kernel(PrepassArgs a) {
    const long long n_groups = (a.n_elems + 3) >> 2;
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < n_groups;
         g += (long long)gridDim.x * kBlock) {
        const long long e0 = g << 2;
        if (VEC4 && e0 + 4 <= a.n_elems) {
            const float4 mu = *reinterpret_cast<const float4 *>(a.mean + e0);
            const float4 s2 = *reinterpret_cast<const float4 *>(a.m2 + e0);
            const float4 s3 = *reinterpret_cast<const float4 *>(a.m3 + e0);
            float4 mc, dc;
            const float *pmu = &mu.x, *ps2 = &s2.x, *ps3 = &s3.x;
            float *pmc = &mc.x, *pdc = &dc.x;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ni = a.n[(e0 + j) / a.channels];
                prepass_elem(ni, a.welch ? 1.f : t_quantile(a.table, ni - 1), pmu[j], ps2[j], ps3[j], pmc[j], pdc[j], a.small_n_exclude);
            }
            *reinterpret_cast<float4 *>(a.mean_corr + e0) = mc;
            *reinterpret_cast<float4 *>(a.disc + e0) = dc;
        } else {
            for (long long e = e0; e < a.n_elems && e < e0 + 4; e++) {
                const int ni = a.n[e / a.channels];
                float mc, dc;
                prepass_elem(ni, a.welch ? 1.f : t_quantile(a.table, ni - 1), a.mean[e], a.m2[e], a.m3[e], mc, dc, a.small_n_exclude);
                a.mean_corr[e] = mc;
                a.disc[e] = dc;
            }
        }
    }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

hipError_t launch_prepass(const PrepassArgs &a, hipStream_t s) {
    const bool vec = aligned16(a.mean) && aligned16(a.m2) && aligned16(a.m3) && aligned16(a.mean_corr) &&
                     aligned16(a.disc);
    const int grid = grid_for((a.n_elems + 3) / 4);
    if (vec)
        hipLaunchKernelGGL(prepass_kernel<true>, dim3(grid), dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(prepass_kernel<false>, dim3(grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ pre-pass + pack
// The multi-GPU block path needs the five filter inputs of the owned block as one 15-channel block +
// halo image (statmc_filter.hip: pack_inputs_kernel).  Doing the pre-pass in the same pass saves the
// round trip of mean_corr / disc through HBM (184 -> 136 B per pixel, or 160 with the two by-products
// still written out) and a launch.  Same prepass_elem, same bits.
struct f3p {
    float x, y, z;
};
__global__ __launch_bounds__(kBlock) void prepass_pack_kernel(PrepassPackArgs a) {
    const long long n_px = (long long)a.src_w * a.src_h;
    for (long long i0 = (long long)blockIdx.x * kBlock + threadIdx.x; i0 < n_px; i0 += (long long)gridDim.x * kBlock) {
        const int y0 = (int)(i0 / a.src_w), x = (int)(i0 - (long long)y0 * a.src_w);
        const int y = y0 + (y0 >= a.split_row ? a.skip_rows : 0);   // the second of two row ranges
        const long long i = (long long)y * a.src_w + x;
        const int ni = a.n[i];
        const float t = a.welch ? 1.f : t_quantile(a.table, ni - 1);
        const f3p mu = reinterpret_cast<const f3p *>(a.mean)[i], s2 = reinterpret_cast<const f3p *>(a.m2)[i],
                  s3 = reinterpret_cast<const f3p *>(a.m3)[i];
        f3p mc, dc;
        prepass_elem(ni, t, mu.x, s2.x, s3.x, mc.x, dc.x, a.small_n_exclude);
        prepass_elem(ni, t, mu.y, s2.y, s3.y, mc.y, dc.y, a.small_n_exclude);
        prepass_elem(ni, t, mu.z, s2.z, s3.z, mc.z, dc.z, a.small_n_exclude);
        if (a.mean_corr) reinterpret_cast<f3p *>(a.mean_corr)[i] = mc;
        if (a.disc) reinterpret_cast<f3p *>(a.disc)[i] = dc;
        float *px = a.packed + ((long long)(y + a.dst_y0) * a.dst_w + (x + a.dst_x0)) * a.ch;
        f3p *dst = reinterpret_cast<f3p *>(px);
        const f3p zero = {0.f, 0.f, 0.f};
        dst[0] = mc;
        dst[1] = dc;
        dst[2] = reinterpret_cast<const f3p *>(a.colour)[i];
        dst[3] = a.g0 ? reinterpret_cast<const f3p *>(a.g0)[i] : zero;
        dst[4] = a.g1 ? reinterpret_cast<const f3p *>(a.g1)[i] : zero;
        if (a.ch >= 17) {
            px[15] = a.s0 ? a.s0[i] : 0.f;
            px[16] = a.s1 ? a.s1[i] : 0.f;
            if (a.ch == 18) px[17] = __int_as_float(ni);
        } else if (a.ch == 16) {
            px[15] = __int_as_float(ni);
        }
    }
}

hipError_t launch_prepass_pack(const PrepassPackArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(prepass_pack_kernel, dim3(grid_for((long long)a.src_w * a.src_h)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ mean vars
__global__ __launch_bounds__(kBlock) void mean_vars_kernel(MeanVarsArgs a) {
    const long long n_elems = (long long)a.width * a.height * a.channels;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < n_elems;
         e += (long long)gridDim.x * kBlock) {
        const long long px = e / a.channels;
        const long long npx = a.row_n_quirk ? (px / a.width) * a.width : px;  // estimator.cpp:540,558
        const float nf = (float)a.n[npx];
        a.film_var[e] = a.film_m2[e] / ((nf - 1.f) * nf);
    }
}

hipError_t launch_mean_vars(const MeanVarsArgs &a, hipStream_t s) {
    const long long n_elems = (long long)a.width * a.height * a.channels;
    hipLaunchKernelGGL(mean_vars_kernel, dim3(grid_for(n_elems)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ accumulate
// One lane owns 4 consecutive PIXELS of one stat type (4*C consecutive scalar elements = C
// float4 per image plane, one int4 of counts) and walks the batch's samples in order (sample s
// of pixel p, channel c is at samples[s*n_elems + p*C + c]), so the update sequence per element
// is the reference's (estimator.h:162-226).  Owning whole pixels keeps the count n private to
// the lane (no cross-lane read/write race on n).  blockIdx.y selects the stat type.
struct ElemState {
    float mean, m2, m3, fmean, fm2;
};

// d / n for an integer-valued divisor n in [1, 2^24): `r` is 1/n refined from v_rcp_f32 by one
// Newton step (shared by every division by the same count: all channels, both Welford chains),
// then one residual correction of the quotient (Markstein).  Bit-identical to the IEEE
// quotient for the operand ranges of this path (tests/test_gpu_parity.py::test_exact_division
// sweeps n = 1..4096 against `/`); 3 + 3 instructions instead of ~10 per division.
__device__ __forceinline__ float refined_rcp(float nf) {
    const float y0 = __builtin_amdgcn_rcpf(nf);
    const float e = __builtin_fmaf(-nf, y0, 1.f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_by_count(float d, float nf, float r) {
    const float q0 = d * r;
    const float rem = __builtin_fmaf(-q0, nf, d);
    return __builtin_fmaf(rem, r, q0);
}

template <int MAXM, bool TRANSFORM>
__device__ __forceinline__ void add_sample(ElemState &st, float nf, float r, float smp) {
    // estimator.h:215 -- boxCox(sample, .5f) = (pow(v, .5) - 1) / .5; v_sqrt_f32 (1 ulp) stands
    // in for pow(v, .5), itself only faithfully rounded in the reference's libm.
    const float v = TRANSFORM ? (__builtin_amdgcn_sqrtf(smp) - 1.f) / .5f : smp;
    const float d = v - st.mean;
    const float dN = div_by_count(d, nf, r);
    if (MAXM >= 3) {
        const float d2 = d * d;
        const float dN2 = dN * dN;
        st.mean += dN;
        st.m2 += d * (d - dN);
        st.m3 += -3.f * dN * st.m2 + d * (d2 - dN2);
    } else if (MAXM == 2) {
        st.mean += dN;
        st.m2 += d * (d - dN);
    } else {
        st.mean += dN;
    }
    if (TRANSFORM) {  // estimator.h:217-225
        const float fd = smp - st.fmean;
        const float fdN = div_by_count(fd, nf, r);
        st.fmean += fdN;
        st.fm2 += fd * (fd - fdN);
    }
}

// The same update on two elements at once (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one issue slot for two lanes'
// worth of IEEE fp32 arithmetic, each component rounded exactly like the scalar instruction, so the bits are those of
// add_sample).  The kernel holds ~250 VGPRs, i.e. two waves per SIMD, and while one of them waits for memory the other
// issues one vector instruction per 4 cycles: the radiance type (Box-Cox + three moments + the raw-sample chain, 27
// instructions per element and sample) was bound by exactly that issue rate (5.1 TB/s on its own); paired, it is
// bound by the memory like the feature types.
typedef float v2f __attribute__((ext_vector_type(2)));
struct PairState {
    v2f mean, m2, m3, fmean, fm2;
};
__device__ __forceinline__ v2f div_by_count2(v2f d, v2f nf, v2f r) {
    const v2f q0 = d * r;
    const v2f rem = __builtin_elementwise_fma(-q0, nf, d);
    return __builtin_elementwise_fma(rem, r, q0);
}
// G pairs at a time, stage by stage: consecutive instructions belong to different pairs, so none waits for the one
// before it (a dependent packed instruction costs a wait state, and the kernel is compiled without the machine scheduler:
// the source order is the issue order).
#define STATMC_PAIRS(i) _Pragma("unroll") for (int i = 0; i < G; i++)
template <int G, int MAXM, bool TRANSFORM>
__device__ __forceinline__ void add_sample2(PairState *st, const v2f *nf, const v2f *r, const v2f *smp) {
    v2f v[G], d[G], q0[G], dN[G];
    if (TRANSFORM) {   // estimator.h:215 -- boxCox(sample, .5f), as in add_sample
        STATMC_PAIRS(i) v[i] = v2f{__builtin_amdgcn_sqrtf(smp[i].x), __builtin_amdgcn_sqrtf(smp[i].y)};
        // (root - 1) / .5 in one instruction: fma(root, 2, -2) rounds 2 root - 2 = 2 (root - 1) once, and doubling commutes with
        // rounding (no overflow: root < 2^64; no underflow: |root - 1| is 0 or at least 2^-24) -- the bits of the two-step form
        STATMC_PAIRS(i) v[i] = __builtin_elementwise_fma(v[i], v2f{2.f, 2.f}, v2f{-2.f, -2.f});
    } else {
        STATMC_PAIRS(i) v[i] = smp[i];
    }
    STATMC_PAIRS(i) d[i] = v[i] - st[i].mean;
    STATMC_PAIRS(i) q0[i] = d[i] * r[i];                                         // div_by_count, three stages
    STATMC_PAIRS(i) dN[i] = __builtin_elementwise_fma(-q0[i], nf[i], d[i]);
    STATMC_PAIRS(i) dN[i] = __builtin_elementwise_fma(dN[i], r[i], q0[i]);
    STATMC_PAIRS(i) st[i].mean += dN[i];
    if (MAXM >= 2) {
        v2f t[G];
        STATMC_PAIRS(i) t[i] = d[i] - dN[i];
        STATMC_PAIRS(i) t[i] = d[i] * t[i];
        STATMC_PAIRS(i) st[i].m2 += t[i];
    }
    if (MAXM >= 3) {   // m3 += -3 dN m2 + d (d^2 - dN^2), with the m2 already updated (estimator.h:178-180)
        v2f a[G], b[G];
        STATMC_PAIRS(i) a[i] = -3.f * dN[i];
        STATMC_PAIRS(i) b[i] = d[i] * d[i];
        STATMC_PAIRS(i) dN[i] = dN[i] * dN[i];
        STATMC_PAIRS(i) a[i] = a[i] * st[i].m2;
        STATMC_PAIRS(i) b[i] = b[i] - dN[i];
        STATMC_PAIRS(i) b[i] = d[i] * b[i];
        STATMC_PAIRS(i) a[i] = a[i] + b[i];
        STATMC_PAIRS(i) st[i].m3 += a[i];
    }
    if (TRANSFORM) {   // the raw-sample chain (estimator.h:217-225)
        STATMC_PAIRS(i) d[i] = smp[i] - st[i].fmean;
        STATMC_PAIRS(i) q0[i] = d[i] * r[i];
        STATMC_PAIRS(i) dN[i] = __builtin_elementwise_fma(-q0[i], nf[i], d[i]);
        STATMC_PAIRS(i) dN[i] = __builtin_elementwise_fma(dN[i], r[i], q0[i]);
        STATMC_PAIRS(i) st[i].fmean += dN[i];
        STATMC_PAIRS(i) dN[i] = d[i] - dN[i];
        STATMC_PAIRS(i) dN[i] = d[i] * dN[i];
        STATMC_PAIRS(i) st[i].fm2 += dN[i];
    }
}
#undef STATMC_PAIRS

// The lane's 4 consecutive pixels starting at film pixel p0 (16-B aligned planes): load the
// state, fold S samples in order, store.  Sample s of the lane's elements is at sp + s * stride
// (4*C consecutive floats): stride = n_elems for sample-major film planes, = tile pixels * C for
// the tile-major arena of accumulate_tiles.
// UMUL: prefetch depth multiplier (tile-fed path, types whose state is small enough to afford the registers)
// DMA (C == 3 only): the sample rows arrive by LDS-DMA in the wave's ring.  That walk is COOPERATIVE -- the 16-byte piece a
// lane fetches belongs to another lane's pixels -- so the whole wave calls in: lane l owns the 4-pixel group at
// sp_wave + 12 l floats, `active` says whether that group exists, `n_active` (wave-uniform) how many lanes' groups do
// (they are the first n_active lanes).  Without DMA an inactive lane returns at once.
template <int C, int MAXM, bool TRANSFORM, int UMUL = 1, int DMA = 0>
__device__ __forceinline__ void accumulate_lane(const AccumulateType &t, long long p0_in, const float *sp,
                                                long long stride, int S, float *ring = nullptr, bool active = true, int n_active = 64,
                                                bool dma_first = false) {
    constexpr bool kDma = DMA > 0 && C == 3;
    constexpr int kD = DMA > 0 ? DMA : 1;               // ring depth: sample rows in flight
    if constexpr (!kDma) {
        if (!active) return;
    }
    const long long p0 = active ? p0_in : 0;   // an inactive lane of the cooperative walk touches no state: its loads read group 0, it stores nothing
    // LDS-DMA walk: the geometry of the wave's sample rows.  (dma_first: the first D rows requested BEFORE the state loads, so
    // that the two latencies a wave pays before its first fold overlap instead of adding up -- measured in round 4 at 4 .. 256
    // samples per pixel, 720p / 1080p / 4K: within +- 0.5 % of requesting them behind the state, profiles/r04_acc_launch.log;
    // kept as a switch of the A/B hook, off.)
    [[maybe_unused]] int dma_piece[3] = {0, 0, 0};
    [[maybe_unused]] const float *dma_row0 = nullptr;
    auto dma_issue = [&](int s, int slot) {
        const float *src = dma_row0 + (long long)s * stride;
        float *dst = ring + slot * 768;
#pragma unroll
        for (int k = 0; k < 3; k++)
            __builtin_amdgcn_global_load_lds(src + dma_piece[k], (__attribute__((address_space(3))) void *)(dst + 256 * k), 16, 0, 2);
    };
    if constexpr (kDma) {
        const int lane = threadIdx.x & 63;
        dma_row0 = sp - 12 * lane;                          // the wave's row of sample 0 (lane l sits 12 l floats in; the same value in every lane)
        const int row_floats = 12 * n_active;               // the part of the wave's row that exists
        // Every lane issues all three transfers of a row, whatever part of the row exists: the waits of the walk COUNT
        // transfers (vmcnt), so their number per row must not depend on n_active.  A piece beyond the row's end re-reads the
        // row's first 16 bytes (memory that exists) into a part of the slot nobody reads.
#pragma unroll
        for (int k = 0; k < 3; k++) dma_piece[k] = 256 * k + 4 * lane < row_floats ? 256 * k + 4 * lane : 0;
        if (dma_first) {
#pragma unroll
            for (int d = 0; d < kD; d++)
                if (d < S) dma_issue(d, d);
        }
    }
    constexpr int NE = 4 * C;  // elements per lane
    const long long e0 = p0 * C;
    // (the epilogue's descriptor fields, read BEFORE the first store: read behind the stores they keep the by-value kernel argument
    // in a private copy -- 2 KB of scratch per lane and 17 more VGPRs, accumulation 3.67 -> 4.60 ms; measured, round 6)
    float *const pre_mc = MAXM >= 3 ? t.mean_corr : nullptr;
    float *const pre_dc = MAXM >= 3 ? t.disc : nullptr;
    const int pre_table = t.pre_table, pre_flags = t.pre_flags;
    PairState st[NE / 2];   // element pairs (2 i, 2 i + 1) of the lane's 4 C consecutive elements
    float tmp[NE];
    const int4 n4 = *reinterpret_cast<const int4 *>(t.n + p0);
    const int n0[4] = {n4.x, n4.y, n4.z, n4.w};
#define STATMC_LOAD_PLANE(ptr, field, enabled)                                   \
    if (enabled) {                                                               \
        _Pragma("unroll") for (int k = 0; k < C; k++) {                          \
            const float4 v = *reinterpret_cast<const float4 *>((ptr) + e0 + 4 * k); \
            tmp[4 * k] = v.x; tmp[4 * k + 1] = v.y; tmp[4 * k + 2] = v.z; tmp[4 * k + 3] = v.w; \
        }                                                                        \
    } else {                                                                     \
        _Pragma("unroll") for (int j = 0; j < NE; j++) tmp[j] = 0.f;             \
    }                                                                            \
    _Pragma("unroll") for (int j = 0; j < NE / 2; j++) st[j].field = v2f{tmp[2 * j], tmp[2 * j + 1]};
    STATMC_LOAD_PLANE(t.mean, mean, true)
    STATMC_LOAD_PLANE(t.m2, m2, MAXM >= 2)
    STATMC_LOAD_PLANE(t.m3, m3, MAXM >= 3)
    STATMC_LOAD_PLANE(t.film_mean, fmean, TRANSFORM)
    STATMC_LOAD_PLANE(t.film_m2, fm2, TRANSFORM)
#undef STATMC_LOAD_PLANE
    // Software-pipelined sample walk: the loads of the next U samples are issued before
    // the current U are folded into the moments, so 2U sample rows per lane are in flight
    // (the compiler would otherwise drain each unrolled body before loading again).
    // C = 3: the three 16-B loads of a lane interleave across the wave (48-B lane stride);
    // they only coalesce through the cache, so they must be plain loads (non-temporal ones
    // re-fetch the shared lines: 4.8 vs 6.4 TB/s, tools/microbench/hbm_read.hip).  C = 1
    // streams with non-temporal loads.
    constexpr int U = UMUL * (C == 3 ? (TRANSFORM ? STATMC_ACC_U_RGB_T : STATMC_ACC_U_RGB) : STATMC_ACC_U_F);
    auto load_sample = [&](vfloat4 (&dst)[C], const float *src) {
#pragma unroll
        for (int k = 0; k < C; k++)
            dst[k] = C == 1 ? __builtin_nontemporal_load(reinterpret_cast<const vfloat4 *>(src + 4 * k))
                            : *reinterpret_cast<const vfloat4 *>(src + 4 * k);
    };
    // SAME: the lane's 4 pixels hold the same count (every film whose pixels have seen the
    // same number of samples, i.e. all but adaptively sampled ones) -> one count conversion and
    // one refined reciprocal per sample instead of four (a fifth of the kernel's VALU work).
    auto fold_sample = [&](const vfloat4 (&q)[C], int s, auto same) {
        if constexpr (decltype(same)::value) {
            const float nf0 = (float)(n0[0] + s + 1);
            const float rc0 = refined_rcp(nf0);
            v2f nf2[NE / 2], rc2[NE / 2], smp2[NE / 2];
#pragma unroll
            for (int i = 0; i < NE / 2; i++) {   // elements 2 i and 2 i + 1
                nf2[i] = v2f{nf0, nf0};
                rc2[i] = v2f{rc0, rc0};
                smp2[i] = v2f{q[i >> 1][2 * (i & 1)], q[i >> 1][2 * (i & 1) + 1]};
            }
            constexpr int G = C == 3 ? STATMC_ACC_PAIR_GROUP : 2;
#pragma unroll
            for (int g = 0; g < NE / 2; g += G) add_sample2<G, MAXM, TRANSFORM>(st + g, nf2 + g, rc2 + g, smp2 + g);
        } else {
            // ragged counts (adaptively sampled films): element by element, one reciprocal per pixel
            float nf[4], rc[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                nf[p] = (float)(n0[p] + s + 1);
                rc[p] = refined_rcp(nf[p]);
            }
#pragma unroll
            for (int j = 0; j < NE; j++) {
                ElemState e = {st[j >> 1].mean[j & 1], st[j >> 1].m2[j & 1], st[j >> 1].m3[j & 1], st[j >> 1].fmean[j & 1], st[j >> 1].fm2[j & 1]};
                add_sample<MAXM, TRANSFORM>(e, nf[j / C], rc[j / C], q[j >> 2][j & 3]);
                st[j >> 1].mean[j & 1] = e.mean;
                if (MAXM >= 2) st[j >> 1].m2[j & 1] = e.m2;
                if (MAXM >= 3) st[j >> 1].m3[j & 1] = e.m3;
                if (TRANSFORM) { st[j >> 1].fmean[j & 1] = e.fmean; st[j >> 1].fm2[j & 1] = e.fm2; }
            }
        }
    };
    auto walk_samples = [&](auto same) {
        vfloat4 cur[U][C], nxt[U][C];
        const int S_main = S - S % U;
        if (S_main > 0) {
#pragma unroll
            for (int u = 0; u < U; u++) load_sample(cur[u], sp + (long long)u * stride);
        }
        for (int s = 0; s < S_main; s += U) {
            const float *np = sp + (long long)(s + U) * stride;
            if (s + U < S_main) {
#pragma unroll
                for (int u = 0; u < U; u++) load_sample(nxt[u], np + (long long)u * stride);
            }
#pragma unroll
            for (int u = 0; u < U; u++) fold_sample(cur[u], s + u, same);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < C; k++) cur[u][k] = nxt[u][k];
        }
        for (int s = S_main; s < S; s++) {  // remainder (S not a multiple of U)
            vfloat4 q[C];
            load_sample(q, sp + (long long)s * stride);
            fold_sample(q, s, same);
        }
    };
    // The same walk with the sample rows arriving by LDS-DMA (C == 3): D rows in flight in the wave's ring; row s has
    // landed when at most 3 (D - 1) transfers issued after it are outstanding (VMEM operations of a wave complete in
    // order; nothing else touches memory inside the walk).  A slot is refilled once every lane has read its 48 B.
    auto walk_samples_dma = [&](auto same) {
        constexpr int D = kD;
        const int lane = threadIdx.x & 63;
        auto issue = dma_issue;
        auto take = [&](vfloat4 (&q)[C], int slot) {
            const float *mine = ring + slot * 768 + 12 * lane;
#pragma unroll
            for (int k = 0; k < C; k++) q[k] = *reinterpret_cast<const vfloat4 *>(mine + 4 * k);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the slot is refilled
        };
        // every earlier access of the wave to memory (the state loads above; with dma_first also the first D rows, requested
        // ahead of them) has completed before the counted waits start
        acc_wait_vmcnt<0>();
        if (!dma_first) {
#pragma unroll
            for (int d = 0; d < D; d++)
                if (d < S) issue(d, d);
        }
        const int S_full = S >= D ? S - D + 1 : 0;          // samples s < S_full have D - 1 later rows in flight behind them
        // One row per trip, the slot a run-time index (round 5): unrolled by D with compile-time slots the compiler hoisted the
        // LDS reads of all D rows to the top of the body -- D x 12 registers, which is what capped the ring at three rows
        // (240 VGPRs; 5 rows: 256 + AGPR copies, one wave per SIMD).  The body is ~200 instructions; the loop costs nothing.
        int slot = 0;
#pragma unroll 1
        for (int s = 0; s < S; s++) {
            vfloat4 q[C];
            if (s < S_full) acc_wait_vmcnt<3 * (D - 1)>(); else acc_wait_vmcnt<0>();   // the last D - 1 rows: nothing is issued behind them
            take(q, slot);
            if (s + D < S) issue(s + D, slot);
            fold_sample(q, s, same);
            slot = slot + 1 == D ? 0 : slot + 1;
        }
    };
    // wave-uniform choice: the fast walk only when every active lane qualifies
    const bool lane_same = !active || (n0[0] == n0[1] && n0[1] == n0[2] && n0[2] == n0[3]);
    if constexpr (kDma) {
        if (__builtin_amdgcn_ballot_w64(!lane_same) == 0) walk_samples_dma(std::true_type{});
        else walk_samples_dma(std::false_type{});
        if (!active) return;                                // nothing of an inactive lane is stored
    } else {
        if (__builtin_amdgcn_ballot_w64(!lane_same) == 0) walk_samples(std::true_type{});
        else walk_samples(std::false_type{});
    }
#define STATMC_STORE_PLANE(ptr, field, enabled)                                  \
    if (enabled) {                                                               \
        _Pragma("unroll") for (int k = 0; k < C; k++) {                          \
            const vfloat4 v = {st[2 * k].field.x, st[2 * k].field.y, st[2 * k + 1].field.x, st[2 * k + 1].field.y}; \
            if (STATMC_ACC_SKIP_STORES && v.x != 12345.678f) continue;  /* the arithmetic stays alive, the store never happens */ \
            if (STATMC_ACC_NT_STORES) __builtin_nontemporal_store(v, reinterpret_cast<vfloat4 *>((ptr) + e0 + 4 * k)); \
            else *reinterpret_cast<vfloat4 *>((ptr) + e0 + 4 * k) = v;            \
        }                                                                        \
    }
    STATMC_STORE_PLANE(t.mean, mean, true)
    STATMC_STORE_PLANE(t.m2, m2, MAXM >= 2)
    STATMC_STORE_PLANE(t.m3, m3, MAXM >= 3)
    STATMC_STORE_PLANE(t.film_mean, fmean, TRANSFORM)
    STATMC_STORE_PLANE(t.film_m2, fm2, TRANSFORM)
#undef STATMC_STORE_PLANE
    // Merge*Tile casts the tile's uint64 count to int32 (estimator.cpp:347,380)
    typedef int vint4 __attribute__((ext_vector_type(4)));
    const vint4 n_out = {n0[0] + S, n0[1] + S, n0[2] + S, n0[3] + S};
    if (STATMC_ACC_NT_STORES) __builtin_nontemporal_store(n_out, reinterpret_cast<vint4 *>(t.n + p0));
    else *reinterpret_cast<vint4 *>(t.n + p0) = n_out;
    // Optional epilogue (round 6): the pre-pass of the moments just written, from the registers that hold them -- the same
    // prepass_elem as prepass_kernel, hence the same bits -- instead of a launch that reads 40 B per pixel back.
    if constexpr (MAXM >= 3) {
        if (pre_mc != nullptr) {
            float tq[4];
#pragma unroll
            for (int p = 0; p < 4; p++) tq[p] = (pre_flags & 1) ? 1.f : t_quantile(pre_table, n_out[p] - 1);
#pragma unroll
            for (int k = 0; k < C; k++) {
                vfloat4 mc, dc;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = 4 * k + j, px = e / C;
                    float m, d;
                    prepass_elem(n_out[px], tq[px], st[e >> 1].mean[e & 1], st[e >> 1].m2[e & 1], st[e >> 1].m3[e & 1], m, d, (pre_flags & 2) != 0);
                    mc[j] = m;
                    dc[j] = d;
                }
                *reinterpret_cast<vfloat4 *>(pre_mc + e0 + 4 * k) = mc;
                *reinterpret_cast<vfloat4 *>(pre_dc + e0 + 4 * k) = dc;
            }
        }
    }
}

// One pixel, scalar accesses: unaligned images, ragged ends, tiles whose rows do not split into
// 4-pixel groups.  Sample s of channel c is at sp[s * stride + c].
template <int C, int MAXM, bool TRANSFORM>
__device__ __forceinline__ void accumulate_pixel(const AccumulateType &t, long long p, const float *sp,
                                                 long long stride, int S) {
    const int n0 = t.n[p];
    float *const pre_mc = MAXM >= 3 ? t.mean_corr : nullptr;      // (read before the first store: see accumulate_lane)
    float *const pre_dc = MAXM >= 3 ? t.disc : nullptr;
    const int pre_table = t.pre_table, pre_flags = t.pre_flags;
    for (int c = 0; c < C; c++) {
        const long long e = p * C + c;
        ElemState st = {t.mean[e], MAXM >= 2 ? t.m2[e] : 0.f, MAXM >= 3 ? t.m3[e] : 0.f,
                        TRANSFORM ? t.film_mean[e] : 0.f, TRANSFORM ? t.film_m2[e] : 0.f};
        for (int s = 0; s < S; s++) {
            const float nf = (float)(n0 + s + 1);
            add_sample<MAXM, TRANSFORM>(st, nf, refined_rcp(nf), sp[(long long)s * stride + c]);
        }
        t.mean[e] = st.mean;
        if (MAXM >= 2) t.m2[e] = st.m2;
        if (MAXM >= 3) t.m3[e] = st.m3;
        if (TRANSFORM) {
            t.film_mean[e] = st.fmean;
            t.film_m2[e] = st.fm2;
        }
        if (MAXM >= 3 && pre_mc != nullptr) {   // (the optional pre-pass epilogue, as in accumulate_lane)
            const int ni = n0 + S;
            float m, d;
            prepass_elem(ni, (pre_flags & 1) ? 1.f : t_quantile(pre_table, ni - 1), st.mean, st.m2, st.m3, m, d, (pre_flags & 2) != 0);
            pre_mc[e] = m;
            pre_dc[e] = d;
        }
    }
    t.n[p] = n0 + S;
}

// Film-major batch: one lane owns 4 consecutive PIXELS of one stat type and walks the batch's
// samples in order (sample s of pixel p, channel c is at samples[s*n_elems + p*C + c]).
template <int C, int MAXM, bool TRANSFORM, bool VEC, int UMUL, int DMA>
__device__ __forceinline__ void accumulate_type(const AccumulateType &t, long long blk, long long nblk, float *ring, bool dma_first) {
    const long long n_px = t.n_elems / C;
    const long long n_groups = (n_px + 3) >> 2;
    const long long n_full = VEC ? (n_px >> 2) : 0;          // complete 4-pixel groups: the vector path's share
    const int lane = threadIdx.x & 63;
    // wave-uniform walk over the groups: the waves of a workgroup take 64 consecutive groups each (the LDS-DMA walk of the
    // RGB types is cooperative: every lane of a wave with at least one complete group calls in)
    for (long long gw = blk * kBlock + (threadIdx.x & ~63); gw < n_groups; gw += nblk * kBlock) {
        const long long g = gw + lane, p0 = g << 2;
        const bool active = g < n_full;
        const long long left = n_full - gw;
        const int n_active = left >= 64 ? 64 : left > 0 ? (int)left : 0;
        if (n_active > 0)
            accumulate_lane<C, MAXM, TRANSFORM, (!TRANSFORM && MAXM == 1) ? UMUL : 1, DMA>(t, p0, t.samples + p0 * C, t.stride, t.n_samples,
                                                                                            ring, active, n_active, dma_first);
        if (!active && g < n_groups) {   // unaligned images, the ragged last group
            for (long long p = p0; p < n_px && p < p0 + 4; p++)
                accumulate_pixel<C, MAXM, TRANSFORM>(t, p, t.samples + p * C, t.stride, t.n_samples);
        }
    }
}

template <int C, bool VEC, int UMUL, int DMA>
__device__ __forceinline__ void accumulate_dispatch(const AccumulateType &t, long long blk, long long nblk, float *ring, bool dma_first) {
    if (t.transform) {
        if (t.max_moment >= 3) accumulate_type<C, 3, true, VEC, UMUL, DMA>(t, blk, nblk, ring, dma_first);
        else if (t.max_moment == 2) accumulate_type<C, 2, true, VEC, UMUL, DMA>(t, blk, nblk, ring, dma_first);
        else accumulate_type<C, 1, true, VEC, UMUL, DMA>(t, blk, nblk, ring, dma_first);
    } else {
        if (t.max_moment >= 3) accumulate_type<C, 3, false, VEC, UMUL, DMA>(t, blk, nblk, ring, dma_first);
        else if (t.max_moment == 2) accumulate_type<C, 2, false, VEC, UMUL, DMA>(t, blk, nblk, ring, dma_first);
        else accumulate_type<C, 1, false, VEC, UMUL, DMA>(t, blk, nblk, ring, dma_first);
    }
}

// Default grid: stat types are interleaved over a large 1-D grid (block b works on type
// b % n_types) so that the ALU-heavy radiance blocks (Box-Cox + third moment + raw-sample
// Welford) and the purely bandwidth-bound feature blocks are resident together.
// Resident grid (a.resident_blocks > 0, experiment hook): that many workgroups in total; each
// walks every stat type, starting at a different one, with a grid-stride loop.  Measured use:
// running this bandwidth-bound kernel beside the VALU-bound window filter of the previous
// iteration on a second stream gains <= 15 % (the two contend for VALU issue), so bench.py
// keeps the kernels back to back.
template <bool VEC, int UMUL, int DMA, int OCC = STATMC_ACC_WAVES>
__global__ __launch_bounds__(kBlock, OCC) void accumulate_kernel(AccumulateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float acc_lds[];
    // the wave's LDS-DMA ring (RGB types, vector path); DMA = false (debug hook, A/B) keeps every type on register loads
    float *ring = DMA ? acc_lds + (threadIdx.x >> 6) * acc_ring_floats(DMA) : nullptr;
    if (a.resident_blocks > 0) {
        for (int i = 0; i < a.n_types; i++) {
#ifndef STATMC_ACC_RESIDENT_START
#define STATMC_ACC_RESIDENT_START 0
#endif
            // (which type a workgroup starts with: 0 = its index, so that every type is walked by a fifth of the grid at any time; experiment
            // builds: 1 = every workgroup the same type -- 3.92 against 3.75 ms --, 2 = the eight workgroups of a dispatch round -- one per
            // XCD -- the same type: no difference; profiles/r06_ab_resident_start.log)
            const int first = STATMC_ACC_RESIDENT_START == 1 ? 0 : STATMC_ACC_RESIDENT_START == 2 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
            const AccumulateType t = a.t[(first + i) % a.n_types];   // (by value: see below)
            if (t.channels == 3) accumulate_dispatch<3, VEC, UMUL, DMA>(t, blockIdx.x, gridDim.x, ring, a.dma_first != 0);
            else accumulate_dispatch<1, VEC, UMUL, DMA>(t, blockIdx.x, gridDim.x, ring, a.dma_first != 0);
        }
        return;
    }
    if (a.grid_mode == 1) {
        // One pass per workgroup: block b serves stat type b % n_types, groups [256 (b / n_types), + 256) -- every
        // workgroup of a type does the same amount of work, none walks a second, ragged stride.  The dispatcher hands the
        // blocks out in order, so the resident mix always holds every type and the launch ends within one workgroup's time.
        const int ti = blockIdx.x % a.n_types;
        // (a COPY of the descriptor, not a reference into the by-value kernel argument: with the pre-pass epilogue's fields in use a
        // reference made the compiler keep the whole 2-KB argument in a private copy -- scratch 2032 B per lane, + 17 VGPRs, the
        // accumulation 3.67 -> 4.60 ms; round 6)
        const AccumulateType t = a.t[ti];
        const long long blk = blockIdx.x / a.n_types, nblk = gridDim.x / a.n_types;
        if (t.channels == 3) accumulate_dispatch<3, VEC, UMUL, DMA>(t, blk, nblk, ring, a.dma_first != 0);
        else accumulate_dispatch<1, VEC, UMUL, DMA>(t, blk, nblk, ring, a.dma_first != 0);
        return;
    }
    // Workgroups are dealt to the stat types in rounds of n_slots, each type holding a number of
    // slots proportional to its cost (bytes per pixel, radiance weighted up for its ALU work), so
    // that all types finish together instead of leaving the expensive one to run on alone.
    const int slot = blockIdx.x % a.n_slots, round = blockIdx.x / a.n_slots, n_rounds = gridDim.x / a.n_slots;
    const int ti = a.slot_type[slot];
    const AccumulateType t = a.t[ti];
    const long long blk = (long long)round * a.type_slots[ti] + a.slot_rank[slot];
    const long long nblk = (long long)n_rounds * a.type_slots[ti];
    if (t.channels == 3) accumulate_dispatch<3, VEC, UMUL, DMA>(t, blk, nblk, ring, a.dma_first != 0);
    else accumulate_dispatch<1, VEC, UMUL, DMA>(t, blk, nblk, ring, a.dma_first != 0);
}

int acc_diagnostic_bits() { return STATMC_ACC_SKIP_STORES ? 128 : 0; }

static thread_local unsigned g_last_acc_grid = 0;
unsigned last_accumulate_grid() { return g_last_acc_grid; }   // workgroups of the calling thread's last film-major launch (tests)

hipError_t launch_accumulate(const AccumulateArgs &a_in, hipStream_t s) {
    AccumulateArgs a = a_in;
    // slots per type ~ relative cost: 4 B x channels per sample, x1.3 for transform types
    a.n_slots = 0;
    for (int i = 0; i < a.n_types; i++) {
        int w = a.t[i].channels * (a.t[i].transform ? 4 : 3);  // 12 / 9 / 3 (rgb transform / rgb / float)
        w = (w + 2) / 3;                                       // 4 / 3 / 1 slots
        if (w < 1) w = 1;
        a.type_slots[i] = w;
        for (int r = 0; r < w && a.n_slots < kMaxSlots; r++) {
            a.slot_type[a.n_slots] = (unsigned char)i;
            a.slot_rank[a.n_slots] = (unsigned char)r;
            a.n_slots++;
        }
    }
    // interleave the slots of different types (round-robin over types) so neighbours differ
    {
        unsigned char st[kMaxSlots], sr[kMaxSlots];
        int n = 0;
        for (int r = 0; n < a.n_slots; r++)
            for (int i = 0; i < a.n_types; i++)
                if (r < a.type_slots[i]) { st[n] = (unsigned char)i; sr[n] = (unsigned char)r; n++; }
        for (int k = 0; k < a.n_slots; k++) { a.slot_type[k] = st[k]; a.slot_rank[k] = sr[k]; }
    }
    bool vec = true;
    long long max_groups = 1;
    for (int i = 0; i < a.n_types; i++) {
        const AccumulateType &t = a.t[i];
        vec = vec && aligned16(t.samples) && aligned16(t.n) && aligned16(t.mean) &&
              (t.max_moment < 2 || aligned16(t.m2)) && (t.max_moment < 3 || aligned16(t.m3)) &&
              (!t.transform || (aligned16(t.film_mean) && aligned16(t.film_m2))) &&
              (!t.mean_corr || (aligned16(t.mean_corr) && aligned16(t.disc))) &&      // the pre-pass epilogue's images
              (t.n_elems % 4 == 0) && (t.stride % 4 == 0);  // sample planes stay 16-B aligned
        const long long groups = (t.n_elems / t.channels + 3) / 4;
        if (groups > max_groups) max_groups = groups;
    }
    const int rounds = (grid_for(max_groups, 256 * 8) * a.n_types + a.n_slots - 1) / a.n_slots;
    const long long units = (max_groups + kBlock - 1) / kBlock;     // workgroups that cover the largest type once
    // Launch shape.  Long batches: a capped grid, every type holding slots in proportion to its cost, grid-stride walk
    // (the resident mix favours the ALU-heavy radiance type: + 1 - 2 % at 64 and 256 samples per pixel).  Short batches --
    // the 4-, 4-, 8-, 16-sample iterations the reference's progressive schedule starts with (statpath.cpp:272-279) -- are
    // bound by the chain of latencies a workgroup pays per pass, and the capped grid gives the cheap 1-channel types up to
    // three passes per workgroup: there every workgroup makes ONE pass, types round-robin (1080p: 4 spp 5.1 -> 6.6 TB/s,
    // 8 spp 6.1 -> 7.0, 16 spp 6.2 -> 6.6; 4K: 4 spp 4.9 -> 5.7, 16 spp 5.74 -> 5.63; profiles/r04_acc_launch.log).
    int max_s = 0;
    for (int i = 0; i < a.n_types; i++) max_s = a.t[i].n_samples > max_s ? a.t[i].n_samples : max_s;
    // Round 5: with the samples and the moments in different interference classes (a.apart: statmc_malloc_placed blocks) the one-pass
    // shape also wins on 4K films at every batch length (+ 1 - 5 %) and on long batches at 1080p (256 spp: 3.61 against 3.67 - 3.70
    // ms); 32 - 64 samples on films up to 1080p keep the capped grid (+ 1.5 - 2.5 %).  Three processes, profiles/r05_acc_launch_placed.log.
    // Round 6, last session: ONE workgroup per CU walking every type (four waves per CU, half the occupancy) on 1080p-sized films from
    // 256 samples per launch up, placed buffers: the accumulation 3.72 - 3.73 against 3.77 - 3.78 ms in the step, three rounds on each of
    // two boxes with the same placement (profiles/r06_ab_resident.log, r06_ab_resident2.log; round 5 had seen 3.57 / 3.62 and 3.74 / 3.76
    // back to back and left it an experiment hook: it loses at 720p -- 900 units on 256 workgroups -- and at 4K).  Fewer CUs than all
    // starve the stream (192 workgroups: 4.17 ms) though the filter behind it then holds 2.15 instead of 1.98 GHz (1.49 against 1.61 ms).
    // Only there: at 128 samples a tie, at 64 slower (1.044 against 1.016 ms); 1280 x 720, 2560 x 1440 (14.06 walks per workgroup: the
    // fifteenth is nearly empty) and 3840 x 2160 slower by 2 - 6 % (profiles/r06_ab_resident_spp.log).  So: the film's groups must fill the
    // workgroups' last walk (>= 97 %), and the film must be of about that size.
    if (a.resident_blocks == 0 && a.grid_mode < 0 && a.apart && a.cus > 0 && max_s >= 256 && max_groups >= (1 << 18) && max_groups < 3 * (1 << 18)) {
        const double walks = (double)max_groups / ((double)a.cus * kBlock);
        if (walks / ceil(walks) >= 0.97) a.resident_blocks = a.cus;
    }
    if (a.resident_blocks < 0) a.resident_blocks = 0;
    if (a.grid_mode < 0) {
        const bool big = max_groups >= (1 << 20);
        if (a.apart) a.grid_mode = (max_s <= 16 || big || max_s >= 128) ? 1 : 0;
        else a.grid_mode = (max_s <= 8 || (max_s <= 16 && !big)) ? 1 : 0;
    }
    const dim3 grid(a.resident_blocks > 0 ? a.resident_blocks : a.grid_mode == 1 ? (unsigned)(units * a.n_types) : rounds * a.n_slots);
    g_last_acc_grid = grid.x;
    // a.dma: 0 = loads into registers (A/B), 1 = the default ring depth, 3 .. 6 = that depth where the build holds it
    // (STATMC_ACC_DMA_DEPTHS: experiment builds instantiate every depth, the product build the default one)
    const int depth = a.dma == 1 ? kAccDmaD : a.dma;
#define STATMC_LAUNCH_DEPTH(D) \
    if (vec && depth == D) { hipLaunchKernelGGL((accumulate_kernel<true, 1, D>), grid, dim3(kBlock), acc_lds_bytes(D), s, a); return hipGetLastError(); }
#if STATMC_ACC_OCC_AB       // experiment builds: the default depth compiled for three waves per SIMD as well (statmc_debug_accumulate_occupancy)
    if (vec && depth == kAccDmaD && a.occ == 3) {
        hipLaunchKernelGGL((accumulate_kernel<true, 1, kAccDmaD, 3>), grid, dim3(kBlock), acc_lds_bytes(kAccDmaD), s, a);
        return hipGetLastError();
    }
#endif
    STATMC_LAUNCH_DEPTH(kAccDmaD)
#if STATMC_ACC_DMA_DEPTHS
    STATMC_LAUNCH_DEPTH(3) STATMC_LAUNCH_DEPTH(4) STATMC_LAUNCH_DEPTH(5) STATMC_LAUNCH_DEPTH(6)
#endif
#undef STATMC_LAUNCH_DEPTH
    if (vec && depth != 0) return hipErrorInvalidValue;      // a ring depth this build does not hold
    if (vec && a.umul == 2)
        hipLaunchKernelGGL((accumulate_kernel<true, 2, 0>), grid, dim3(kBlock), 0, s, a);
    else if (vec)
        hipLaunchKernelGGL((accumulate_kernel<true, 1, 0>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((accumulate_kernel<false, 1, 0>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ accumulate, tile by tile
// The same update fed the way StatPathIntegrator::Render produces samples (statpath.cpp:132-190,
// 355-388): every tile of the film (16 x 16 there) hands over the samples of one iteration as one
// block [S_tile][tile_h][tile_w][C] of a per-type arena, S_tile the same for every pixel of the
// tile but free to differ between tiles.  One wave per (tile, type): a 16 x 16 tile is exactly
// 64 lanes x 4 pixels, a sample plane of the block is one contiguous 1 / 3 KiB read of the wave.
template <int C, int MAXM, bool TRANSFORM, int UMUL, int DMA>
__device__ __forceinline__ void accumulate_tile(const AccumulateType &t, const AccumulateTilesArgs &a, int x0, int y0,
                                                int tw, int th, long long off, int S, float *ring) {
    const int lane = threadIdx.x & 63;
    const int npx = tw * th;
    const float *base = t.samples + off * C;
    const long long stride = (long long)npx * C;
    // rows split into 16-B aligned 4-pixel groups: tile width, tile origin, image width and the
    // block's offset are all multiples of 4 (and the images themselves 16-B aligned: a.vec)
    const bool fast = a.vec && ((tw | x0 | a.width) & 3) == 0 && (off & 3) == 0;
    if (fast) {
        const int n_g = npx >> 2;
        for (int gw = 0; gw < n_g; gw += 64) {   // wave-uniform: the LDS-DMA walk of the RGB types is cooperative
            const int g = gw + lane;
            const bool active = g < n_g;
            const int i = g << 2, row = i / tw, col = i - row * tw;
            const long long p0 = (long long)(y0 + row) * a.width + x0 + col;
            // deeper prefetch only where the state is one plane (mean-only feature types): the registers are there
            accumulate_lane<C, MAXM, TRANSFORM, (!TRANSFORM && MAXM == 1) ? UMUL : 1, DMA>(t, p0, base + (long long)i * C, stride, S, ring, active,
                                                                                            n_g - gw >= 64 ? 64 : n_g - gw, a.dma_first != 0);
        }
    } else {
        for (int i = lane; i < npx; i += 64) {
            const int row = i / tw, col = i - row * tw;
            const long long p = (long long)(y0 + row) * a.width + x0 + col;
            accumulate_pixel<C, MAXM, TRANSFORM>(t, p, base + (long long)i * C, stride, S);
        }
    }
}

template <int C, int UMUL, int DMA>
__device__ __forceinline__ void accumulate_tile_dispatch(const AccumulateType &t, const AccumulateTilesArgs &a, int x0,
                                                         int y0, int tw, int th, long long off, int S, float *ring) {
    if (t.transform) {
        if (t.max_moment >= 3) accumulate_tile<C, 3, true, UMUL, DMA>(t, a, x0, y0, tw, th, off, S, ring);
        else if (t.max_moment == 2) accumulate_tile<C, 2, true, UMUL, DMA>(t, a, x0, y0, tw, th, off, S, ring);
        else accumulate_tile<C, 1, true, UMUL, DMA>(t, a, x0, y0, tw, th, off, S, ring);
    } else {
        if (t.max_moment >= 3) accumulate_tile<C, 3, false, UMUL, DMA>(t, a, x0, y0, tw, th, off, S, ring);
        else if (t.max_moment == 2) accumulate_tile<C, 2, false, UMUL, DMA>(t, a, x0, y0, tw, th, off, S, ring);
        else accumulate_tile<C, 1, false, UMUL, DMA>(t, a, x0, y0, tw, th, off, S, ring);
    }
}

template <int UMUL, int DMA>
__global__ __launch_bounds__(kBlock) void accumulate_tiles_kernel(AccumulateTilesArgs a) {
    extern __shared__ __attribute__((aligned(16))) float acc_lds[];
    float *ring = DMA ? acc_lds + (threadIdx.x >> 6) * acc_ring_floats(DMA) : nullptr;
    // a.order = 2 (the default since round 5): a workgroup = four consecutive tiles of ONE type, workgroups round-robin over the
    // types -- the four waves' state rows are 768 contiguous bytes (192 with the types of ONE tile per workgroup, order 0, where
    // the neighbouring tile's rows are another workgroup's, usually on another XCD: every line of the moments fetched into two L2s):
    // + 2 - 6 % at 4 .. 64 samples per tile, 1080p and 4K (profiles/r05_tiles_order.log); CUs still hold every type at any time
    const long long n_items = a.order == 2 ? (((long long)a.n_tiles + 3) >> 2) * 4 * a.n_types : (long long)a.n_tiles * a.n_types;
    const long long n_waves = (long long)gridDim.x * (kBlock / 64);
    // a.order = 0 (rounds 2 - 4; A/B): item = (tile, type), types innermost: the waves of a workgroup work on the types of one
    // tile, so ALU-heavy radiance items and bandwidth-only feature items share every CU
    // (a.order = 1, experiment: tiles innermost -- neighbouring waves read neighbouring blocks of one type's arena)
    // Round 6, two experiments on why this walk trails the film-major kernel at 4 - 16 samples per tile (same HBM
    // bytes, same L2 hits and misses, 1.28 x the wave cycles: profiles/r06_tiles_pmc_S4.log), both measured and removed:
    // the NEXT item's tile record (bounds, sample count, arena offset) fetched before the current item starts -- 0.627 -> 0.632 of the
    // HBM peak at 4 samples, nothing above (profiles/r06_tiles_prefetch.log); the four waves of a workgroup TOGETHER on four adjacent
    // tiles, a band of four rows each (moments in 768-byte runs instead of 192-byte ones) -- 0.675 -> 0.648 at 4 samples, 0.720 -> 0.685
    // at 8, slower at every length (profiles/r06_tiles_quad.log).  Neither the latency of the record nor the shape of the accesses, then.
    for (long long item = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); item < n_items; item += n_waves) {
        int tile = __builtin_amdgcn_readfirstlane(a.order ? (int)(item % a.n_tiles) : (int)(item / a.n_types));
        int ti = __builtin_amdgcn_readfirstlane(a.order ? (int)(item / a.n_tiles) : (int)(item % a.n_types));
        if (a.order == 2) {
            const long long q = item >> 2;
            ti = __builtin_amdgcn_readfirstlane((int)(q % a.n_types));
            tile = __builtin_amdgcn_readfirstlane((int)(q / a.n_types) * 4 + (int)(item & 3));
            if (tile >= a.n_tiles) continue;
        }
        const int x0 = a.tile_bounds[4 * tile], y0 = a.tile_bounds[4 * tile + 1];
        const int x1 = a.tile_bounds[4 * tile + 2], y1 = a.tile_bounds[4 * tile + 3];
        const int S = a.tile_samples[tile];
        if (S <= 0 || x1 <= x0 || y1 <= y0) continue;
        const AccumulateType &t = a.t[ti];
        if (t.channels == 3) accumulate_tile_dispatch<3, UMUL, DMA>(t, a, x0, y0, x1 - x0, y1 - y0, a.tile_offsets[tile], S, ring);
        else accumulate_tile_dispatch<1, UMUL, DMA>(t, a, x0, y0, x1 - x0, y1 - y0, a.tile_offsets[tile], S, ring);
    }
}

hipError_t launch_accumulate_tiles(const AccumulateTilesArgs &a_in, hipStream_t s) {
    AccumulateTilesArgs a = a_in;
    bool vec = true;
    for (int i = 0; i < a.n_types; i++) {
        const AccumulateType &t = a.t[i];
        vec = vec && aligned16(t.samples) && aligned16(t.n) && aligned16(t.mean) &&
              (t.max_moment < 2 || aligned16(t.m2)) && (t.max_moment < 3 || aligned16(t.m3)) &&
              (!t.transform || (aligned16(t.film_mean) && aligned16(t.film_m2))) &&
              (!t.mean_corr || (aligned16(t.mean_corr) && aligned16(t.disc)));
    }
    a.vec = vec ? 1 : 0;
    const long long items = a.order == 2 ? (((long long)a.n_tiles + 3) >> 2) * 4 * a.n_types : (long long)a.n_tiles * a.n_types;
    // one wave per item; films up to ~ 4 Mpixels: at most 8 workgroups per CU, walking the items with a grid stride; larger ones
    // (4K: 32 400 tiles): every wave ONE item, the dispatcher hands the workgroups out -- + 1 - 4 % at 4 .. 64 samples per tile there,
    // - 1 - 2 % at 1080p (profiles/r05_tiles_first.log; the film-major launch has the same rule, launch_accumulate)
    const bool big = a.n_tiles >= 16384;
    const int grid = grid_for(items * 64, a.wg_per_cu > 0 ? 256 * a.wg_per_cu : big ? (1 << 30) : 256 * 8);
    // (the DMA walk needs the vector path: a.vec; the scalar path of unaligned images never touches the ring)
    if (a.vec && a.dma && a.umul == 2) hipLaunchKernelGGL((accumulate_tiles_kernel<2, kAccTilesDmaD>), dim3(grid), dim3(kBlock), acc_lds_bytes(kAccTilesDmaD), s, a);
    else if (a.vec && a.dma) hipLaunchKernelGGL((accumulate_tiles_kernel<1, kAccTilesDmaD>), dim3(grid), dim3(kBlock), acc_lds_bytes(kAccTilesDmaD), s, a);
    else if (a.umul == 2) hipLaunchKernelGGL((accumulate_tiles_kernel<2, 0>), dim3(grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((accumulate_tiles_kernel<1, 0>), dim3(grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ merge tiles
// AoS StatTilePixel<T> (estimator.h:104-124): uint64 n at byte 0, then mean, m2, m3, filmMean,
// filmM2 (T each) from byte 8; sizeof = 64 (T = float) / 128 (T = Vec3).
template <int C>
__global__ __launch_bounds__(kBlock) void merge_tiles_kernel(MergeTilesArgs a) {
    const int t = blockIdx.y;
    const int x0 = a.tile_bounds[4 * t], y0 = a.tile_bounds[4 * t + 1];
    const int x1 = a.tile_bounds[4 * t + 2], y1 = a.tile_bounds[4 * t + 3];
    const int tw = x1 - x0, npx = tw * (y1 - y0);
    constexpr int kStride = C == 1 ? 64 : 128;
    const unsigned char *base = static_cast<const unsigned char *>(a.tile_pixels) + a.tile_offsets[t] * kStride;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < npx; i += gridDim.x * kBlock) {
        const int x = x0 + i % tw, y = y0 + i / tw;  // estimator.h:44-48
        if (x >= a.width || y >= a.height) continue;
        const long long off = (long long)y * a.width + x;
        const unsigned char *px = base + (long long)i * kStride;
        const unsigned long long cnt = *reinterpret_cast<const unsigned long long *>(px);
        const float *f = reinterpret_cast<const float *>(px + 8);
        a.n[off] = (int32_t)cnt;
#pragma unroll
        for (int c = 0; c < C; c++) {
            a.mean[off * C + c] = f[c];
            a.m2[off * C + c] = f[C + c];
            a.m3[off * C + c] = f[2 * C + c];
            if (a.transform) {  // MergeTransformTile only (estimator.cpp:385-386)
                a.film_mean[off * C + c] = f[3 * C + c];
                a.film_m2[off * C + c] = f[4 * C + c];
            }
        }
    }
}

hipError_t launch_merge_tiles(const MergeTilesArgs &a, int n_tiles, int max_tile_pixels, hipStream_t s) {
    const dim3 grid((max_tile_pixels + kBlock - 1) / kBlock, n_tiles);
    if (a.channels == 3)
        hipLaunchKernelGGL(merge_tiles_kernel<3>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(merge_tiles_kernel<1>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ film update
// One lane per Film::Pixel (32 B AoS = two 16-B loads), interleaved RGB out.  film.cpp:188-222.
__device__ __forceinline__ void xyz_to_rgb(const float xyz[3], float rgb[3]) {  // spectrum.h:66-70
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

__global__ __launch_bounds__(kBlock) void film_update_kernel(const float4 *pixels, long long n, float splat_scale,
                                                             float scale, float *rgb_out) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float4 a = pixels[2 * i], b = pixels[2 * i + 1];  // {xyz, weight}, {splat xyz, pad}
        const float xyz[3] = {a.x, a.y, a.z}, sxyz[3] = {b.x, b.y, b.z};
        float o[3], s[3];
        xyz_to_rgb(xyz, o);
        if (a.w != 0.f) {
            const float inv = 1.f / a.w;
#pragma unroll
            for (int c = 0; c < 3; c++) o[c] = __builtin_fmaxf(0.f, o[c] * inv);
        }
        xyz_to_rgb(sxyz, s);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            o[c] += splat_scale * s[c];
            o[c] *= scale;
            rgb_out[3 * i + c] = o[c];
        }
    }
}

hipError_t launch_film_update(const void *pixels, long long n, float splat_scale, float scale, float *rgb, hipStream_t s) {
    hipLaunchKernelGGL(film_update_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, static_cast<const float4 *>(pixels), n,
                       splat_scale, scale, rgb);
    return hipGetLastError();
}

// ------------------------------------------------------------------ tile moments
// One wavefront per (tile, channel): each lane folds its share of the tile's pixels with
// Welford's update, then the 64 partial (count, mean, M2) triples are merged pairwise with
// Chan's formula through wave shuffles.
__device__ __forceinline__ void chan_merge(float &na, float &ma, float &sa, float nb, float mb, float sb) {
    const float n = na + nb;
    if (n > 0.f) {
        const float d = mb - ma;
        const float f = nb / n;
        ma = ma + d * f;
        sa = sa + sb + d * d * na * f;
        na = n;
    }
}

__global__ __launch_bounds__(64) void tile_moments_kernel(TileMomentsArgs a) {
    const int tx = blockIdx.x, ty = blockIdx.y, c = blockIdx.z;
    const int lane = threadIdx.x;
    const int ts = a.tile_size, tpx = ts * ts;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int i = lane; i < tpx; i += 64) {
        const int x = tx * ts + i % ts, y = ty * ts + i / ts;
        if (x < a.width && y < a.height) {
            const float v = a.values[((long long)y * a.width + x) * a.channels + c];
            cnt += 1.f;
            const float d = v - mean;
            mean += d / cnt;
            m2 += d * (v - mean);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float nb = __shfl_xor(cnt, off, 64), mb = __shfl_xor(mean, off, 64), sb = __shfl_xor(m2, off, 64);
        chan_merge(cnt, mean, m2, nb, mb, sb);
    }
    if (lane == 0) {
        float *o = a.out + (((long long)ty * a.tiles_x + tx) * a.channels + c) * 3;
        o[0] = cnt;
        o[1] = mean;
        o[2] = m2;
    }
}

hipError_t launch_tile_moments(const TileMomentsArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(tile_moments_kernel, dim3(a.tiles_x, a.tiles_y, a.channels), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace statmc
