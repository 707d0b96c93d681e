/* statmc_oracle.h -- CPU restatement of StatMC's statistics hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (statmc_amd/, include/) may include, link
 * or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - accumulate / merge / mean-vars: restated from /root/reference/src/statistics/estimator.h
 *     and estimator.cpp (line numbers at each function).  The reference cannot be compiled in
 *     this image without writing stand-ins for OpenCV and glog headers, so it is treated as
 *     unbuildable; the restatement is pinned by the single known-answer vector recorded in
 *     SURVEY.md Appendix A (produced by the survey's probe build of the reference header) and
 *     is otherwise PARITY UNPINNED.
 *   - pre-pass / filter: the reference's arithmetic lives in the un-vendored submodule
 *     cg-tuwien/StatMC-opencv_contrib (modules/cudaimgproc/src/cuda/stat_denoiser.cu, pinned
 *     version unknown, README says OpenCV 4.8.1).  Only its call sites are in the tree
 *     (estimator.cpp:437-487).  This file restates the published algorithm as the build's own
 *     spec (DESIGN.md "Filter spec v2"; every choice the tree leaves open is an option of
 *     oracle_filter_spec): PARITY UNPINNED.
 */
#ifndef STATMC_ORACLE_H
#define STATMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1: StatTilePixel<T>  (estimator.h:104-124, float build) --------------------------
 * {uint64 n; T mean, m2, m3, filmMean, filmM2} __attribute__((aligned(64))).
 * sizeof = 64 for T=float, 128 for T=Vec3. */
typedef struct __attribute__((aligned(64))) {
    uint64_t n;
    float mean, m2, m3, film_mean, film_m2;
} oracle_tile_pixel_f1;

typedef struct __attribute__((aligned(64))) {
    uint64_t n;
    float mean[3], m2[3], m3[3], film_mean[3], film_m2[3];
} oracle_tile_pixel_f3;

/* a3: boxCox(val, lambda) = (pow(val, lambda) - 1) / lambda   (estimator.h:135-145) */
float oracle_box_cox(float v, float lambda);

/* a4/a5: one sample into one AoS pixel.  channels = 1 or 3; px points at an
 * oracle_tile_pixel_f1 / _f3.  transform != 0 -> AddTransformSample (estimator.h:212-226),
 * else AddSample (estimator.h:206-211).  max_moment in {1,2,3} selects
 * AddStatSampleM1/M2/M3 (estimator.h:162-205). */
void oracle_add_sample(void *px, int channels, const float *sample, int transform, int max_moment);

/* a2/a6/a7 in one go: run a sample stream through 16x16 StatTiles and scatter them into
 * full-frame images, exactly as Render<T> + Merge[Transform]Tile do
 * (statpath.cpp:355-371, estimator.cpp:341-407).
 *   samples : [n_samples][height][width][channels] fp32
 *   images  : n (int32, 1 ch), mean/m2/m3/film_mean/film_m2 ([height][width][channels]);
 *             the state in the images is loaded into the tiles first, so calls chain the way
 *             tiles persist across iterations (statpath.cpp:173-190).
 * For non-transform types film_mean/film_m2 receive copies of mean/m2 (estimator.h:209-210).
 * threads <= 0 -> all cores (OpenMP over tiles, mirrors ParallelFor2D, statpath.cpp:218). */
void oracle_accumulate_image(int width, int height, int channels, int transform, int max_moment,
                             int n_samples, const float *samples,
                             int32_t *n, float *mean, float *m2, float *m3,
                             float *film_mean, float *film_m2,
                             int tile_size, int threads);
/* the same with samples laid out [tile][pixel][n_samples][channels] (the order Render<T> produces them in) */
void oracle_accumulate_tile_stream(int width, int height, int channels, int transform, int max_moment,
                             int n_samples, const float *samples,
                             int32_t *n, float *mean, float *m2, float *m3,
                             float *film_mean, float *film_m2,
                             int tile_size, int threads);

/* a7 alone: scatter one AoS tile (tile pixel bounds [x0,x1) x [y0,y1)) into the images
 * (estimator.cpp:341-352 MergeTile, 376-388 MergeTransformTile).  film_* may be NULL
 * (MergeTile does not write them). */
void oracle_merge_tile(const void *tile_pixels, int channels, int x0, int y0, int x1, int y1,
                       int width, int32_t *n, float *mean, float *m2, float *m3,
                       float *film_mean, float *film_m2);

/* a14: Estimator::CalculateMeanVars (estimator.cpp:524-568):
 * var = film_m2 / ((n-1)*n).  row_n_quirk != 0 reproduces the reference reading n once per
 * row (first pixel of the row) -- estimator.cpp:540,558. */
void oracle_mean_vars(int width, int height, int channels, const int32_t *n,
                      const float *film_m2, float *film_mean_var, int row_n_quirk);

/* a15: Film::UpdateImage (src/core/film.cpp:188-222) over the Film::Pixel array
 * {float xyz[3]; float filterWeightSum; float splatXYZ[3]; float pad} (film.h:72-78, 32 B):
 * XYZ -> RGB (spectrum.h:66-70), / weight sum, clamp >= 0, + splatScale * splat RGB, * scale. */
typedef struct {
    float xyz[3];
    float filter_weight_sum;
    float splat_xyz[3];
    float pad;
} oracle_film_pixel;
void oracle_film_update(const oracle_film_pixel *pixels, size_t n_pixels, float splat_scale, float scale, float *rgb);

/* FP contraction of the restated reference arithmetic: 0 (default) = every operation rounds on its own (a g++
 * build of the reference); 1 = the fused multiply-adds clang -O3 -march=native (the reference's own recipe,
 * scripts/_build.sh:14-19,37-41; -ffp-contract=on is clang's default) emits in StatTile<Float> and
 * Film::UpdateImage.  See statmc_oracle.c. */
void oracle_set_fp_contract(int on);
int oracle_get_fp_contract(void);

/* ---- filter spec v2 (self-specified; see header comment) --------------------------------
 * Everything SURVEY.md App. B lists as unknown about cv::cuda::stat_denoiser::filter<T> is a field here, mirrored
 * by statmc_filter_spec of the C ABI (include/statmc.h).  All-zero = this build's default. */
enum { ORACLE_GATE_SYMMETRIC = 0, ORACLE_GATE_ASYMMETRIC = 1, ORACLE_GATE_CENTRE = 2 };
enum { ORACLE_CHANNELS_AND = 0, ORACLE_CHANNELS_JOINT = 1 };
enum { ORACLE_SIDES_TWO = 0, ORACLE_SIDES_ONE = 1 };
enum { ORACLE_DOF_PIXEL = 0, ORACLE_DOF_WELCH = 1 };
enum { ORACLE_BORDER_CLIP = 0, ORACLE_BORDER_CLAMP = 1 };
enum { ORACLE_SMALL_N_ACCEPT = 0, ORACLE_SMALL_N_EXCLUDE = 1 };
typedef struct {
    int32_t gate;         /* SYMMETRIC: fma(d, d, -(D_p + D_q)) <= 0;  ASYMMETRIC (spec v1): fma(d, d, -D_q) <= D_p;
                             CENTRE: d * d <= D_p -- the neighbour's mean inside the CENTRE pixel's confidence interval, its own
                             interval playing no part: the membership of Moon et al. 2013, which the reference's CUDA source
                             switches to under -DMEMFNC=1 (README.md:147-150) */
    int32_t channel_rule; /* AND: every channel passes;  JOINT: sum_c lhs_c <= sum_c rhs_c */
    int32_t sides;        /* TWO: t_{1-alpha/2};  ONE: t_{1-alpha} */
    int32_t dof;          /* PIXEL: D = t(n-1)^2 s^2/n per pixel;  WELCH: the discriminator image holds s^2/n and the
                             pair takes t at floor(Welch-Satterthwaite dof); the gate form is then symmetric */
    int32_t border;       /* CLIP: taps outside the image are skipped;  CLAMP: their coordinates are clamped */
    int32_t small_n;      /* ACCEPT: n < 2 -> D = +inf (passes every test);  EXCLUDE: the pixel takes no part */
} oracle_filter_spec;
void oracle_default_spec(oracle_filter_spec *spec);   /* the pinned spec (include/statmc_pinned_spec.h) */
int oracle_default_significance(void);

/* tq(dof) = Student-t quantile of table `table` = alpha_index + 3 * sides (alpha_index in {0: 0.005, 1: 0.002,
 * 2: 0.05}); dof < 1 -> +inf; dof clamped to the last table entry. */
float oracle_t_quantile(int table, int dof);
/* user-supplied quantiles for dof 1..n_dof in table slot `table` (NULL restores the built-in table) */
void oracle_set_t_quantiles(int table, const float *quantiles, int n_dof);

/* Pre-pass: (n, mean, m2, m3) -> Johnson-corrected mean and discriminator, per channel. */
void oracle_prepass(int width, int height, int channels, int alpha_index,
                    const int32_t *n, const float *mean, const float *m2, const float *m3,
                    float *mean_corr, float *discriminator);

void oracle_prepass_spec(int width, int height, int channels, int alpha_index, const oracle_filter_spec *spec,
                         const int32_t *n, const float *mean, const float *m2, const float *m3,
                         float *mean_corr, float *discriminator);

/* Window filter of one buffer.
 *   channels          : 1 (filter<float>) or 3 (filter<float3>)
 *   mean_corr, disc   : pre-pass outputs, [h][w][channels]
 *   colour            : image being filtered ("film" or tX-bY-film-mean), [h][w][channels]
 *   g_buffers[i]      : [h][w][g_channels[i]] feature means; g_dr[i] = -0.5/sd_i^2
 *   ds                : -0.5/filtersd^2  (estimator.h:259)
 *   roi               : outputs are computed for x in [rx0,rx1), y in [ry0,ry1); the window is
 *                       clipped to the [0,w) x [0,h) image.
 * threads <= 0 -> all cores (OpenMP over rows). */
void oracle_filter(int width, int height, int channels, float ds, int radius,
                   const float *mean_corr, const float *disc, const float *colour,
                   int n_g, const float *const *g_buffers, const int *g_channels, const float *g_dr,
                   float *out, int rx0, int ry0, int rx1, int ry1, int threads);

/* The same under a non-default spec.  n (the sample counts) is read in Welch mode only and may be NULL otherwise;
 * alpha_index selects the quantile table of the Welch lookup. */
void oracle_filter_spec_run(int width, int height, int channels, float ds, int radius, int alpha_index,
                            const oracle_filter_spec *spec, const int32_t *n,
                            const float *mean_corr, const float *disc, const float *colour,
                            int n_g, const float *const *g_buffers, const int *g_channels, const float *g_dr,
                            float *out, int rx0, int ry0, int rx1, int ry1, int threads);

/* Tile-local pooled moments {count, mean, M2} per tile_size x tile_size tile and channel (out: [tiles_y][tiles_x]
 * [channels][3]); the lane order and merge tree of the product's wave-level kernel, restated (statmc_oracle.c). */
void oracle_tile_moments(int width, int height, int channels, const float *values, int tile_size, float *out);

int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
