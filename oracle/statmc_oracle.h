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
 *     frozen spec (DESIGN.md "Filter spec v1"): PARITY UNPINNED.
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

/* ---- filter spec v1 (self-specified; see header comment) -------------------------------- */

/* tq(dof) = two-sided Student-t quantile, table index alpha_index in {0: 0.005, 1: 0.002,
 * 2: 0.05}; dof < 1 -> +inf; dof clamped to the last table entry. */
float oracle_t_quantile(int alpha_index, int dof);
/* user-supplied quantiles for dof 1..n_dof in slot alpha_index (NULL restores the built-in table) */
void oracle_set_t_quantiles(int alpha_index, const float *quantiles, int n_dof);

/* Pre-pass: (n, mean, m2, m3) -> Johnson-corrected mean and discriminator, per channel. */
void oracle_prepass(int width, int height, int channels, int alpha_index,
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

int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
