/* statmc_oracle.c -- see statmc_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  Contraction is disabled
 * so that every fp32 operation below rounds exactly once, in the order written; the HIP
 * kernels are built the same way, which is what makes the integer/compare parts of the path
 * (membership decisions) bit-identical between the two.
 */
#include "../include/statmc_pinned_spec.h"
#include "statmc_oracle.h"
#include "t_quantiles_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* FP contraction mode of the restated reference arithmetic.
 *   0 (default): every multiply and add rounds separately -- what g++ emits for the reference (gcc's
 *      -ffp-contract=fast only acts with -march flags that have FMA; the survey's probe build was plain -O2).
 *   1: what the reference's OWN recipe emits on an FMA-capable host: scripts/_build.sh:14-19,37-41 builds with
 *      clang++ -O3 -march=native, and clang's default -ffp-contract=on fuses a multiply with the add or subtract
 *      that consumes it inside ONE source expression.  In StatTile<Float> (estimator.h:173-205,217-225) that is
 *      `m2 += d * (d - dN)` -> fma(d, d - dN, m2), `m3 += -3.f * dN * m2 + d * (d2 - dN2)` ->
 *      m3 + fma(-3.f * dN, m2, d * (d2 - dN2)), `filmM2 += filmD * (filmD - filmDN)` -> fma(...); in
 *      StatTile<Vec3> the multiply and the add sit in different inlined operator functions (cv::Vec operators,
 *      estimator.h:127-133), which clang's source-level contraction does not fuse.  XYZToRGB (spectrum.h:66-70)
 *      and `rgb += splatScale * splatRGB` in Film::UpdateImage (film.cpp:194,211-213) fuse as well.
 *      Checked against AMD clang 22 (-O3 -march=x86-64-v3) on a restatement of those expressions: the emitted
 *      vfmadd instructions are exactly the ones written below. */
static int g_fp_contract = 0;
void oracle_set_fp_contract(int on) { g_fp_contract = on ? 1 : 0; }
int oracle_get_fp_contract(void) { return g_fp_contract; }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* estimator.h:135-137 -- (std::pow(val, lambda) - 1.f) / lambda */
float oracle_box_cox(float v, float lambda) { return (powf(v, lambda) - 1.f) / lambda; }

/* estimator.h:162-172 */
static inline void add_m1(uint64_t n, float *mean, float s) {
    const float d = s - *mean;
    const float dN = d / (float)n; /* Float / uint64_t: the integer converts to float */
    *mean += dN;
}
/* estimator.h:173-186 */
static inline void add_m2(uint64_t n, float *mean, float *m2, float s, int contract) {
    const float d = s - *mean;
    const float dN = d / (float)n;
    *mean += dN;
    if (contract)
        *m2 = fmaf(d, d - dN, *m2);
    else
        *m2 += d * (d - dN);
}
/* estimator.h:187-205; note m3 uses the already updated m2 */
static inline void add_m3(uint64_t n, float *mean, float *m2, float *m3, float s, int contract) {
    const float d = s - *mean;
    const float d2 = d * d;
    const float dN = d / (float)n;
    const float dN2 = dN * dN;
    *mean += dN;
    if (contract) {
        *m2 = fmaf(d, d - dN, *m2);
        *m3 += fmaf(-3.f * dN, *m2, d * (d2 - dN2));
    } else {
        *m2 += d * (d - dN);
        *m3 += -3.f * dN * *m2 + d * (d2 - dN2);
    }
}

/* `scalar`: the pixel belongs to a StatTile<Float> (the only tiles whose updates clang contracts) */
static inline void add_channel(uint64_t n, float *mean, float *m2, float *m3, float *fmean,
                               float *fm2, float s, int transform, int max_moment, int scalar) {
    const int contract = g_fp_contract && scalar;
    const float v = transform ? oracle_box_cox(s, .5f) : s; /* estimator.h:215 */
    if (max_moment >= 3)
        add_m3(n, mean, m2, m3, v, contract);
    else if (max_moment == 2)
        add_m2(n, mean, m2, v, contract);
    else
        add_m1(n, mean, v);
    if (transform) { /* estimator.h:217-225, n is the already incremented count */
        const float fd = s - *fmean;
        const float fdN = fd / (float)n;
        *fmean += fdN;
        if (contract)
            *fm2 = fmaf(fd, fd - fdN, *fm2);
        else
            *fm2 += fd * (fd - fdN);
    } else { /* estimator.h:209-210 */
        *fmean = *mean;
        *fm2 = *m2;
    }
}

void oracle_add_sample(void *px, int channels, const float *sample, int transform, int max_moment) {
    if (channels == 1) {
        oracle_tile_pixel_f1 *p = (oracle_tile_pixel_f1 *)px;
        p->n++;
        add_channel(p->n, &p->mean, &p->m2, &p->m3, &p->film_mean, &p->film_m2, sample[0],
                    transform, max_moment, 1);
    } else {
        oracle_tile_pixel_f3 *p = (oracle_tile_pixel_f3 *)px;
        p->n++;
        for (int c = 0; c < 3; c++)
            add_channel(p->n, &p->mean[c], &p->m2[c], &p->m3[c], &p->film_mean[c], &p->film_m2[c],
                        sample[c], transform, max_moment, 0);
    }
}

/* estimator.cpp:341-352 / 376-388 */
void oracle_merge_tile(const void *tile_pixels, int channels, int x0, int y0, int x1, int y1,
                       int width, int32_t *n, float *mean, float *m2, float *m3,
                       float *film_mean, float *film_m2) {
    const int tw = x1 - x0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const size_t off = (size_t)y * width + x;
            const size_t ti = (size_t)(y - y0) * tw + (x - x0); /* estimator.h:44-48 */
            if (channels == 1) {
                const oracle_tile_pixel_f1 *p = (const oracle_tile_pixel_f1 *)tile_pixels + ti;
                n[off] = (int32_t)p->n;
                mean[off] = p->mean;
                m2[off] = p->m2;
                m3[off] = p->m3;
                if (film_mean) film_mean[off] = p->film_mean;
                if (film_m2) film_m2[off] = p->film_m2;
            } else {
                const oracle_tile_pixel_f3 *p = (const oracle_tile_pixel_f3 *)tile_pixels + ti;
                n[off] = (int32_t)p->n;
                for (int c = 0; c < 3; c++) {
                    mean[3 * off + c] = p->mean[c];
                    m2[3 * off + c] = p->m2[c];
                    m3[3 * off + c] = p->m3[c];
                    if (film_mean) film_mean[3 * off + c] = p->film_mean[c];
                    if (film_m2) film_m2[3 * off + c] = p->film_m2[c];
                }
            }
        }
}

void oracle_accumulate_image(int width, int height, int channels, int transform, int max_moment,
                             int n_samples, const float *samples,
                             int32_t *n, float *mean, float *m2, float *m3,
                             float *film_mean, float *film_m2,
                             int tile_size, int threads) {
    if (tile_size <= 0) tile_size = 16; /* statpath.cpp:132 */
    const int ntx = (width + tile_size - 1) / tile_size;
    const int nty = (height + tile_size - 1) / tile_size;
    const size_t plane = (size_t)width * height * channels;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int t = 0; t < ntx * nty; t++) {
        const int x0 = (t % ntx) * tile_size, y0 = (t / ntx) * tile_size;
        const int x1 = x0 + tile_size < width ? x0 + tile_size : width;
        const int y1 = y0 + tile_size < height ? y0 + tile_size : height;
        const int tw = x1 - x0, th = y1 - y0;
        const size_t psz = channels == 1 ? sizeof(oracle_tile_pixel_f1) : sizeof(oracle_tile_pixel_f3);
        unsigned char *tile = (unsigned char *)aligned_alloc(64, psz * tw * th);
        /* load the persistent per-pixel state into the tile (tiles live across iterations) */
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const size_t off = (size_t)y * width + x;
                void *px = tile + psz * ((size_t)(y - y0) * tw + (x - x0));
                if (channels == 1) {
                    oracle_tile_pixel_f1 *p = (oracle_tile_pixel_f1 *)px;
                    p->n = (uint64_t)n[off];
                    p->mean = mean[off]; p->m2 = m2[off]; p->m3 = m3[off];
                    p->film_mean = film_mean[off]; p->film_m2 = film_m2[off];
                } else {
                    oracle_tile_pixel_f3 *p = (oracle_tile_pixel_f3 *)px;
                    p->n = (uint64_t)n[off];
                    for (int c = 0; c < 3; c++) {
                        p->mean[c] = mean[3 * off + c]; p->m2[c] = m2[3 * off + c];
                        p->m3[c] = m3[3 * off + c];
                        p->film_mean[c] = film_mean[3 * off + c]; p->film_m2[c] = film_m2[3 * off + c];
                    }
                }
            }
        /* pixel-major, then sample order -- statpath.cpp:255,294-375 */
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                void *px = tile + psz * ((size_t)(y - y0) * tw + (x - x0));
                for (int s = 0; s < n_samples; s++)
                    oracle_add_sample(px, channels,
                                      samples + (size_t)s * plane + ((size_t)y * width + x) * channels,
                                      transform, max_moment);
            }
        oracle_merge_tile(tile, channels, x0, y0, x1, y1, width, n, mean, m2, m3, film_mean, film_m2);
        free(tile);
    }
}

/* The same accumulation with the samples laid out the way StatPathIntegrator::Render produces them: tile after tile
 * (statpath.cpp:132, 16 x 16), inside a tile pixel after pixel (statpath.cpp:255), every pixel's samples one after the other
 * (statpath.cpp:294-375) -- `samples` is [tile][pixel][n_samples][channels], tiles in row-major order over the film, pixels in
 * row-major order inside the tile (ragged edge tiles hold their own pixel count).  The reference never stores a sample: it
 * hands each one to AddSample as it is traced, so its per-sample memory traffic is nil; a CPU timing of this path should read
 * the samples as a stream, which this layout gives, and not with a cache miss per sample, which the film-major planes of
 * oracle_accumulate_image cost a CPU (one plane of the film between a pixel's consecutive samples).  Same arithmetic, same
 * order per pixel: the two entry points leave the same bits (tests/test_oracle_cpu.py). */
void oracle_accumulate_tile_stream(int width, int height, int channels, int transform, int max_moment,
                                   int n_samples, const float *samples,
                                   int32_t *n, float *mean, float *m2, float *m3,
                                   float *film_mean, float *film_m2,
                                   int tile_size, int threads) {
    if (tile_size <= 0) tile_size = 16;
    const int ntx = (width + tile_size - 1) / tile_size;
    const int nty = (height + tile_size - 1) / tile_size;
    /* offset of every tile's block, in pixels (x n_samples x channels floats) */
    size_t *first = (size_t *)malloc(sizeof(size_t) * ((size_t)ntx * nty + 1));
    first[0] = 0;
    for (int t = 0; t < ntx * nty; t++) {
        const int x0 = (t % ntx) * tile_size, y0 = (t / ntx) * tile_size;
        const int tw = (x0 + tile_size < width ? x0 + tile_size : width) - x0;
        const int th = (y0 + tile_size < height ? y0 + tile_size : height) - y0;
        first[t + 1] = first[t] + (size_t)tw * th;
    }
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int t = 0; t < ntx * nty; t++) {
        const int x0 = (t % ntx) * tile_size, y0 = (t / ntx) * tile_size;
        const int x1 = x0 + tile_size < width ? x0 + tile_size : width;
        const int y1 = y0 + tile_size < height ? y0 + tile_size : height;
        const int tw = x1 - x0, th = y1 - y0;
        const size_t psz = channels == 1 ? sizeof(oracle_tile_pixel_f1) : sizeof(oracle_tile_pixel_f3);
        unsigned char *tile = (unsigned char *)aligned_alloc(64, psz * tw * th);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const size_t off = (size_t)y * width + x;
                void *px = tile + psz * ((size_t)(y - y0) * tw + (x - x0));
                if (channels == 1) {
                    oracle_tile_pixel_f1 *p = (oracle_tile_pixel_f1 *)px;
                    p->n = (uint64_t)n[off];
                    p->mean = mean[off]; p->m2 = m2[off]; p->m3 = m3[off];
                    p->film_mean = film_mean[off]; p->film_m2 = film_m2[off];
                } else {
                    oracle_tile_pixel_f3 *p = (oracle_tile_pixel_f3 *)px;
                    p->n = (uint64_t)n[off];
                    for (int c = 0; c < 3; c++) {
                        p->mean[c] = mean[3 * off + c]; p->m2[c] = m2[3 * off + c];
                        p->m3[c] = m3[3 * off + c];
                        p->film_mean[c] = film_mean[3 * off + c]; p->film_m2[c] = film_m2[3 * off + c];
                    }
                }
            }
        const float *sp = samples + first[t] * (size_t)n_samples * channels;
        for (int i = 0; i < tw * th; i++) {
            void *px = tile + psz * (size_t)i;
            for (int s = 0; s < n_samples; s++, sp += channels)
                oracle_add_sample(px, channels, sp, transform, max_moment);
        }
        oracle_merge_tile(tile, channels, x0, y0, x1, y1, width, n, mean, m2, m3, film_mean, film_m2);
        free(tile);
    }
    free(first);
}

/* estimator.cpp:524-568 */
void oracle_mean_vars(int width, int height, int channels, const int32_t *n,
                      const float *film_m2, float *film_mean_var, int row_n_quirk) {
    for (int row = 0; row < height; row++) {
        const float n_row = (float)n[(size_t)row * width];
        for (int col = 0; col < width; col++) {
            const size_t off = (size_t)row * width + col;
            const float nf = row_n_quirk ? n_row : (float)n[off];
            for (int c = 0; c < channels; c++)
                film_mean_var[off * channels + c] = film_m2[off * channels + c] / ((nf - 1.f) * nf);
        }
    }
}

/* spectrum.h:66-70 */
static inline void xyz_to_rgb(const float xyz[3], float rgb[3]) {
    if (g_fp_contract) { /* a*x - b*y - c*z as clang fuses it: fma(z, -c, fma(x, a, -b * y)) */
        rgb[0] = fmaf(xyz[2], -0.498535f, fmaf(xyz[0], 3.240479f, -1.537150f * xyz[1]));
        rgb[1] = fmaf(xyz[2], 0.041556f, fmaf(xyz[0], -0.969256f, 1.875991f * xyz[1]));
        rgb[2] = fmaf(xyz[2], 1.057311f, fmaf(xyz[0], 0.055648f, -0.204043f * xyz[1]));
        return;
    }
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

/* film.cpp:188-222 */
void oracle_film_update(const oracle_film_pixel *pixels, size_t n_pixels, float splat_scale, float scale, float *rgb) {
    for (size_t i = 0; i < n_pixels; i++) {
        const oracle_film_pixel *p = &pixels[i];
        float *o = rgb + 3 * i;
        xyz_to_rgb(p->xyz, o);
        if (p->filter_weight_sum != 0) {
            const float inv = 1.f / p->filter_weight_sum;
            for (int c = 0; c < 3; c++) o[c] = fmaxf(0.f, o[c] * inv);
        }
        float splat_rgb[3];
        xyz_to_rgb(p->splat_xyz, splat_rgb);
        for (int c = 0; c < 3; c++) {
            o[c] = g_fp_contract ? fmaf(splat_scale, splat_rgb[c], o[c]) : o[c] + splat_scale * splat_rgb[c];
            o[c] *= scale;
        }
    }
}

/* ---------------------------- tile-local pooled moments ----------------------------------
 * No reference function: BASELINE.json's north_star asks for "wavefront-level Welford reductions for tile-local
 * variance"; the product's tile_moments_kernel (statmc_pointwise.hip) gives every lane of a 64-wide wave the pixels
 * lane, lane + 64, ... of a tile (Welford, in that order) and merges the 64 partial (count, mean, M2) triples with
 * Chan's formula in a butterfly of xor-shuffles (offsets 32, 16, ..., 1); lane 0 holds the tile's result.  Restated
 * here lane by lane, level by level, so that the comparison is bit for bit. */
static void chan_merge(float *na, float *ma, float *sa, float nb, float mb, float sb) {
    const float n = *na + nb;
    if (n > 0.f) {
        const float d = mb - *ma;
        const float f = nb / n;
        *ma = *ma + d * f;
        *sa = *sa + sb + d * d * *na * f;
        *na = n;
    }
}

void oracle_tile_moments(int width, int height, int channels, const float *values, int tile_size, float *out) {
    const int tiles_x = (width + tile_size - 1) / tile_size, tiles_y = (height + tile_size - 1) / tile_size;
    const int tpx = tile_size * tile_size;
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++)
            for (int c = 0; c < channels; c++) {
                float cnt[64], mean[64], m2[64];
                for (int lane = 0; lane < 64; lane++) {
                    cnt[lane] = mean[lane] = m2[lane] = 0.f;
                    for (int i = lane; i < tpx; i += 64) {
                        const int x = tx * tile_size + i % tile_size, y = ty * tile_size + i / tile_size;
                        if (x < width && y < height) {
                            const float v = values[((size_t)y * width + x) * channels + c];
                            cnt[lane] += 1.f;
                            const float d = v - mean[lane];
                            mean[lane] += d / cnt[lane];
                            m2[lane] += d * (v - mean[lane]);
                        }
                    }
                }
                for (int off = 32; off >= 1; off >>= 1) {
                    float nb[64], mb[64], sb[64];
                    for (int lane = 0; lane < 64; lane++) { nb[lane] = cnt[lane ^ off]; mb[lane] = mean[lane ^ off]; sb[lane] = m2[lane ^ off]; }
                    for (int lane = 0; lane < 64; lane++) chan_merge(&cnt[lane], &mean[lane], &m2[lane], nb[lane], mb[lane], sb[lane]);
                }
                float *o = out + (((size_t)ty * tiles_x + tx) * channels + c) * 3;
                o[0] = cnt[0];
                o[1] = mean[0];
                o[2] = m2[0];
            }
}

/* ---------------------------- filter spec v2 -------------------------------------------
 * The reference's arithmetic for this half is not in the tree (header comment of statmc_oracle.h).  What the
 * tree leaves open (SURVEY.md App. B "Unknown") is carried by oracle_filter_spec; the all-zero spec is this
 * build's default. */

void oracle_default_spec(oracle_filter_spec *s) {   /* include/statmc_pinned_spec.h: all zero until tools/pin_from_dumps.sh has run */
    static const oracle_filter_spec pinned = STATMC_PINNED_SPEC;
    *s = pinned;
}
int oracle_default_significance(void) { return STATMC_PINNED_SIGNIFICANCE; }

static float tq_override[ORACLE_TQ_N_TABLES][ORACLE_TQ_N_DOF];
static int tq_overridden[ORACLE_TQ_N_TABLES];

void oracle_set_t_quantiles(int table, const float *q, int n_dof) {
    if (!q) {  /* back to the built-in table */
        tq_overridden[table] = 0;
        return;
    }
    for (int i = 0; i < ORACLE_TQ_N_DOF; i++) tq_override[table][i] = q[i < n_dof ? i : n_dof - 1];
    tq_overridden[table] = 1;
}

/* table = alpha_index (0..2) + 3 * sides */
float oracle_t_quantile(int table, int dof) {
    if (dof < 1) return INFINITY;
    if (dof > ORACLE_TQ_N_DOF) dof = ORACLE_TQ_N_DOF;
    return tq_overridden[table] ? tq_override[table][dof - 1] : oracle_tq_tables[table][dof - 1];
}

void oracle_prepass_spec(int width, int height, int channels, int alpha_index, const oracle_filter_spec *spec,
                         const int32_t *n, const float *mean, const float *m2, const float *m3,
                         float *mean_corr, float *discriminator) {
    const size_t npx = (size_t)width * height;
    const int table = alpha_index + ORACLE_TQ_N_ALPHAS * (spec->sides ? 1 : 0);
    for (size_t i = 0; i < npx; i++) {
        const int32_t ni = n[i];
        const float nf = (float)ni;
        /* Welch mode keeps the quantile out of the per-pixel term: the pair looks it up at its own dof */
        const float t = spec->dof == ORACLE_DOF_WELCH ? 1.f : oracle_t_quantile(table, ni - 1);
        for (int c = 0; c < channels; c++) {
            const size_t e = i * channels + c;
            const float mu = mean[e], s2sum = m2[e];
            if (ni >= 2 && s2sum > 0.f) {
                const float var = s2sum / (nf - 1.f);  /* unbiased sample variance s^2 */
                const float mu3 = m3[e] / nf;          /* third central sample moment */
                mean_corr[e] = mu + mu3 / (6.f * var * nf); /* Johnson (1978) */
                discriminator[e] = (t * t) * (var / nf);    /* squared CI half-width (Welch: var / n) */
            } else if (ni < 2 && spec->small_n == ORACLE_SMALL_N_EXCLUDE) {
                mean_corr[e] = NAN;  /* the pixel takes no part in any window */
                discriminator[e] = NAN;
            } else {
                mean_corr[e] = mu;
                discriminator[e] = ni >= 2 ? 0.f : INFINITY;
            }
        }
    }
}

void oracle_prepass(int width, int height, int channels, int alpha_index,
                    const int32_t *n, const float *mean, const float *m2, const float *m3,
                    float *mean_corr, float *discriminator) {
    oracle_filter_spec spec;
    oracle_default_spec(&spec);
    oracle_prepass_spec(width, height, channels, alpha_index, &spec, n, mean, m2, m3, mean_corr, discriminator);
}

/* a pixel takes part in windows (as centre and as tap) when its corrected mean is finite in every channel, its
 * discriminator is not NaN, its colour is finite (v2: a NaN / inf colour would otherwise spread to every
 * window that accepts the pixel; the reference never produces one -- statpath.cpp:333-351 blacks such samples) and
 * -- v2.1, round 4 -- every G-buffer value of the pixel is finite (a NaN feature makes the range weight of every pair
 * with the pixel NaN: the same spreading, through the weight instead of the colour) */
static int pixel_valid(int channels, const float *mc, const float *disc, const float *colour, int n_g,
                       const float *const *g_buffers, const int *g_channels, size_t p) {
    int v = 1;
    for (int c = 0; c < channels; c++) {
        v &= isfinite(mc[p * channels + c]) != 0;
        v &= !isnan(disc[p * channels + c]);
        v &= isfinite(colour[p * channels + c]) != 0;
    }
    for (int g = 0; g < n_g; g++)
        for (int c = 0; c < g_channels[g]; c++) v &= isfinite(g_buffers[g][p * g_channels[g] + c]) != 0;
    return v;
}

/* membership of the pair (p, q); both already known to be valid pixels */
static int pair_member(const oracle_filter_spec *spec, int channels, int table, const float *mc, const float *disc,
                       const int32_t *n, size_t p, size_t q) {
    int all = 1;
    float lhs_sum = 0.f, rhs_sum = 0.f;
    for (int c = 0; c < channels; c++) {
        const float d = mc[p * channels + c] - mc[q * channels + c];
        const float Dp = disc[p * channels + c], Dq = disc[q * channels + c];
        float lhs, rhs;
        if (spec->dof == ORACLE_DOF_WELCH) {
            /* disc holds v = s^2 / n; the pair's quantile is looked up at the Welch-Satterthwaite dof */
            const float s = Dp + Dq;
            float Dsum = s;  /* 0 (both variances zero) and +inf (a pixel with fewer than two samples) stay */
            if (s > 0.f && isfinite(s)) {
                const float nu = (s * s) / (Dp * Dp / ((float)n[p] - 1.f) + Dq * Dq / ((float)n[q] - 1.f));
                int dof = nu >= 1.f ? (nu < (float)ORACLE_TQ_N_DOF ? (int)nu : ORACLE_TQ_N_DOF) : 1;
                const float t = oracle_t_quantile(table, dof);
                Dsum = (t * t) * s;
            }
            lhs = fmaf(d, d, -Dsum);
            rhs = 0.f;
        } else if (spec->gate == ORACLE_GATE_CENTRE) {
            lhs = d * d;           /* Moon et al. 2013 (the reference's -DMEMFNC=1): q's mean inside p's confidence interval */
            rhs = Dp;
        } else if (spec->gate == ORACLE_GATE_ASYMMETRIC) {
            lhs = fmaf(d, d, -Dq); /* spec v1.x */
            rhs = Dp;
        } else {
            lhs = fmaf(d, d, -(Dp + Dq)); /* d^2 <= D_p + D_q, the same bits for (p, q) and (q, p) */
            rhs = 0.f;
        }
        all &= (lhs <= rhs);
        lhs_sum = c == 0 ? lhs : lhs_sum + lhs;
        rhs_sum = c == 0 ? rhs : rhs_sum + rhs;
    }
    return spec->channel_rule == ORACLE_CHANNELS_JOINT ? (lhs_sum <= rhs_sum) : all;
}

void oracle_filter_spec_run(int width, int height, int channels, float ds, int radius, int alpha_index,
                            const oracle_filter_spec *spec, const int32_t *n,
                            const float *mean_corr, const float *disc, const float *colour,
                            int n_g, const float *const *g_buffers, const int *g_channels, const float *g_dr,
                            float *out, int rx0, int ry0, int rx1, int ry1, int threads) {
    const int table = alpha_index + ORACLE_TQ_N_ALPHAS * (spec->sides ? 1 : 0);
    const int clamp = spec->border == ORACLE_BORDER_CLAMP;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int y = ry0; y < ry1; y++) {
        for (int x = rx0; x < rx1; x++) {
            const size_t p = (size_t)y * width + x;
            float sum_w = 0.f, acc[3] = {0.f, 0.f, 0.f};
            const int p_valid = pixel_valid(channels, mean_corr, disc, colour, n_g, g_buffers, g_channels, p);
            for (int dy = -radius; p_valid && dy <= radius; dy++) {
                int qy = y + dy;
                if (qy < 0 || qy >= height) {
                    if (!clamp) continue;
                    qy = qy < 0 ? 0 : height - 1;
                }
                for (int dx = -radius; dx <= radius; dx++) {
                    int qx = x + dx;
                    if (qx < 0 || qx >= width) {
                        if (!clamp) continue;
                        qx = qx < 0 ? 0 : width - 1;
                    }
                    const size_t q = (size_t)qy * width + qx;
                    if (!pixel_valid(channels, mean_corr, disc, colour, n_g, g_buffers, g_channels, q)) continue;
                    if (!pair_member(spec, channels, table, mean_corr, disc, n, p, q)) continue;
                    float e = ds * (float)(dx * dx + dy * dy);
                    for (int g = 0; g < n_g; g++) {
                        const int gc = g_channels[g];
                        const float *G = g_buffers[g];
                        float d0 = G[p * gc] - G[q * gc];
                        float dist2 = d0 * d0;
                        for (int c = 1; c < gc; c++) {
                            const float dc = G[p * gc + c] - G[q * gc + c];
                            dist2 = fmaf(dc, dc, dist2);
                        }
                        e = fmaf(g_dr[g], dist2, e);
                    }
                    const float w = expf(e);
                    sum_w += w;
                    for (int c = 0; c < channels; c++)
                        acc[c] = fmaf(w, colour[q * channels + c], acc[c]);
                }
            }
            for (int c = 0; c < channels; c++)
                out[p * channels + c] = sum_w > 0.f ? acc[c] / sum_w : colour[p * channels + c];
        }
    }
}

void oracle_filter(int width, int height, int channels, float ds, int radius,
                   const float *mean_corr, const float *disc, const float *colour,
                   int n_g, const float *const *g_buffers, const int *g_channels, const float *g_dr,
                   float *out, int rx0, int ry0, int rx1, int ry1, int threads) {
    oracle_filter_spec spec;
    oracle_default_spec(&spec);
    oracle_filter_spec_run(width, height, channels, ds, radius, 0, &spec, NULL, mean_corr, disc, colour, n_g, g_buffers,
                           g_channels, g_dr, out, rx0, ry0, rx1, ry1, threads);
}
