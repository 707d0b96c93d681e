/* statmc_oracle.c -- see statmc_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  Contraction is disabled
 * so that every fp32 operation below rounds exactly once, in the order written; the HIP
 * kernels are built the same way, which is what makes the integer/compare parts of the path
 * (membership decisions) bit-identical between the two.
 */
#include "statmc_oracle.h"
#include "t_quantiles_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

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
static inline void add_m2(uint64_t n, float *mean, float *m2, float s) {
    const float d = s - *mean;
    const float dN = d / (float)n;
    *mean += dN;
    *m2 += d * (d - dN);
}
/* estimator.h:187-205; note m3 uses the already updated m2 */
static inline void add_m3(uint64_t n, float *mean, float *m2, float *m3, float s) {
    const float d = s - *mean;
    const float d2 = d * d;
    const float dN = d / (float)n;
    const float dN2 = dN * dN;
    *mean += dN;
    *m2 += d * (d - dN);
    *m3 += -3.f * dN * *m2 + d * (d2 - dN2);
}

static inline void add_channel(uint64_t n, float *mean, float *m2, float *m3, float *fmean,
                               float *fm2, float s, int transform, int max_moment) {
    const float v = transform ? oracle_box_cox(s, .5f) : s; /* estimator.h:215 */
    if (max_moment >= 3)
        add_m3(n, mean, m2, m3, v);
    else if (max_moment == 2)
        add_m2(n, mean, m2, v);
    else
        add_m1(n, mean, v);
    if (transform) { /* estimator.h:217-225, n is the already incremented count */
        const float fd = s - *fmean;
        const float fdN = fd / (float)n;
        *fmean += fdN;
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
                    transform, max_moment);
    } else {
        oracle_tile_pixel_f3 *p = (oracle_tile_pixel_f3 *)px;
        p->n++;
        for (int c = 0; c < 3; c++)
            add_channel(p->n, &p->mean[c], &p->m2[c], &p->m3[c], &p->film_mean[c], &p->film_m2[c],
                        sample[c], transform, max_moment);
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
            o[c] += splat_scale * splat_rgb[c];
            o[c] *= scale;
        }
    }
}

/* ---------------------------- filter spec v1 ------------------------------------------- */

static float tq_override[ORACLE_TQ_N_TABLES][ORACLE_TQ_N_DOF];
static int tq_overridden[ORACLE_TQ_N_TABLES];

void oracle_set_t_quantiles(int alpha_index, const float *q, int n_dof) {
    if (!q) {  /* back to the built-in table */
        tq_overridden[alpha_index] = 0;
        return;
    }
    for (int i = 0; i < ORACLE_TQ_N_DOF; i++) tq_override[alpha_index][i] = q[i < n_dof ? i : n_dof - 1];
    tq_overridden[alpha_index] = 1;
}

float oracle_t_quantile(int alpha_index, int dof) {
    if (dof < 1) return INFINITY;
    if (dof > ORACLE_TQ_N_DOF) dof = ORACLE_TQ_N_DOF;
    return tq_overridden[alpha_index] ? tq_override[alpha_index][dof - 1] : oracle_tq_tables[alpha_index][dof - 1];
}

void oracle_prepass(int width, int height, int channels, int alpha_index,
                    const int32_t *n, const float *mean, const float *m2, const float *m3,
                    float *mean_corr, float *discriminator) {
    const size_t npx = (size_t)width * height;
    for (size_t i = 0; i < npx; i++) {
        const int32_t ni = n[i];
        const float nf = (float)ni;
        const float t = oracle_t_quantile(alpha_index, ni - 1);
        for (int c = 0; c < channels; c++) {
            const size_t e = i * channels + c;
            const float mu = mean[e], s2sum = m2[e];
            if (ni >= 2 && s2sum > 0.f) {
                const float var = s2sum / (nf - 1.f);  /* unbiased sample variance s^2 */
                const float mu3 = m3[e] / nf;          /* third central sample moment */
                mean_corr[e] = mu + mu3 / (6.f * var * nf); /* Johnson (1978) */
                discriminator[e] = (t * t) * (var / nf);    /* squared CI half-width */
            } else {
                mean_corr[e] = mu;
                discriminator[e] = ni >= 2 ? 0.f : INFINITY;
            }
        }
    }
}

void oracle_filter(int width, int height, int channels, float ds, int radius,
                   const float *mean_corr, const float *disc, const float *colour,
                   int n_g, const float *const *g_buffers, const int *g_channels, const float *g_dr,
                   float *out, int rx0, int ry0, int rx1, int ry1, int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int y = ry0; y < ry1; y++) {
        for (int x = rx0; x < rx1; x++) {
            const size_t p = (size_t)y * width + x;
            float sum_w = 0.f, acc[3] = {0.f, 0.f, 0.f};
            /* a pixel whose corrected mean is not finite takes no part: it filters nothing (its output
             * is its own colour) and joins no other window.  (NaN does so by itself; +-inf would pass
             * `inf <= inf` against a pixel with fewer than two samples.) */
            int p_valid = 1;
            for (int c = 0; c < channels; c++) p_valid &= isfinite(mean_corr[p * channels + c]) != 0;
            for (int dy = -radius; p_valid && dy <= radius; dy++) {
                const int qy = y + dy;
                if (qy < 0 || qy >= height) continue;
                for (int dx = -radius; dx <= radius; dx++) {
                    const int qx = x + dx;
                    if (qx < 0 || qx >= width) continue;
                    const size_t q = (size_t)qy * width + qx;
                    /* membership: every channel must pass fma(d,d,-D_q) <= D_p */
                    int member = 1;
                    for (int c = 0; c < channels; c++) member &= isfinite(mean_corr[q * channels + c]) != 0;
                    for (int c = 0; c < channels; c++) {
                        const float d = mean_corr[p * channels + c] - mean_corr[q * channels + c];
                        const float lhs = fmaf(d, d, -disc[q * channels + c]);
                        member &= (lhs <= disc[p * channels + c]);
                    }
                    if (!member) continue;
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
