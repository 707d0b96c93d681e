// statmc_device.h -- kernel-side argument blocks and launch prototypes shared by the
// .hip translation units of libstatmc_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/statmc.h"

namespace statmc {

// ---------------------------------------------------------------- pointwise kernels
struct PrepassArgs {
    const int32_t *n;
    const float *mean, *m2, *m3;
    float *mean_corr, *disc;
    long long n_elems;  // width*height*channels
    int channels;
    int table;            // alpha_index + 3 * sides
    int welch;            // discriminator = s^2 / n (no quantile)
    int small_n_exclude;  // n < 2 -> NaN mean / discriminator
};

struct MeanVarsArgs {
    const int32_t *n;
    const float *film_m2;
    float *film_var;
    int width, height, channels, row_n_quirk;
};

constexpr int kMaxStatTypes = 16;   // stat types of a call x row ranges of a call (statmc_accumulate_row_ranges)
struct AccumulateType {
    const float *samples;
    int32_t *n;
    float *mean, *m2, *m3, *film_mean, *film_m2;
    long long n_elems;  // elements this launch updates: width * rows * channels
    long long stride;   // floats between consecutive samples of an element: width * height * channels (> n_elems when the launch covers a range of rows)
    int channels, n_samples, transform, max_moment;
    // optional epilogue (max_moment 3): the pre-pass of the updated moments -- Johnson-corrected mean and discriminator, what
    // prepass_kernel computes from n / mean / m2 / m3 -- written while the state is still in registers (NULL: off)
    float *mean_corr, *disc;
    int pre_table;      // Student-t table of the device's significance level and sides (t_quantile)
    int pre_flags;      // 1: Welch degrees of freedom (t = 1: the pair looks its quantile up), 2: n < 2 excludes the pixel
};
constexpr int kMaxSlots = 64;
struct AccumulateArgs {
    AccumulateType t[kMaxStatTypes];
    int n_types;
    int resident_blocks;  // 0: by shape (launch_accumulate); > 0: that many workgroups walk all types; -1: never (A/B)
    int cus;              // compute units of the device (the resident grid's size where the shape calls for one)
    int umul;             // 2: the mean-only feature types prefetch twice as deep (statmc_debug_accumulate_umul; A/B)
    int dma;              // RGB sample planes arrive by LDS-DMA (default 1; 0: loads into registers, A/B)
    int grid_mode;        // -1: by batch length (default); 0: capped grid, slots per type in proportion to cost, grid-stride; 1: one pass per workgroup, types round-robin
    int dma_first;        // the first rows of the LDS-DMA ring are requested before the state loads
    int occ;              // experiment builds (STATMC_ACC_OCC_AB): 3 = the build for three waves per SIMD
    int apart;            // 1: samples and moments are known to lie in different interference classes (statmc_malloc_placed blocks)
    // large grid: workgroup b serves slot b % n_slots; slots are dealt to types in proportion to cost
    int n_slots;
    int type_slots[kMaxStatTypes];
    unsigned char slot_type[kMaxSlots], slot_rank[kMaxSlots];
};

// samples of every type arrive tile by tile: AccumulateType::samples is the type's arena, tile k's
// block starts at float offset tile_offsets[k] * channels and holds tile_samples[k] planes of
// tile_h x tile_w pixels; AccumulateType::n_samples and n_elems are not used
struct AccumulateTilesArgs {
    AccumulateType t[kMaxStatTypes];
    int n_types;
    const int32_t *tile_bounds;      // device, {x0, y0, x1, y1} per tile
    const long long *tile_offsets;   // device, in pixel-samples
    const int32_t *tile_samples;     // device
    int n_tiles, width, height, vec;
    int dma;                         // RGB sample planes arrive by LDS-DMA (default 1)
    int umul, order, wg_per_cu;      // experiment knobs (statmc_debug_accumulate_tiles_variant): prefetch depth x2, item order, grid size
    int dma_first;                   // A/B (statmc_debug_accumulate_launch): the first rows of the LDS-DMA ring requested before the state loads
};

struct MergeTilesArgs {
    const void *tile_pixels;
    const int32_t *tile_bounds;
    const long long *tile_offsets;
    int32_t *n;
    float *mean, *m2, *m3, *film_mean, *film_m2;
    int width, height, channels, transform;
};

struct TileMomentsArgs {
    const float *values;
    float *out;
    int width, height, channels, tile_size, tiles_x, tiles_y;
};

// ---------------------------------------------------------------- window filter
struct GBufferDesc {
    const float *data;
    int channels;
    float dr;  // -0.5/sd^2
};

// one buffer (one colour image, one stat set) per launch
struct FilterArgs {
    const float *mean_corr, *disc, *colour;
    float *out;
    // filter spec (statmc_filter_spec): every field but dof = Welch has an LDS kernel (statmc_filter.hip, statmc_filter_sym.hip)
    int gate, channel_rule, dof, border;
    // dispatch overrides of the device (statmc_abi.hip DeviceState): force_variant 0 auto, 1 generic, 2 lds_rt (one-sided,
    // runtime radius), 3 lds_r20 (one-sided, compile-time radius 20); force_parts 0 automatic, k >= 1: window-sweep parts
    // per tile (statmc_set_filter_split)
    int force_variant, force_parts;
    const int32_t *n;            // Welch mode: sample counts
    const float *tq;             // Welch mode: this device's quantile table (4096 entries)
    const float *tq2;            // ... and its squares: tq2[dof] = fl(t_dof * t_dof), dof = 0 .. 4096, entry 0 = entry 1 (pair-symmetric kernel)
    int width, height;           // local image
    int rx0, ry0, rx1, ry1;      // output ROI
    int rx_split, n_main_items;  // LDS kernel: regular tiles cover [rx0, rx_split), DUAL tiles [rx_split, rx1)
    int radius;
    float ds;                    // -0.5/sd_s^2
    int n_g;
    GBufferDesc g[STATMC_MAX_GBUFFERS];
    // fast path only (T = float3, two 3-channel G-buffers):
    const float *spatial_tab;    // [(2r+1)][2*RP+7] log2-domain spatial exponents, -inf outside r
    float gscale0, gscale1;      // sqrt(-dr_g * log2(e))
    // G-buffer sets other than "up to two RGB images" whose channels still fit the kernel's six feature
    // slots (e.g. normal + depth + material id): slot f reads data[pixel * stride + offset] * scale;
    // gscale0 = gscale1 = 1 then.  scale 0 = empty slot.
    int feat_generic;
    struct FeatSlot {
        const float *data;
        int stride, offset;
        float scale;
    } feat[6];
    int n_parts;                 // window rows are swept by n_parts workgroups per tile ...
    float *partial;              // ... which leave their sums here: [n_parts][height][width][4 (RGB) | 8 (float x3)]
    // fast path, filter<float>: up to three 1-channel buffers per launch (f_active of them real;
    // the rest repeat the last one and are not stored)
    const float *f_mean_corr[3], *f_disc[3], *f_colour[3];
    const int32_t *f_n[3];       // Welch degrees of freedom: the buffers' sample counts (pair-symmetric kernel: two buffers per launch)
    float *f_out[3];
    int f_active;
    const float *packed;         // optional [height][width][packed_ch] inputs: mc, disc, colour, g0, g1 (RGB each)[, s0, s1]
    int packed_ch;               // 15, 16 (+ the sample count's bits: the Welch builds of the pair-symmetric kernel) or 17 (+ two
                                 // 1-channel G-buffers: its eight-plane build)
    // pair-symmetric kernel (statmc_filter_sym.hip): tiles of 128 x 8 pixels on a grid fixed in film coordinates
    struct SymGeom {
        int tx0, ty0, ntx, nty;   // tile range of the launch (film tile indices)
        int fx0, fy0;             // film coordinates of local pixel (0, 0)
        long long item_stride4;   // float4 per work item (tile, part) in the patch workspace
        float4 *patch;            // [items][p-side 8 x 128 | q-side rows x 168] (sum w*colour rgb, sum w)
        int pair;                 // filter<float>: f_active (1 or 2) 1-channel buffers (f_mean_corr / f_disc / f_colour / f_out) per launch
        float *pair_images;       // ... staged from three [height][width][3] images this launch packs them into
        float4 *border_extra;     // border rule "clamp": per pixel, the sums over the taps beyond the image (border_virtual_kernel)
        int *redo;                // Welch: one flag per work item, set by the band build for the items the far build computes again
        // eight feature planes (NG = 8 build): up to two RGB and up to two 1-channel G-buffers of the argument list, sorted
        // into slots by sym_feature_slots(); scale = sqrt(-dr * log2 e), 0 = empty slot (never read)
        // Tail split (round 4): the tiles of the film's tile rows >= split_ty sweep with parts_hi workgroups each instead of
        // n_parts, so that the launch's last round of workgroups is full (1280 x 720: 900 tiles = 3.5 rounds of 256).  The
        // work items of the n_parts-tiles come first (n_lo_items of them: the launch's tile rows below split_ty), then
        // those of the parts_hi-tiles.  parts_hi = 0: every tile has n_parts.  Chosen for the WHOLE local image, like n_parts.
        int parts_hi, split_ty, n_lo_tiles, n_lo_items;
        int steps;                // window rows a tile sweeps: radius + 1
        const float *tab_rt;      // runtime-radius build (radius < 20): [radius + 1][47] spatial exponents, row = dy, -inf beyond the radius
        int g8;
        const float *rgb[2], *sc[2];
        float rgb_scale[2], sc_scale[2];
    } sym;
};

struct PackArgs {
    const float *mean_corr, *disc, *colour, *g0, *g1;  // [src_h][src_w][3]  (an absent G-buffer: nullptr, packed as zeros)
    float *packed;                                      // [dst_h][dst_w][ch]
    int src_w, src_h, dst_w, dst_x0, dst_y0;
    const float *s0, *s1;                               // ch = 17: the two 1-channel G-buffers [src_h][src_w] (nullptr: zeros)
    int ch;                                             // 15 | 16 | 17
    const int32_t *n;                                   // ch = 16: the sample counts (channel 15 holds their bits: Welch dof)
};
hipError_t launch_pack_inputs(const PackArgs &a, hipStream_t s);

// pre-pass and pack in one pass (RGB): statistics + colour + two G-buffers in, packed image out
// (mean_corr / disc also written to their own images when the pointers are set)
struct PrepassPackArgs {
    const int32_t *n;
    const float *mean, *m2, *m3, *colour, *g0, *g1;
    float *mean_corr, *disc;   // optional
    float *packed;             // [dst_h][dst_w][ch]
    int src_w, src_h, dst_w, dst_x0, dst_y0, table, welch, small_n_exclude;
    int split_row, skip_rows;   // rows >= split_row of the launch's src_h rows sit skip_rows further down in every image (two row ranges in one launch)
    const float *s0, *s1;       // ch = 17: the two 1-channel G-buffers (nullptr: zeros); g0 / g1 may be nullptr as well then
    int ch;                     // 15 | 16 (channel 15 = the bits of n: Welch degrees of freedom) | 17
};
hipError_t launch_prepass_pack(const PrepassPackArgs &a, hipStream_t s);

hipError_t upload_t_tables();
hipError_t upload_t_table(int table, const float *host_4096);
const float *t_table_device_ptr(int table);  // current device's copy of quantile table `table`
const float *t_table_sq_device_ptr(int table);  // ... of its squares (fl(t * t), what the Welch pair test multiplies with)
hipError_t launch_prepass(const PrepassArgs &a, hipStream_t s);
hipError_t launch_mean_vars(const MeanVarsArgs &a, hipStream_t s);
hipError_t launch_accumulate(const AccumulateArgs &a, hipStream_t s);
unsigned last_accumulate_grid();
hipError_t launch_accumulate_tiles(const AccumulateTilesArgs &a, hipStream_t s);
hipError_t launch_merge_tiles(const MergeTilesArgs &a, int n_tiles, int max_tile_pixels, hipStream_t s);
hipError_t launch_tile_moments(const TileMomentsArgs &a, hipStream_t s);
hipError_t launch_film_update(const void *pixels, long long n, float splat_scale, float scale, float *rgb, hipStream_t s);

// Returns the variant name through *variant.  channels = 1 or 3.
hipError_t launch_window_filter(const FilterArgs &a, int channels, hipStream_t s, const char **variant);
hipError_t launch_lds_packed(const FilterArgs &a, hipStream_t s, const char **variant);
// Size in floats of the spatial table the fast path wants for radius r (0 if r unsupported).
size_t spatial_table_floats(int radius);
size_t sym_rt_table_floats(int radius);                   // pair-symmetric kernel, radius < 20
void fill_sym_rt_table(float *host_tab, int radius, float ds);
void fill_spatial_table(float *host_tab, int radius, float ds);
bool fast_path_eligible(const FilterArgs &a, int channels);
void set_feature_layout(FilterArgs &a);   // gscale0/1, feat[] of an eligible G-buffer set
bool lds_path_selected(const FilterArgs &a, int channels);

// pair-symmetric kernel
bool sym_eligible(const FilterArgs &a, int channels);
void sym_feature_slots(FilterArgs &a);                       // fills a.sym.g8 / rgb / sc from a.g[] (eligible sets only)
bool sym_path_selected(const FilterArgs &a, int channels);   // eligible and not overridden
void sym_geometry(FilterArgs &a);                            // fills a.sym.tx0 .. nty from the ROI and film origin
int sym_tiles(const FilterArgs &a);
int sym_choose_parts(int tiles, int n_cus, int steps);
int sym_filter_parts(const FilterArgs &a, int n_cus);
void sym_choose_split(FilterArgs &whole, int n_cus);        // n_parts, sym.parts_hi, sym.split_ty for the whole local image (a.sym geometry filled)
void sym_apply_split(FilterArgs &a);                        // n_lo_tiles / n_lo_items of this launch's tile range
long long sym_items(const FilterArgs &a);
size_t sym_patch_floats(const FilterArgs &a, int n_parts);
hipError_t launch_sym(FilterArgs a, hipStream_t s);
int sym_diagnostic_bits();   // non-zero: built with a STATMC_SYM_* experiment switch (statmc_sym_experiments.h)
int acc_diagnostic_bits();   // non-zero: the accumulation was built with a timing-only switch (STATMC_ACC_SKIP_STORES: bit 7)
hipError_t launch_border_virtual(const FilterArgs &a, hipStream_t s);   // the clamped border's taps beyond the image (RGB)
int choose_parts(int tiles, int n_rows, int n_cus);
// statmc_placement.hip (device memory placed by HBM rank)
int abi_fail(int code, const char *fmt, ...);   // records the calling thread's statmc_last_error() text, returns `code` (statmc_abi.hip)
int placement_role_of(const void *ptr);         // STATMC_MEM_STATE / _STREAM when `ptr` lies in a block dealt with the wanted class, else -1
int placement_free(void *ptr);                  // 1: `ptr` was a statmc_malloc_placed block and is free now; 0: not the placed allocator's; -1: inside its ranges, not a live block's start
hipError_t workspace_alloc(void **p, size_t bytes);   // the library's own read-and-written workspaces: STATE role where the device's caller uses placed memory, hipMalloc otherwise
hipError_t workspace_free(void *p);
hipError_t placement_grant_peer(int owner_device, int peer_device);   // blocks of `owner`'s placed allocator become valid operands of copies device `peer` executes
// parts per tile the LDS kernel would use for this ROI on a device with n_cus compute units
int lds_filter_parts(const FilterArgs &a, int n_cus);

}  // namespace statmc
