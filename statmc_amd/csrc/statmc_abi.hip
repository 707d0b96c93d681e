// statmc_abi.hip -- the extern "C" surface declared in include/statmc.h.
// Validation + argument marshalling only; kernels live in statmc_pointwise.hip / statmc_filter.hip.

#include <cmath>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "statmc_device.h"

#include <atomic>

#include "t_quantiles.h"
#include "../../include/statmc_pinned_spec.h"

namespace {

thread_local char g_err[512] = "";
thread_local const char *g_variant = "none";
thread_local int g_last_parts = 0, g_last_parts_hi = 0, g_last_tail_rows = 0;
thread_local const int *g_last_redo = nullptr;   // the item flags of the calling thread's last Welch launch (device memory)
thread_local int g_last_redo_n = 0;
std::mutex g_mu;

// Everything the library remembers is kept per device (one Estimator per device in a process that drives
// several GPUs; cf. one cv::cuda::Stream per Estimator, src/statistics/estimator.h:326): readiness (the
// quantile tables are a per-device __device__ symbol), significance level, filter spec, CU count.
struct DeviceState {
    bool ready = false;
    int alpha_index = STATMC_PINNED_SIGNIFICANCE;   // include/statmc_pinned_spec.h (tools/pin_from_dumps.sh)
    statmc_filter_spec spec = STATMC_PINNED_SPEC;
    int cus = 0;
    int split = 0;   // window-sweep parts per tile (statmc_set_filter_split): 0 = automatic
    // A/B and test switches (include/statmc_debug.h) -- per device like everything else, so that a test that pins one
    // device's dispatch does not reach a second Estimator on another device
    int force_variant = 0;                                    // statmc_debug_force_filter_variant
    int acc_resident_blocks = 0, acc_umul = 1, acc_dma = 1;   // film-major accumulation
    int acc_occ = 0;                                          // experiment builds: 3 = the accumulate kernel compiled for three waves per SIMD
    int acc_grid_mode = -1, acc_dma_first = 0;                // launch shape: -1 automatic (by batch length), 0 capped grid, 1 one pass per workgroup; A/B: ring rows requested before the state
    int tiles_umul = 2, tiles_order = 2, tiles_wg_per_cu = 0; // tile-fed accumulation (deeper prefetch of the mean-only types; a workgroup = four consecutive tiles of one type)
};
std::unordered_map<int, DeviceState> g_dev;  // guarded by g_mu

// snapshot of the current device's state; ready == false when statmc_setup has not run for it
DeviceState current_state(int *dev_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DeviceState();
    if (dev_out) *dev_out = dev;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dev.find(dev);
    return it == g_dev.end() ? DeviceState() : it->second;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace
namespace statmc {
int abi_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace statmc
namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(STATMC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define NEED_READY()                                                                                      \
    const DeviceState dstate = current_state();                                                           \
    do {                                                                                                  \
        if (!dstate.ready)                                                                                \
            return fail(STATMC_ERR_NO_DEVICE, "statmc_setup() has not been called for the current device"); \
    } while (0)

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

// spatial-exponent tables of the LDS filter, cached per (device, radius, ds)
struct TabKey {
    int dev, r;
    float ds;
    bool operator==(const TabKey &o) const { return dev == o.dev && r == o.r && memcmp(&ds, &o.ds, 4) == 0; }
};
struct TabHash {
    size_t operator()(const TabKey &k) const {
        unsigned u;
        memcpy(&u, &k.ds, 4);
        return (size_t)u * 1315423911u ^ (size_t)k.r * 2654435761u ^ (size_t)k.dev;
    }
};
std::unordered_map<TabKey, float *, TabHash> g_tabs;

// sym_rt: the table of the pair-symmetric kernel's runtime-radius build (cached under the key (dev, -radius, ds))
int spatial_table(int radius, float ds, const float **out, bool sym_rt = false) {
    *out = nullptr;
    const size_t n = sym_rt ? statmc::sym_rt_table_floats(radius) : statmc::spatial_table_floats(radius);
    if (n == 0) return STATMC_OK;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    const TabKey key{dev, sym_rt ? -radius : radius, ds};
    auto it = g_tabs.find(key);
    if (it == g_tabs.end()) {
        std::vector<float> host(n);
        if (sym_rt) statmc::fill_sym_rt_table(host.data(), radius, ds);
        else statmc::fill_spatial_table(host.data(), radius, ds);
        float *d = nullptr;
        HIP_TRY(hipMalloc(&d, n * sizeof(float)));
        // synchronous copy: the table must be resident before any stream uses it
        HIP_TRY(hipMemcpy(d, host.data(), n * sizeof(float), hipMemcpyHostToDevice));
        it = g_tabs.emplace(key, d).first;
    }
    *out = it->second;
    return STATMC_OK;
}

// workspace for the window filter's per-part partial sums, grown on demand, one per (device,
// stream): filter calls on different streams may be in flight together and must not share it
struct Workspace {
    float *ptr = nullptr;
    size_t bytes = 0;
};
struct WsKey {
    int dev;
    void *stream;
    bool operator==(const WsKey &o) const { return dev == o.dev && stream == o.stream; }
};
struct WsHash {
    size_t operator()(const WsKey &k) const { return std::hash<void *>()(k.stream) ^ ((size_t)k.dev * 0x9e3779b97f4a7c15ull); }
};
std::unordered_map<WsKey, Workspace, WsHash> g_ws;
struct DenseArena;
std::vector<std::shared_ptr<DenseArena>> detach_dense_arenas(void *stream);   // packed twins of pitched images (below); caller holds g_mu
void free_detached_arenas(std::vector<std::shared_ptr<DenseArena>> &arenas, void *stream);   // caller does not

int partial_workspace(size_t bytes, void *stream, float **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    Workspace &w = g_ws[WsKey{dev, stream}];
    if (w.bytes < bytes) {
        if (w.ptr) {
            HIP_TRY(hipStreamSynchronize(S(stream)));  // earlier launches on this stream may still read the old block
            HIP_TRY(statmc::workspace_free(w.ptr));
        }
        w.ptr = nullptr;
        w.bytes = 0;
        HIP_TRY(statmc::workspace_alloc(reinterpret_cast<void **>(&w.ptr), bytes));
        w.bytes = bytes;
    }
    *out = w.ptr;
    return STATMC_OK;
}

// kernel-side view of the current device's filter spec
void apply_spec(const DeviceState &d, statmc::FilterArgs &k, const statmc_filter_args *a) {
    k.gate = d.spec.gate;
    k.channel_rule = d.spec.channel_rule;
    k.dof = d.spec.dof;
    k.border = d.spec.border;
    k.force_variant = d.force_variant;
    k.force_parts = d.split;
    k.n = nullptr;
    k.tq = nullptr;
    k.tq2 = nullptr;
    if (d.spec.dof == STATMC_DOF_WELCH) {
        const int table = d.alpha_index + STATMC_TQ_N_ALPHAS * (d.spec.sides ? 1 : 0);
        k.tq = statmc::t_table_device_ptr(table);
        k.tq2 = statmc::t_table_sq_device_ptr(table);   // indexed by dof = 0 .. 4096 (entry 0 = entry 1)
    }
    (void)a;
}
// pair-symmetric kernel: tile range, parts and the patch workspace of this launch
// Window-sweep parts per tile are chosen for the whole local image, whatever output region a call asks for: the parts
// decide how a pixel's sums are grouped, so a pixel filtered as part of a band of rows (Estimator's Upload / Denoise /
// Download pipeline) gets the same bits as in a whole-image call.
int parts_for_whole_image(const statmc::FilterArgs &k, int cus, bool sym) {
    statmc::FilterArgs w = k;
    w.rx0 = 0;
    w.ry0 = 0;
    w.rx1 = k.width;
    w.ry1 = k.height;
    if (!sym) return statmc::lds_filter_parts(w, cus);
    statmc::sym_geometry(w);
    return statmc::sym_filter_parts(w, cus);
}

// pair: filter<float> on the pair-symmetric kernel -- the workspace also holds the three RGB-shaped images every
// launch packs its two buffers into (behind the patches, 16-byte aligned).
int prepare_sym(const DeviceState &d, statmc::FilterArgs &k, const statmc_filter_args *a, bool pair = false) {
    k.sym.fx0 = a->film_x0;
    k.sym.fy0 = a->film_y0;
    k.sym.tab_rt = nullptr;
    if (k.radius != 20 || k.dof == STATMC_DOF_WELCH) {   // (the Welch modes run the runtime-radius build at r = 20 too)
        if (int rc = spatial_table(k.radius, k.ds, &k.sym.tab_rt, true)) return rc;
    }
    // parts -- and the tail split, if the whole image's tile count leaves its last round mostly empty -- are chosen for
    // the WHOLE local image whatever region this call filters; so is the workspace: the bands of the Upload / Denoise /
    // Download pipeline then share one allocation (growing it mid-pipeline means a stream synchronisation and a hipFree
    // between two bands)
    statmc::FilterArgs whole = k;
    whole.rx0 = 0; whole.ry0 = 0; whole.rx1 = k.width; whole.ry1 = k.height;
    statmc::sym_geometry(whole);
    statmc::sym_choose_split(whole, d.cus);
    statmc::sym_apply_split(whole);
    k.n_parts = whole.n_parts;
    k.sym.parts_hi = whole.sym.parts_hi;
    k.sym.split_ty = whole.sym.split_ty;
    statmc::sym_geometry(k);
    statmc::sym_apply_split(k);
    float *ws = nullptr;
    const size_t patch_floats = (statmc::sym_patch_floats(whole, k.n_parts) + 3) & ~(size_t)3;
    const size_t image_floats = pair ? (size_t)9 * k.width * k.height : 0;
    const size_t extra_floats = k.border == STATMC_BORDER_CLAMP ? (size_t)4 * k.width * k.height : 0;
    const size_t redo_floats = k.dof == STATMC_DOF_WELCH ? ((size_t)statmc::sym_items(whole) + 3) & ~(size_t)3 : 0;
    if (int rc = partial_workspace((patch_floats + image_floats + extra_floats + redo_floats) * sizeof(float), a->stream, &ws)) return rc;
    k.sym.patch = reinterpret_cast<float4 *>(ws);
    k.sym.pair = pair ? 1 : 0;
    k.sym.pair_images = pair ? ws + patch_floats : nullptr;
    k.sym.border_extra = extra_floats ? reinterpret_cast<float4 *>(ws + patch_floats + image_floats) : nullptr;
    k.sym.redo = redo_floats ? reinterpret_cast<int *>(ws + patch_floats + image_floats + extra_floats) : nullptr;
    g_last_redo = k.sym.redo;
    g_last_redo_n = k.sym.redo ? (int)statmc::sym_items(k) : 0;
    g_last_parts_hi = whole.sym.parts_hi;
    g_last_tail_rows = whole.sym.parts_hi ? whole.sym.ty0 + whole.sym.nty - whole.sym.split_ty : 0;
    return STATMC_OK;
}
int prepass_table(const DeviceState &d) { return d.alpha_index + STATMC_TQ_N_ALPHAS * (d.spec.sides ? 1 : 0); }

int check_image(const statmc_image &im, int w, int h, int channels, const char *what, int idx) {
    if (!im.data) return fail(STATMC_ERR_INVALID, "%s[%d]: null device pointer", what, idx);
    if (im.cols != w || im.rows != h)
        return fail(STATMC_ERR_INVALID, "%s[%d]: %dx%d image, expected %dx%d", what, idx, im.cols, im.rows, w, h);
    if (im.step < (size_t)w * channels * 4)
        return fail(STATMC_ERR_INVALID, "%s[%d]: row pitch %zu is shorter than a row (%zu)", what, idx, im.step,
                    (size_t)w * channels * 4);
    if (im.step != (size_t)w * channels * 4)   // filter<T>, pre-pass, window filter and mean-vars take pitched images (packed twins)
        return fail(STATMC_ERR_UNSUPPORTED, "%s[%d]: row pitch %zu, this entry point needs packed rows (%zu)", what, idx,
                    im.step, (size_t)w * channels * 4);
    return STATMC_OK;
}


// Channels per pixel of a block + halo image: 15 (mean-corr, discriminator, colour, two RGB G-buffers), 16 (+ the sample
// count, as its bits: Welch degrees of freedom read it per pair), 17 (+ two 1-channel G-buffers: depth, material id --
// statpath.cpp:828-835) or 18 (both: channels 15, 16 the 1-channel G-buffers, channel 17 the count); 0 = none of them.
int packed_channels(const statmc_image &im) {
    if (im.cols <= 0) return 0;
    for (int ch = 15; ch <= 18; ch++)
        if (im.step == (size_t)im.cols * ch * 4) return ch;
    return 0;
}
inline bool packed_has_counts(int ch) { return ch == 16 || ch == 18; }
inline bool packed_has_scalars(int ch) { return ch == 17 || ch == 18; }
// The G-buffers of a call sorted into the slots of the packed image: up to two RGB images (argument order), then up to
// two 1-channel images (argument order).  15 channels: exactly two RGB images, as the reference's shipped configurations have.
struct PackedSlots {
    const float *rgb[2] = {nullptr, nullptr}, *sc[2] = {nullptr, nullptr};
};
int packed_slots(const statmc_filter_args *a, int ch, int W, int H, PackedSlots &out) {
    if (ch == 15 || ch == 16) {
        if (a->n_g_buffers != 2 || !a->g_buffers || (a->g_channel_counts && (a->g_channel_counts[0] != 3 || a->g_channel_counts[1] != 3)))
            return fail(STATMC_ERR_INVALID, "a 15- or 16-channel block + halo image holds exactly two RGB G-buffers");
    } else if (!a->g_buffers || !a->g_channel_counts || a->n_g_buffers > 4) {
        return fail(STATMC_ERR_INVALID, "a 17- or 18-channel block + halo image holds up to two RGB and two 1-channel G-buffers (g_channel_counts needed)");
    }
    int n_rgb = 0, n_sc = 0;
    for (size_t g = 0; g < a->n_g_buffers; g++) {
        const int gc = a->g_channel_counts ? a->g_channel_counts[g] : 3;
        if ((gc != 1 && gc != 3) || (gc == 3 && n_rgb == 2) || (gc == 1 && (n_sc == 2 || !packed_has_scalars(ch))))
            return fail(STATMC_ERR_UNSUPPORTED, "g_buffers[%zu]: %d channels do not fit the %d-channel block + halo image", g, gc, ch);
        if (int rc = check_image(a->g_buffers[g], W, H, gc, "g_buffers", (int)g)) return rc;
        (gc == 3 ? out.rgb[n_rgb++] : out.sc[n_sc++]) = static_cast<const float *>(a->g_buffers[g].data);
    }
    return STATMC_OK;
}

#define CHECK_IMG(im, ch, what, idx)                                          \
    do {                                                                      \
        int rc_ = check_image((im), W, H, (ch), (what), (idx));               \
        if (rc_) return rc_;                                                  \
    } while (0)

int check_common(const statmc_filter_args *a, int channels) {
    if (!a) return fail(STATMC_ERR_INVALID, "null args");
    if (channels != 1 && channels != 3) return fail(STATMC_ERR_INVALID, "channels must be 1 or 3");
    if (a->width == 0 || a->height == 0) return fail(STATMC_ERR_INVALID, "empty image");
    if (a->n_buffers > STATMC_MAX_BUFFERS) return fail(STATMC_ERR_INVALID, "n_buffers > %d", STATMC_MAX_BUFFERS);
    if (a->n_g_buffers > STATMC_MAX_GBUFFERS)
        return fail(STATMC_ERR_UNSUPPORTED, "n_g_buffers > %d", STATMC_MAX_GBUFFERS);
    return STATMC_OK;
}

// one wave: `cycles` shader clocks against the constant-rate clock (statmc_clock_probe)
__global__ void clock_probe_kernel(long long *out, long long cycles) {
    if (threadIdx.x != 0) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long c = c0;
    while (c - c0 < (unsigned long long)cycles) c = clock64();   // ends: the shader clock advances
    const unsigned long long w1 = wall_clock64();
    out[0] = (long long)(c - c0);
    out[1] = (long long)(w1 - w0);
}

}  // namespace

extern "C" {

const char *statmc_last_error(void) { return g_err; }
const char *statmc_last_filter_variant(void) { return g_variant; }
int statmc_version(void) { return 100; }

int statmc_setup(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(STATMC_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return fail(STATMC_ERR_INVALID, "device %d out of range [0,%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(STATMC_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    // check, table upload and `ready` under one lock: two threads setting up the same device must not both upload (the
    // second upload would overwrite quantiles a concurrent statmc_set_t_quantiles has just loaded)
    static std::mutex setup_mu;
    std::lock_guard<std::mutex> setup_lk(setup_mu);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_dev[device].ready) return STATMC_OK;  // idempotent: settings and loaded tables of the device stay
    }
    HIP_TRY(statmc::upload_t_tables());
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 256;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceState &d = g_dev[device];
    d.ready = true;
    d.cus = cus;
    return STATMC_OK;
}

int statmc_device_cus(void) { return current_state().cus; }

int statmc_set_device(int device) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_dev.find(device);
        if (it == g_dev.end() || !it->second.ready)
            return fail(STATMC_ERR_NO_DEVICE, "statmc_setup(%d) has not been called", device);
    }
    HIP_TRY(hipSetDevice(device));
    return STATMC_OK;
}

int statmc_set_significance(int alpha_index) {
    if (alpha_index < 0 || alpha_index > 2) return fail(STATMC_ERR_INVALID, "alpha_index must be 0, 1 or 2");
    int dev = 0;
    NEED_READY();
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].alpha_index = alpha_index;
    return STATMC_OK;
}
int statmc_get_significance(void) { return current_state().alpha_index; }

int statmc_set_filter_spec(const statmc_filter_spec *spec) {
    if (!spec) return fail(STATMC_ERR_INVALID, "null spec");
    const int32_t v[6] = {spec->gate, spec->channel_rule, spec->sides, spec->dof, spec->border, spec->small_n};
    for (int i = 0; i < 6; i++)
        if (v[i] < 0 || v[i] > (i == 0 ? 2 : 1)) return fail(STATMC_ERR_INVALID, "filter spec field %d: %d is out of range", i, v[i]);
    int dev = 0;
    NEED_READY();
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].spec = *spec;
    return STATMC_OK;
}
int statmc_reset_filter_spec(void) {   // back to what a freshly set-up device has: the pinned spec and significance level
    int dev = 0;
    NEED_READY();
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].spec = statmc_filter_spec STATMC_PINNED_SPEC;
    g_dev[dev].alpha_index = STATMC_PINNED_SIGNIFICANCE;
    return STATMC_OK;
}
const char *statmc_pinned_from(void) { return STATMC_PINNED_FROM; }
int statmc_get_filter_spec(statmc_filter_spec *spec) {
    if (!spec) return fail(STATMC_ERR_INVALID, "null spec");
    *spec = current_state().spec;
    return STATMC_OK;
}

int statmc_set_t_quantiles(int table, const float *quantiles, int n_dof) {
    NEED_READY();
    if (table < 0 || table >= STATMC_TQ_N_TABLES) return fail(STATMC_ERR_INVALID, "table must be 0..%d", STATMC_TQ_N_TABLES - 1);
    if (!quantiles && n_dof == 0) {  // back to the built-in table
        HIP_TRY(statmc::upload_t_table(table, statmc_tq_tables[table]));
        return STATMC_OK;
    }
    if (!quantiles || n_dof < 1 || n_dof > 4096) return fail(STATMC_ERR_INVALID, "need 1..4096 quantiles");
    for (int i = 0; i < n_dof; i++)
        if (!(quantiles[i] >= 0.f) || !std::isfinite(quantiles[i])) return fail(STATMC_ERR_INVALID, "quantile %d is not a finite non-negative number", i + 1);
    std::vector<float> t(4096);
    for (int i = 0; i < 4096; i++) t[i] = quantiles[i < n_dof ? i : n_dof - 1];
    HIP_TRY(statmc::upload_t_table(table, t.data()));
    return STATMC_OK;
}

// significance level, filter spec and the six quantile tables of `src_device` -> `dst_device` (both set up): what a host
// that shards one Estimator's film over several devices calls before it filters blocks there (FilmShards).
int statmc_copy_device_settings(int src_device, int dst_device) {
    DeviceState src;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto a = g_dev.find(src_device), b = g_dev.find(dst_device);
        if (a == g_dev.end() || !a->second.ready || b == g_dev.end() || !b->second.ready)
            return fail(STATMC_ERR_NO_DEVICE, "statmc_setup() has not been called for device %d or %d", src_device, dst_device);
        src = a->second;
    }
    if (src_device == dst_device) return STATMC_OK;
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    std::vector<float> tables((size_t)STATMC_TQ_N_TABLES * STATMC_TQ_N_DOF);
    hipError_t e = hipSetDevice(src_device);
    if (e == hipSuccess) e = hipMemcpy(tables.data(), statmc::t_table_device_ptr(0), tables.size() * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipSetDevice(dst_device);
    for (int t = 0; t < STATMC_TQ_N_TABLES && e == hipSuccess; t++) e = statmc::upload_t_table(t, tables.data() + (size_t)t * STATMC_TQ_N_DOF);
    (void)hipSetDevice(cur);
    if (e != hipSuccess) return fail(STATMC_ERR_HIP, "copying the quantile tables: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dst_device].alpha_index = src.alpha_index;
    g_dev[dst_device].spec = src.spec;
    g_dev[dst_device].split = src.split;
    return STATMC_OK;
}

int statmc_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(STATMC_ERR_INVALID, "null dev_ptr");
    HIP_TRY(hipMalloc(dev_ptr, bytes));
    return STATMC_OK;
}
int statmc_free(void *dev_ptr) {
    if (dev_ptr) {      // a statmc_malloc_placed block goes back to its slab
        const int placed = statmc::placement_free(dev_ptr);
        if (placed > 0) return STATMC_OK;
        if (placed < 0) return statmc::abi_fail(STATMC_ERR_INVALID, "statmc_free: %p lies inside the placed allocator's range but is not the start of a live block", dev_ptr);
    }
    HIP_TRY(hipFree(dev_ptr));
    return STATMC_OK;
}
int statmc_malloc_host(void **host_ptr, size_t bytes) {
    if (!host_ptr) return fail(STATMC_ERR_INVALID, "null host_ptr");
    HIP_TRY(hipHostMalloc(host_ptr, bytes, hipHostMallocPortable));
    return STATMC_OK;
}
int statmc_free_host(void *host_ptr) {
    HIP_TRY(hipHostFree(host_ptr));
    return STATMC_OK;
}
int statmc_memset(void *dev_ptr, int value, size_t bytes, void *stream) {
    HIP_TRY(hipMemsetAsync(dev_ptr, value, bytes, S(stream)));
    return STATMC_OK;
}
int statmc_upload(void *dev_dst, const void *host_src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, S(stream)));
    return STATMC_OK;
}
int statmc_download(void *host_dst, const void *dev_src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, S(stream)));
    return STATMC_OK;
}
int statmc_stream_create(void **stream) {
    if (!stream) return fail(STATMC_ERR_INVALID, "null stream");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return STATMC_OK;
}
// priority_class: 0 normal, > 0 high, < 0 low.  The HIP runtime multiplexes the streams of a process over
// GPU_MAX_HW_QUEUES (default 4) hardware queues PER PRIORITY LEVEL; a stream beyond that number shares a hardware queue
// with another one, and a barrier packet of one tenant (hipStreamWaitEvent) then holds back the other tenant's packets.
// Streams that must never stall each other (the copy queues of the band pipeline against the kernel stream) are
// therefore created in different priority classes: different pools, no shared hardware queue.
int statmc_stream_create_with_priority(void **stream, int priority_class) {
    if (!stream) return fail(STATMC_ERR_INVALID, "null stream");
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));   // numerically: greatest <= least
    const int prio = priority_class > 0 ? greatest : priority_class < 0 ? least : (least + greatest) / 2;
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
    *stream = s;
    return STATMC_OK;
}
int statmc_stream_destroy(void *stream) {
    // the stream's filter workspace goes with it (a later stream may get the same handle)
    std::vector<std::shared_ptr<DenseArena>> arenas;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_ws.begin(); it != g_ws.end();) {
            if (it->first.stream == stream && stream != nullptr) {
                if (it->second.ptr) {
                    (void)hipStreamSynchronize(S(stream));
                    (void)statmc::workspace_free(it->second.ptr);
                }
                it = g_ws.erase(it);
            } else {
                ++it;
            }
        }
        arenas = detach_dense_arenas(stream);
    }
    free_detached_arenas(arenas, stream);
    HIP_TRY(hipStreamDestroy(S(stream)));
    return STATMC_OK;
}
int statmc_event_create(void **event) {
    if (!event) return fail(STATMC_ERR_INVALID, "null event");
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return STATMC_OK;
}
int statmc_event_destroy(void *event) {
    HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return STATMC_OK;
}
int statmc_event_record(void *event, void *stream) {
    if (!event) return fail(STATMC_ERR_INVALID, "null event");
    HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), S(stream)));
    return STATMC_OK;
}
int statmc_stream_wait_event(void *stream, void *event) {
    if (!event) return fail(STATMC_ERR_INVALID, "null event");
    HIP_TRY(hipStreamWaitEvent(S(stream), static_cast<hipEvent_t>(event), 0));
    return STATMC_OK;
}
int statmc_synchronize(void *stream) {
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return STATMC_OK;
}

static int prepass_impl(const statmc_filter_args *a, int channels) {
    NEED_READY();
    if (int rc = check_common(a, channels)) return rc;
    const int W = a->width, H = a->height;
    if (a->n_buffers && (!a->n || !a->mean || !a->m2 || !a->m3 || !a->mean_corr || !a->discriminator))
        return fail(STATMC_ERR_INVALID, "null buffer table");
    for (int b = 0; b < a->n_buffers; b++) {
        CHECK_IMG(a->n[b], 1, "n", b);
        CHECK_IMG(a->mean[b], channels, "mean", b);
        CHECK_IMG(a->m2[b], channels, "m2", b);
        CHECK_IMG(a->m3[b], channels, "m3", b);
        CHECK_IMG(a->mean_corr[b], channels, "mean_corr", b);
        CHECK_IMG(a->discriminator[b], channels, "discriminator", b);
        statmc::PrepassArgs k;
        k.n = static_cast<const int32_t *>(a->n[b].data);
        k.mean = static_cast<const float *>(a->mean[b].data);
        k.m2 = static_cast<const float *>(a->m2[b].data);
        k.m3 = static_cast<const float *>(a->m3[b].data);
        k.mean_corr = static_cast<float *>(a->mean_corr[b].data);
        k.disc = static_cast<float *>(a->discriminator[b].data);
        k.n_elems = (long long)W * H * channels;
        k.channels = channels;
        k.table = prepass_table(dstate);
        k.welch = dstate.spec.dof == STATMC_DOF_WELCH;
        k.small_n_exclude = dstate.spec.small_n == STATMC_SMALL_N_EXCLUDE;
        HIP_TRY(statmc::launch_prepass(k, S(a->stream)));
    }
    return STATMC_OK;
}

static int window_filter_impl(const statmc_filter_args *a, int channels) {
    NEED_READY();
    if (int rc = check_common(a, channels)) return rc;
    const int W = a->width, H = a->height;
    const bool packed_in = a->packed_inputs.data != nullptr;
    if (!packed_in) {
        if (a->n_buffers && (!a->mean_corr || !a->discriminator)) return fail(STATMC_ERR_INVALID, "null buffer table");
        if (a->n_g_buffers && (!a->g_buffers || !a->g_channel_counts || !a->g_dr_factors))
            return fail(STATMC_ERR_INVALID, "null G-buffer table");
    }

    statmc::FilterArgs k;
    memset(&k, 0, sizeof(k));
    k.width = W;
    k.height = H;
    const bool whole = a->roi_x0 == 0 && a->roi_y0 == 0 && a->roi_x1 == 0 && a->roi_y1 == 0;
    k.rx0 = whole ? 0 : a->roi_x0;
    k.ry0 = whole ? 0 : a->roi_y0;
    k.rx1 = whole ? W : a->roi_x1;
    k.ry1 = whole ? H : a->roi_y1;
    if (k.rx0 < 0 || k.ry0 < 0 || k.rx1 > W || k.ry1 > H || k.rx0 >= k.rx1 || k.ry0 >= k.ry1)
        return fail(STATMC_ERR_INVALID, "roi [%d,%d)x[%d,%d) outside the %dx%d image", k.rx0, k.rx1, k.ry0, k.ry1, W, H);
    k.radius = a->filter_radius;
    k.ds = a->filter_ds_factor;
    k.n_g = (int)a->n_g_buffers;
    apply_spec(dstate, k, a);
    if (k.dof == STATMC_DOF_WELCH) {
        if (!k.tq) return fail(STATMC_ERR_HIP, "quantile table of the device not found");
        if (packed_in ? !packed_has_counts(packed_channels(a->packed_inputs)) : !a->n)
            return fail(STATMC_ERR_INVALID, "Welch dof: the window filter reads the sample counts (args->n, or the last channel of a 16- or 18-channel block + halo image)");
    }
    const bool packed = a->packed_inputs.data != nullptr;
    if (packed) {
        // block + halo path: everything the window filter reads comes from one 15- or 17-channel image
        const int pch = packed_channels(a->packed_inputs);
        if (channels != 3 || a->n_buffers != 1 || !a->g_dr_factors || pch == 0)
            return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs: needs T = float3, n_buffers = 1 and a packed 15-, 16-, 17- or 18-channel image");
        CHECK_IMG(a->packed_inputs, pch, "packed_inputs", 0);
        if (!a->film_filtered) return fail(STATMC_ERR_INVALID, "null film_filtered table");
        CHECK_IMG(a->film_filtered[0], 3, "film_filtered", 0);
        if (pch == 15 || pch == 16) {
            if (a->n_g_buffers != 2) return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs (15 / 16 channels): two RGB G-buffers");
            for (int g = 0; g < 2; g++) {
                k.g[g].data = nullptr;
                k.g[g].channels = 3;
                k.g[g].dr = a->g_dr_factors[g];
            }
        } else {
            if (a->n_g_buffers > 4 || !a->g_channel_counts)
                return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs (17 / 18 channels): up to two RGB and two 1-channel G-buffers, g_channel_counts needed");
            for (int g = 0; g < k.n_g; g++) {
                k.g[g].data = nullptr;
                k.g[g].channels = a->g_channel_counts[g];
                k.g[g].dr = a->g_dr_factors[g];
            }
        }
        k.packed = static_cast<const float *>(a->packed_inputs.data);
        k.packed_ch = pch;
        k.out = static_cast<float *>(a->film_filtered[0].data);
        if (pch == 17 || pch == 18) {
            // eight feature planes: the pair-symmetric kernel only (the one-sided kernel has six feature slots); 18 channels: its
            // eight-plane Welch builds, and only they
            if ((pch == 18 && k.dof != STATMC_DOF_WELCH) || !statmc::sym_eligible(k, 3))
                return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs (17 / 18 channels): runs on the pair-symmetric kernel only -- radius 1..20, DR factors "
                                                    "finite and <= 0, at most two RGB and two 1-channel G-buffers; 17 channels under STATMC_DOF_PIXEL, "
                                                    "18 under STATMC_DOF_WELCH");
            if (int rc = spatial_table(k.radius, k.ds, &k.spatial_tab)) return rc;
            if (!statmc::sym_path_selected(k, 3)) return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs (17 / 18 channels): the forced kernel variant cannot read them");
            statmc::sym_feature_slots(k);
            if (int rc = prepare_sym(dstate, k, a)) return rc;
        } else if (pch == 16) {
            // + the sample counts: the Welch builds of the pair-symmetric kernel, and only they
            if (k.dof != STATMC_DOF_WELCH || !statmc::sym_eligible(k, 3))
                return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs (16 channels): for STATMC_DOF_WELCH on the pair-symmetric kernel -- radius 1..20, DR "
                                                    "factors finite and <= 0, two RGB G-buffers");
            if (int rc = spatial_table(k.radius, k.ds, &k.spatial_tab)) return rc;
            k.gscale0 = sqrtf(-k.g[0].dr * 1.44269504088896340736f);
            k.gscale1 = sqrtf(-k.g[1].dr * 1.44269504088896340736f);
            if (!statmc::sym_path_selected(k, 3)) return fail(STATMC_ERR_UNSUPPORTED, "packed_inputs (16 channels): the forced kernel variant cannot read them");
            if (int rc = prepare_sym(dstate, k, a)) return rc;
        } else {
            if (!statmc::fast_path_eligible(k, 3))
                return fail(STATMC_ERR_UNSUPPORTED,
                            "packed_inputs: radius must be 1..20, DR factors finite and <= 0, and the discriminator's degrees "
                            "of freedom per pixel (STATMC_DOF_PIXEL)");
            if (int rc = spatial_table(k.radius, k.ds, &k.spatial_tab)) return rc;
            k.gscale0 = sqrtf(-k.g[0].dr * 1.44269504088896340736f);
            k.gscale1 = sqrtf(-k.g[1].dr * 1.44269504088896340736f);
            // (k.packed is set before the kernel is chosen: sym_eligible looks at it; the one-sided kernel takes what the
            // pair-symmetric one does not -- a forced variant)
            if (statmc::sym_path_selected(k, 3)) {
                if (int rc = prepare_sym(dstate, k, a)) return rc;
            } else {
                k.n_parts = parts_for_whole_image(k, dstate.cus, false);
                if (k.n_parts > 1) {
                    if (int rc = partial_workspace((size_t)k.n_parts * W * H * 4 * sizeof(float), a->stream, &k.partial)) return rc;
                }
            }
        }
        const char *variant = "none";
        HIP_TRY(statmc::launch_lds_packed(k, S(a->stream), &variant));
        g_variant = variant;
        g_last_parts = k.n_parts;
        return STATMC_OK;
    }
    for (int g = 0; g < k.n_g; g++) {
        const int gc = a->g_channel_counts[g];
        if (gc != 1 && gc != 3) return fail(STATMC_ERR_UNSUPPORTED, "g_buffers[%d]: %d channels", g, gc);
        CHECK_IMG(a->g_buffers[g], gc, "g_buffers", g);
        k.g[g].data = static_cast<const float *>(a->g_buffers[g].data);
        k.g[g].channels = gc;
        k.g[g].dr = a->g_dr_factors[g];
    }
    const bool lds_ok = statmc::fast_path_eligible(k, channels);   // one-sided LDS kernel: at most six feature channels
    if (lds_ok || statmc::sym_eligible(k, channels)) {
        if (int rc = spatial_table(k.radius, k.ds, &k.spatial_tab)) return rc;
        if (lds_ok) {
            statmc::set_feature_layout(k);
            k.n_parts = parts_for_whole_image(k, dstate.cus, false);
        } else {
            // pair-symmetric kernel only: eight feature planes (their factors come from sym_feature_slots) or the Welch
            // build, whose six planes take the two-RGB-image factors like every six-plane build
            if (k.dof == STATMC_DOF_WELCH) statmc::set_feature_layout(k);
            k.n_parts = 1;
        }
    } else {
        k.n_parts = 1;
    }
    // (the one-sided kernel's view of the call, before the pair-symmetric preparation below changes parts and workspace: an odd number
    // of float buffers ends with three of them on that kernel -- see the loop)
    const statmc::FilterArgs k_lds = k;
    // the pair-symmetric kernel (r = 20) also takes G-buffer sets the one-sided kernel has no slots for: two RGB + two
    // 1-channel images (normal, albedo, depth, material id)
    const bool sym = statmc::sym_path_selected(k, channels);
    const bool fast = sym || statmc::lds_path_selected(k, channels);
    if (sym) {
        statmc::sym_feature_slots(k);
        if (int rc = prepare_sym(dstate, k, a, channels == 1)) return rc;
    } else if (fast) {
        const int per_px = channels == 3 ? 4 : 8;
        if (k.n_parts > 1) {
            if (int rc = partial_workspace((size_t)k.n_parts * W * H * per_px * sizeof(float), a->stream, &k.partial)) return rc;
        }
    }
    // float buffers share the range weight of a launch: two per launch on the pair-symmetric kernel, three on the
    // one-sided LDS kernel
    const int pair_group = (fast && channels == 1) ? (sym ? 2 : 3) : 1;
    // An ODD number of float buffers (ACRR's five: estimator.cpp:434-460) would end with a launch of the pair-symmetric kernel that
    // carries one buffer at the price of two (1.39 ms at 1080p).  Where the one-sided kernel can take the call as well -- it shares the
    // range weight over THREE buffers, 2.47 ms -- the last three go to it: 5 buffers 1.40 + 2.47 instead of 1.40 + 1.40 + 1.39 ms
    // (round 6).  Only with the whole window sweep in one part there (no partial-sum workspace to share with the patches) and no forced
    // variant; the two kernels agree to 5e-7, each within 1e-5 of the oracle.
    static const bool mix_allowed = [] { const char *e = getenv("STATMC_FLOAT_MIX"); return !(e && e[0] == '0'); }();   // (A/B: tools/experiments/time_float.py)
    const bool mix = mix_allowed && sym && channels == 1 && lds_ok && dstate.force_variant == 0 && (a->n_buffers & 1) && a->n_buffers >= 3 && k_lds.n_parts == 1 &&
                     statmc::lds_path_selected(k_lds, channels);
    statmc::FilterArgs k_tail = k_lds;
    statmc::FilterArgs &k_pairs = k;
    for (int b0 = 0, group = pair_group; b0 < a->n_buffers; b0 += group) {
        const char *variant = "none";
        const bool tail = mix && a->n_buffers - b0 == 3;
        if (tail) group = 3;
        statmc::FilterArgs &k = tail ? k_tail : k_pairs;
        if (group > 1) {
            k.f_active = a->n_buffers - b0 < group ? a->n_buffers - b0 : group;
            if (!a->film || !a->film_filtered) return fail(STATMC_ERR_INVALID, "null film table");
            for (int j = 0; j < 3; j++) {
                const int b = b0 + (j < k.f_active ? j : k.f_active - 1);
                CHECK_IMG(a->mean_corr[b], 1, "mean_corr", b);
                CHECK_IMG(a->discriminator[b], 1, "discriminator", b);
                CHECK_IMG(a->film[b], 1, "film", b);
                CHECK_IMG(a->film_filtered[b], 1, "film_filtered", b);
                k.f_mean_corr[j] = static_cast<const float *>(a->mean_corr[b].data);
                k.f_disc[j] = static_cast<const float *>(a->discriminator[b].data);
                k.f_colour[j] = static_cast<const float *>(a->film[b].data);
                k.f_out[j] = static_cast<float *>(a->film_filtered[b].data);
                k.f_n[j] = nullptr;
                if (k.dof == STATMC_DOF_WELCH) {   // every buffer has its own sample counts
                    CHECK_IMG(a->n[b], 1, "n", b);
                    k.f_n[j] = static_cast<const int32_t *>(a->n[b].data);
                }
                if (k.f_out[j] == k.f_colour[j]) return fail(STATMC_ERR_INVALID, "filter cannot run in place (buffer %d)", b);
            }
            HIP_TRY(statmc::launch_window_filter(k, channels, S(a->stream), &variant));
            // (a call that ran both kernels names both: "sym_r20_f+lds_r20_f")
            if (tail && b0 > 0) {
                static thread_local char both[96];
                snprintf(both, sizeof(both), "%s+%s", g_variant, variant);
                variant = both;
            }
            g_variant = variant;
            g_last_parts = k.n_parts;
            continue;
        }
        const int b = b0;
        // buffer 0 filters the "film" image into "film-f" when denoiseFilm is set
        // (estimator.cpp:143-146,168-172; argument positions 12 and 20 of filter<T>)
        const bool film = a->denoise_film && b == 0 && channels == 3;
        const statmc_image &colour = film ? a->film_buffer : a->film[b];
        const statmc_image &out = film ? a->film_filtered_buffer : a->film_filtered[b];
        if (!film && (!a->film || !a->film_filtered)) return fail(STATMC_ERR_INVALID, "null film table");
        CHECK_IMG(a->mean_corr[b], channels, "mean_corr", b);
        CHECK_IMG(a->discriminator[b], channels, "discriminator", b);
        CHECK_IMG(colour, channels, film ? "film_buffer" : "film", b);
        CHECK_IMG(out, channels, film ? "film_filtered_buffer" : "film_filtered", b);
        k.mean_corr = static_cast<const float *>(a->mean_corr[b].data);
        k.disc = static_cast<const float *>(a->discriminator[b].data);
        k.colour = static_cast<const float *>(colour.data);
        k.out = static_cast<float *>(out.data);
        if (k.dof == STATMC_DOF_WELCH) {
            CHECK_IMG(a->n[b], 1, "n", b);
            k.n = static_cast<const int32_t *>(a->n[b].data);
        }
        if (k.out == k.colour) return fail(STATMC_ERR_INVALID, "filter cannot run in place (buffer %d)", b);
        HIP_TRY(statmc::launch_window_filter(k, channels, S(a->stream), &variant));
        g_variant = variant;
        g_last_parts = k.n_parts;
    }
    return STATMC_OK;
}

int statmc_pack_filter_inputs(const statmc_filter_args *a, const statmc_image *packed, int dst_x0, int dst_y0) {
    NEED_READY();
    if (int rc = check_common(a, 3)) return rc;
    if (!packed || !packed->data) return fail(STATMC_ERR_INVALID, "null packed image");
    if (a->n_buffers < 1 || !a->mean_corr || !a->discriminator) return fail(STATMC_ERR_INVALID, "pack needs buffer 0");
    const int pch = packed_channels(*packed);
    if (pch == 0) return fail(STATMC_ERR_UNSUPPORTED, "packed image must have packed rows of 15, 16, 17 or 18 channels");
    const int W = a->width, H = a->height;
    const bool film = a->denoise_film != 0;
    if (!film && !a->film) return fail(STATMC_ERR_INVALID, "null film table");
    const statmc_image &colour = film ? a->film_buffer : a->film[0];
    if (packed_has_counts(pch)) {
        if (!a->n) return fail(STATMC_ERR_INVALID, "a 16- or 18-channel block + halo image carries the sample counts: null n table");
        CHECK_IMG(a->n[0], 1, "n", 0);
    }
    CHECK_IMG(a->mean_corr[0], 3, "mean_corr", 0);
    CHECK_IMG(a->discriminator[0], 3, "discriminator", 0);
    CHECK_IMG(colour, 3, "colour", 0);
    PackedSlots gs;
    if (int rc = packed_slots(a, pch, W, H, gs)) return rc;
    if (dst_x0 < 0 || dst_y0 < 0 || dst_x0 + W > packed->cols || dst_y0 + H > packed->rows)
        return fail(STATMC_ERR_INVALID, "block %dx%d at (%d,%d) does not fit the %dx%d packed image", W, H, dst_x0, dst_y0,
                    packed->cols, packed->rows);
    statmc::PackArgs k{static_cast<const float *>(a->mean_corr[0].data), static_cast<const float *>(a->discriminator[0].data),
                       static_cast<const float *>(colour.data), gs.rgb[0], gs.rgb[1], static_cast<float *>(packed->data),
                       W, H, packed->cols, dst_x0, dst_y0, gs.sc[0], gs.sc[1], pch,
                       packed_has_counts(pch) ? static_cast<const int32_t *>(a->n[0].data) : nullptr};
    HIP_TRY(statmc::launch_pack_inputs(k, S(a->stream)));
    return STATMC_OK;
}

int statmc_prepass_pack(const statmc_filter_args *a, const statmc_image *packed, int dst_x0, int dst_y0) {
    return statmc_prepass_pack_rows(a, packed, dst_x0, dst_y0, nullptr, 0);
}
// ranges: n_ranges (0 = the whole block, else 1 or 2) disjoint ascending row ranges {y0, y1} of the block: only those rows, in
// one launch (the multi-GPU step pre-passes and packs the two strips its neighbours need before anything else)
int statmc_prepass_pack_rows(const statmc_filter_args *a, const statmc_image *packed, int dst_x0, int dst_y0, const int32_t *ranges,
                             int n_ranges) {
    NEED_READY();
    if (int rc = check_common(a, 3)) return rc;
    if (!packed || !packed->data) return fail(STATMC_ERR_INVALID, "null packed image");
    if (a->n_buffers < 1 || !a->n || !a->mean || !a->m2 || !a->m3)
        return fail(STATMC_ERR_INVALID, "prepass_pack needs buffer 0 (n, mean, m2, m3)");
    const int pch = packed_channels(*packed);
    if (pch == 0) return fail(STATMC_ERR_UNSUPPORTED, "packed image must have packed rows of 15, 16, 17 or 18 channels");
    const int W = a->width, H = a->height;
    const bool film = a->denoise_film != 0;
    if (!film && !a->film) return fail(STATMC_ERR_INVALID, "null film table");
    const statmc_image &colour = film ? a->film_buffer : a->film[0];
    CHECK_IMG(a->n[0], 1, "n", 0);
    CHECK_IMG(a->mean[0], 3, "mean", 0);
    CHECK_IMG(a->m2[0], 3, "m2", 0);
    CHECK_IMG(a->m3[0], 3, "m3", 0);
    CHECK_IMG(colour, 3, "colour", 0);
    PackedSlots gs;
    if (int rc = packed_slots(a, pch, W, H, gs)) return rc;
    float *mc = nullptr, *dc = nullptr;
    if (a->mean_corr && a->discriminator && a->mean_corr[0].data && a->discriminator[0].data) {
        CHECK_IMG(a->mean_corr[0], 3, "mean_corr", 0);
        CHECK_IMG(a->discriminator[0], 3, "discriminator", 0);
        mc = static_cast<float *>(a->mean_corr[0].data);
        dc = static_cast<float *>(a->discriminator[0].data);
    }
    if (dst_x0 < 0 || dst_y0 < 0 || dst_x0 + W > packed->cols || dst_y0 + H > packed->rows)
        return fail(STATMC_ERR_INVALID, "block %dx%d at (%d,%d) does not fit the %dx%d packed image", W, H, dst_x0, dst_y0,
                    packed->cols, packed->rows);
    statmc::PrepassPackArgs k{static_cast<const int32_t *>(a->n[0].data), static_cast<const float *>(a->mean[0].data),
                              static_cast<const float *>(a->m2[0].data), static_cast<const float *>(a->m3[0].data),
                              static_cast<const float *>(colour.data), gs.rgb[0], gs.rgb[1], mc, dc, static_cast<float *>(packed->data),
                              W, H, packed->cols, dst_x0, dst_y0, prepass_table(dstate),
                              dstate.spec.dof == STATMC_DOF_WELCH, dstate.spec.small_n == STATMC_SMALL_N_EXCLUDE, H, 0,
                              gs.sc[0], gs.sc[1], pch};
    if (n_ranges < 0 || n_ranges > 2 || (n_ranges && !ranges)) return fail(STATMC_ERR_INVALID, "0, 1 or 2 row ranges");
    if (n_ranges) {
        const int a0 = ranges[0], a1 = ranges[1], b0 = n_ranges == 2 ? ranges[2] : a1, b1 = n_ranges == 2 ? ranges[3] : a1;
        if (a0 < 0 || a0 > a1 || a1 > b0 || b0 > b1 || b1 > H) return fail(STATMC_ERR_INVALID, "row ranges must be ascending and inside the %d-row block", H);
        if (a1 - a0 + b1 - b0 == 0) return STATMC_OK;
        // the launch walks (a1 - a0) + (b1 - b0) rows of images that start at row a0; its rows from a1 - a0 on sit b0 - a1 further down
        const long long px0 = (long long)a0 * W;
        k.n += px0;
        k.mean += 3 * px0; k.m2 += 3 * px0; k.m3 += 3 * px0; k.colour += 3 * px0;
        if (k.g0) k.g0 += 3 * px0;
        if (k.g1) k.g1 += 3 * px0;
        if (k.s0) k.s0 += px0;
        if (k.s1) k.s1 += px0;
        if (k.mean_corr) { k.mean_corr += 3 * px0; k.disc += 3 * px0; }
        k.dst_y0 = dst_y0 + a0;
        k.src_h = (a1 - a0) + (b1 - b0);
        k.split_row = a1 - a0;
        k.skip_rows = b0 - a1;
    }
    HIP_TRY(statmc::launch_prepass_pack(k, S(a->stream)));
    return STATMC_OK;
}

// peer access src -> dst, once per ordered pair (an error other than "already enabled" is reported)
static int enable_peer(int dst_device, int src_device) {
    if (dst_device == src_device) return STATMC_OK;
    static std::mutex mu;
    static std::vector<std::pair<int, int>> done;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &p : done)
        if (p.first == dst_device && p.second == src_device) return STATMC_OK;
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, dst_device, src_device));
    if (!can) return fail(STATMC_ERR_UNSUPPORTED, "device %d cannot access device %d", dst_device, src_device);
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    HIP_TRY(hipSetDevice(dst_device));
    hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);
    (void)hipSetDevice(cur);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(STATMC_ERR_HIP, "hipDeviceEnablePeerAccess: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    // ... and the blocks of src's placed allocator (hipMemCreate / hipMemMap memory, which the call above does not cover)
    if (hipError_t pe = statmc::placement_grant_peer(src_device, dst_device); pe != hipSuccess)
        return fail(STATMC_ERR_HIP, "hipMemSetAccess (placed blocks of device %d for device %d): %s", src_device, dst_device, hipGetErrorString(pe));
    done.emplace_back(dst_device, src_device);
    return STATMC_OK;
}

int statmc_copy_rect(const statmc_image *dst, int dst_device, int dst_x, int dst_y, const statmc_image *src, int src_device,
                     int src_x, int src_y, int width, int height, int elem_bytes, void *stream) {
    if (!dst || !src || !dst->data || !src->data) return fail(STATMC_ERR_INVALID, "null image");
    if (width <= 0 || height <= 0) return STATMC_OK;
    if (elem_bytes <= 0 || dst_x < 0 || dst_y < 0 || src_x < 0 || src_y < 0 || dst_x + width > dst->cols || dst_y + height > dst->rows ||
        src_x + width > src->cols || src_y + height > src->rows)
        return fail(STATMC_ERR_INVALID, "rectangle outside an image");
    // access in both directions: the copy runs on `stream`, which may belong to either of the two devices (a block's own
    // stream when its result is pasted into another device's image), and whichever device executes it touches the other's memory
    if (int rc = enable_peer(dst_device, src_device)) return rc;
    if (int rc = enable_peer(src_device, dst_device)) return rc;
    const char *s = static_cast<const char *>(src->data) + (size_t)src_y * src->step + (size_t)src_x * elem_bytes;
    char *d = static_cast<char *>(dst->data) + (size_t)dst_y * dst->step + (size_t)dst_x * elem_bytes;
    HIP_TRY(hipMemcpy2DAsync(d, dst->step, s, src->step, (size_t)width * elem_bytes, height, hipMemcpyDeviceToDevice, S(stream)));
    return STATMC_OK;
}

int statmc_halo_exchange(const statmc_block *blocks, int gx, int gy, int block_w, int block_h, int radius) {
    if (!blocks || gx < 1 || gy < 1 || block_w < 1 || block_h < 1 || radius < 0) return fail(STATMC_ERR_INVALID, "bad block grid");
    if ((gx > 1 && block_w < radius) || (gy > 1 && block_h < radius))
        return fail(STATMC_ERR_INVALID, "block %dx%d smaller than the radius %d: a halo comes from one ring of neighbours", block_w, block_h, radius);
    const int n = gx * gy, r = radius;
    if (n == 1 || r == 0) return STATMC_OK;
    auto halo = [&](int bx, int by, int &pl, int &pr, int &pt, int &pb) {
        pl = bx > 0 ? r : 0; pr = bx + 1 < gx ? r : 0; pt = by > 0 ? r : 0; pb = by + 1 < gy ? r : 0;
    };
    const int pch = packed_channels(blocks[0].packed), px_bytes = pch * 4;
    for (int b = 0; b < n; b++) {
        int pl, pr, pt, pb;
        halo(b % gx, b / gx, pl, pr, pt, pb);
        const statmc_image &im = blocks[b].packed;
        if (!im.data || pch == 0 || im.cols != block_w + pl + pr || im.rows != block_h + pt + pb || packed_channels(im) != pch)
            return fail(STATMC_ERR_INVALID, "block %d: packed image is not the %dx%d block + halo image of 15 or 17 channels (the same for every block)", b,
                        block_w + pl + pr, block_h + pt + pb);
    }
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    std::vector<hipEvent_t> packed(n), phase1(n);
    int rc = STATMC_OK;
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == STATMC_OK) rc = fail(STATMC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (int b = 0; b < n && rc == STATMC_OK; b++) {   // every block's pack is complete when its event fires
        hip(hipSetDevice(blocks[b].device), "hipSetDevice");
        hip(hipEventCreateWithFlags(&packed[b], hipEventDisableTiming), "hipEventCreate");
        hip(hipEventCreateWithFlags(&phase1[b], hipEventDisableTiming), "hipEventCreate");
        hip(hipEventRecord(packed[b], S(blocks[b].stream)), "hipEventRecord");
    }
    auto copy = [&](int dst, int dx, int dy, int src, int sx, int sy, int w, int h) {
        if (rc != STATMC_OK) return;
        rc = statmc_copy_rect(&blocks[dst].packed, blocks[dst].device, dx, dy, &blocks[src].packed, blocks[src].device, sx, sy, w, h, px_bytes,
                              blocks[dst].stream);
    };
    // phase 1: columns of the owned rows, from the left / right neighbour's interior
    for (int b = 0; b < n && rc == STATMC_OK; b++) {
        const int bx = b % gx, by = b / gx;
        int pl, pr, pt, pb;
        halo(bx, by, pl, pr, pt, pb);
        hip(hipSetDevice(blocks[b].device), "hipSetDevice");
        if (pl) {
            int nl, nr, nt, nb;
            halo(bx - 1, by, nl, nr, nt, nb);
            hip(hipStreamWaitEvent(S(blocks[b].stream), packed[b - 1], 0), "hipStreamWaitEvent");
            copy(b, 0, pt, b - 1, nl + block_w - r, nt, r, block_h);
        }
        if (pr) {
            int nl, nr, nt, nb;
            halo(bx + 1, by, nl, nr, nt, nb);
            hip(hipStreamWaitEvent(S(blocks[b].stream), packed[b + 1], 0), "hipStreamWaitEvent");
            copy(b, pl + block_w, pt, b + 1, nl, nt, r, block_h);
        }
        hip(hipEventRecord(phase1[b], S(blocks[b].stream)), "hipEventRecord");
    }
    // phase 2: rows over the full widened width (the vertical neighbour's columns were filled by ITS phase 1)
    for (int b = 0; b < n && rc == STATMC_OK; b++) {
        const int bx = b % gx, by = b / gx;
        int pl, pr, pt, pb;
        halo(bx, by, pl, pr, pt, pb);
        hip(hipSetDevice(blocks[b].device), "hipSetDevice");
        const int pw = block_w + pl + pr;
        if (pt) {
            int nl, nr, nt, nb;
            halo(bx, by - 1, nl, nr, nt, nb);
            hip(hipStreamWaitEvent(S(blocks[b].stream), phase1[b - gx], 0), "hipStreamWaitEvent");
            copy(b, 0, 0, b - gx, 0, nt + block_h - r, pw, r);
        }
        if (pb) {
            int nl, nr, nt, nb;
            halo(bx, by + 1, nl, nr, nt, nb);
            hip(hipStreamWaitEvent(S(blocks[b].stream), phase1[b + gx], 0), "hipStreamWaitEvent");
            copy(b, 0, pt + block_h, b + gx, 0, nt, pw, r);
        }
    }
    for (int b = 0; b < n; b++) {   // events may be destroyed while pending: HIP releases them when they complete
        if (packed[b]) (void)hipEventDestroy(packed[b]);
        if (phase1[b]) (void)hipEventDestroy(phase1[b]);
    }
    (void)hipSetDevice(cur);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------- pitched device images
// cv::cuda::GpuMat allocates its rows with a pitch (buffer.h:25 keeps one per Buffer), so a caller that binds the C ABI to
// images it allocated itself may hand over rows longer than width x channels x 4 bytes.  The kernels walk packed rows.
// A call that names such an image runs on packed twins: every pitched image is copied (device to device, on the call's
// stream) into a packed image of a per-stream arena, the call runs on those, and the images it writes are copied back.
// Images of the adaptor and of statmc::Estimator are packed and never come this way.
namespace {
struct DenseTwin {
    statmc_image orig;
    void *dense;
    size_t row_bytes;
    bool copy_back;
};
struct PackedRowsCall {
    statmc_filter_args args;
    std::vector<statmc_image> n, mean, m2, m3, film, g, mc, disc, ff;
    std::vector<DenseTwin> twins;
};
// arena of the packed twins, one per (device, stream).  `in_use` is held for the WHOLE of a with_packed_rows call: a second
// host thread on the same stream must not grow (= free) the arena between this call's pointer fetch and its enqueues.
struct DenseArena : Workspace {
    std::mutex in_use;
};
// Held by shared_ptr: a call that has fetched its arena keeps it alive while statmc_stream_destroy drops the table's
// reference, so a waiter never locks a destroyed mutex.  Lock order: g_mu is never held while in_use is taken
// -- dense_arena_of releases it before the caller locks in_use, statmc_stream_destroy detaches the stream's arenas under
// g_mu and frees them after releasing it (the entry points wrapped by with_packed_rows take g_mu under in_use).
std::unordered_map<WsKey, std::shared_ptr<DenseArena>, WsHash> g_dense;

std::shared_ptr<DenseArena> dense_arena_of(void *stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    auto &slot = g_dense[WsKey{dev, stream}];
    if (!slot) slot = std::make_shared<DenseArena>();
    return slot;
}

int dense_arena(DenseArena &w, size_t bytes, void *stream, char **out) {   // caller holds w.in_use
    if (w.bytes < bytes) {
        if (w.ptr) {
            HIP_TRY(hipStreamSynchronize(S(stream)));
            HIP_TRY(statmc::workspace_free(w.ptr));
        }
        w.ptr = nullptr;
        w.bytes = 0;
        HIP_TRY(statmc::workspace_alloc(reinterpret_cast<void **>(&w.ptr), bytes));
        w.bytes = bytes;
    }
    *out = reinterpret_cast<char *>(w.ptr);
    return STATMC_OK;
}
// caller holds g_mu: takes the stream's arenas out of the table (nothing is locked or freed here)
std::vector<std::shared_ptr<DenseArena>> detach_dense_arenas(void *stream) {
    std::vector<std::shared_ptr<DenseArena>> out;
    for (auto it = g_dense.begin(); it != g_dense.end();) {
        if (it->first.stream == stream && stream != nullptr) {
            out.push_back(it->second);
            it = g_dense.erase(it);
        } else {
            ++it;
        }
    }
    return out;
}
// caller does NOT hold g_mu
void free_detached_arenas(std::vector<std::shared_ptr<DenseArena>> &arenas, void *stream) {
    for (auto &da : arenas) {
        std::lock_guard<std::mutex> busy(da->in_use);   // a call still enqueueing on the stream finishes first
        if (da->ptr) {
            (void)hipStreamSynchronize(S(stream));
            (void)statmc::workspace_free(da->ptr);
            da->ptr = nullptr;
            da->bytes = 0;
        }
    }
    arenas.clear();
}
inline bool pitched(const statmc_image &im, int channels) {
    return im.data && im.cols > 0 && im.rows > 0 && im.step > (size_t)im.cols * channels * 4;
}
inline size_t twin_bytes(const statmc_image &im, int channels) {
    return ((size_t)im.cols * channels * 4 * im.rows + 255) & ~(size_t)255;
}

// Runs fn on `a` with every pitched image replaced by a packed twin.  writes_stats: mean_corr / discriminator are outputs
// of the call (else inputs only); writes_filtered: film_filtered / film_filtered_buffer are outputs.
template <class Fn>
int with_packed_rows(const statmc_filter_args *a, int channels, bool writes_stats, bool writes_filtered, Fn fn) {
    if (!a || (channels != 1 && channels != 3) || a->n_buffers > STATMC_MAX_BUFFERS || a->n_g_buffers > STATMC_MAX_GBUFFERS)
        return fn(a);   // the entry point's own checks report it
    const int nb = a->n_buffers, ng = (int)a->n_g_buffers;
    PackedRowsCall c;
    c.args = *a;
    size_t total = 0;
    bool g_counts_ok = ng == 0 || (a->g_buffers && a->g_channel_counts);
    auto table = [&](const statmc_image *src, std::vector<statmc_image> &dst, const statmc_image *&slot, int ch) {
        if (!src || nb == 0) return;
        dst.assign(src, src + nb);
        slot = dst.data();
        for (auto &im : dst)
            if (pitched(im, ch)) total += twin_bytes(im, ch);
    };
    table(a->n, c.n, c.args.n, 1);
    table(a->mean, c.mean, c.args.mean, channels);
    table(a->m2, c.m2, c.args.m2, channels);
    table(a->m3, c.m3, c.args.m3, channels);
    table(a->film, c.film, c.args.film, channels);
    table(a->mean_corr, c.mc, c.args.mean_corr, channels);
    table(a->discriminator, c.disc, c.args.discriminator, channels);
    table(a->film_filtered, c.ff, c.args.film_filtered, channels);
    if (g_counts_ok && ng) {
        c.g.assign(a->g_buffers, a->g_buffers + ng);
        c.args.g_buffers = c.g.data();
        for (int g = 0; g < ng; g++)
            if (pitched(c.g[g], a->g_channel_counts[g] == 1 ? 1 : 3)) total += twin_bytes(c.g[g], a->g_channel_counts[g] == 1 ? 1 : 3);
    }
    if (pitched(a->film_buffer, 3)) total += twin_bytes(a->film_buffer, 3);
    if (pitched(a->film_filtered_buffer, 3)) total += twin_bytes(a->film_filtered_buffer, 3);
    if (total == 0) return fn(a);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));   // (NEED_READY of the entry point runs inside fn; an arena needs no library state)
    char *arena = nullptr;
    std::shared_ptr<DenseArena> da = dense_arena_of(a->stream);
    if (!da) return fail(STATMC_ERR_HIP, "no current device");
    std::lock_guard<std::mutex> busy(da->in_use);     // until every copy and kernel of this call has been enqueued
    if (int rc = dense_arena(*da, total, a->stream, &arena)) return rc;
    size_t off = 0;
    auto twin = [&](statmc_image &im, int ch, bool copy_back) -> int {
        if (!pitched(im, ch)) return STATMC_OK;
        const size_t row = (size_t)im.cols * ch * 4;
        void *dense = arena + off;
        off += twin_bytes(im, ch);
        // copied in even when the call writes it: a call with a region of interest leaves the rest of the image as it was
        HIP_TRY(hipMemcpy2DAsync(dense, row, im.data, im.step, row, im.rows, hipMemcpyDeviceToDevice, S(a->stream)));
        c.twins.push_back(DenseTwin{im, dense, row, copy_back});
        im.data = dense;
        im.step = row;
        return STATMC_OK;
    };
    for (auto &im : c.n) if (int rc = twin(im, 1, false)) return rc;
    for (auto &im : c.mean) if (int rc = twin(im, channels, false)) return rc;
    for (auto &im : c.m2) if (int rc = twin(im, channels, false)) return rc;
    for (auto &im : c.m3) if (int rc = twin(im, channels, false)) return rc;
    for (auto &im : c.film) if (int rc = twin(im, channels, false)) return rc;
    for (auto &im : c.mc) if (int rc = twin(im, channels, writes_stats)) return rc;
    for (auto &im : c.disc) if (int rc = twin(im, channels, writes_stats)) return rc;
    for (auto &im : c.ff) if (int rc = twin(im, channels, writes_filtered)) return rc;
    for (int g = 0; g < (int)c.g.size(); g++)
        if (int rc = twin(c.g[g], a->g_channel_counts[g] == 1 ? 1 : 3, false)) return rc;
    if (int rc = twin(c.args.film_buffer, 3, false)) return rc;
    if (int rc = twin(c.args.film_filtered_buffer, 3, writes_filtered)) return rc;
    if (int rc = fn(&c.args)) return rc;
    for (const DenseTwin &t : c.twins)
        if (t.copy_back)
            HIP_TRY(hipMemcpy2DAsync(t.orig.data, t.orig.step, t.dense, t.row_bytes, t.row_bytes, t.orig.rows,
                                     hipMemcpyDeviceToDevice, S(a->stream)));
    return STATMC_OK;
}
}  // namespace

extern "C" {
int statmc_prepass(const statmc_filter_args *a, int channels) {
    return with_packed_rows(a, channels, true, false, [&](const statmc_filter_args *p) { return prepass_impl(p, channels); });
}
int statmc_window_filter(const statmc_filter_args *a, int channels) {
    return with_packed_rows(a, channels, false, true, [&](const statmc_filter_args *p) { return window_filter_impl(p, channels); });
}
static int filter_both(const statmc_filter_args *a, int channels) {
    return with_packed_rows(a, channels, true, true, [&](const statmc_filter_args *p) {
        if (int rc = prepass_impl(p, channels)) return rc;
        return window_filter_impl(p, channels);
    });
}
int statmc_filter_f32(const statmc_filter_args *a) { return filter_both(a, 1); }
int statmc_filter_f32x3(const statmc_filter_args *a) { return filter_both(a, 3); }

static int mean_vars_impl(uint8_t n_buffers, uint16_t width, uint16_t height, int channels,
                          const statmc_image *n, const statmc_image *film_m2, const statmc_image *film_var,
                          int row_n_quirk, void *stream) {
    NEED_READY();
    if (channels != 1 && channels != 3) return fail(STATMC_ERR_INVALID, "channels must be 1 or 3");
    if (width == 0 || height == 0) return fail(STATMC_ERR_INVALID, "empty image");
    if (n_buffers && (!n || !film_m2 || !film_var)) return fail(STATMC_ERR_INVALID, "null buffer table");
    const int W = width, H = height;
    for (int b = 0; b < n_buffers; b++) {
        CHECK_IMG(n[b], 1, "n", b);
        CHECK_IMG(film_m2[b], channels, "film_m2", b);
        CHECK_IMG(film_var[b], channels, "film_var", b);
        statmc::MeanVarsArgs k{static_cast<const int32_t *>(n[b].data), static_cast<const float *>(film_m2[b].data),
                               static_cast<float *>(film_var[b].data), W, H, channels, row_n_quirk};
        HIP_TRY(statmc::launch_mean_vars(k, S(stream)));
    }
    return STATMC_OK;
}

int statmc_calculate_mean_vars(uint8_t n_buffers, uint16_t width, uint16_t height, int channels,
                               const statmc_image *n, const statmc_image *film_m2, const statmc_image *film_var,
                               int row_n_quirk, void *stream) {
    // pitched images: the three tables ride through the packed-twin helper in the slots of a filter call
    statmc_filter_args f;
    memset(&f, 0, sizeof(f));
    f.n_buffers = n_buffers;
    f.width = width;
    f.height = height;
    f.n = n;
    f.mean = film_m2;
    f.mean_corr = film_var;
    f.stream = stream;
    return with_packed_rows(&f, channels, true, false, [&](const statmc_filter_args *p) {
        return mean_vars_impl(n_buffers, width, height, channels, p->n, p->mean, p->mean_corr, row_n_quirk, stream);
    });
}

// validates one statmc_stat_type and translates it for the kernels (shared by both accumulate entries)
static int fill_stat_type(const DeviceState &ds, const statmc_stat_type &t, int i, uint16_t width, uint16_t height, bool batch,
                          statmc::AccumulateType &d) {
    if (t.channels != 1 && t.channels != 3) return fail(STATMC_ERR_INVALID, "types[%d]: channels must be 1 or 3", i);
    if (t.max_moment < 1 || t.max_moment > 3) return fail(STATMC_ERR_INVALID, "types[%d]: max_moment must be 1..3", i);
    if (batch && t.n_samples < 0) return fail(STATMC_ERR_INVALID, "types[%d]: negative n_samples", i);
    if (!t.n || !t.mean || ((!batch || t.n_samples) && !t.samples)) return fail(STATMC_ERR_INVALID, "types[%d]: null pointer", i);
    if (t.max_moment >= 2 && !t.m2) return fail(STATMC_ERR_INVALID, "types[%d]: null m2", i);
    if (t.max_moment >= 3 && !t.m3) return fail(STATMC_ERR_INVALID, "types[%d]: null m3", i);
    if (t.transform && (!t.film_mean || !t.film_m2))
        return fail(STATMC_ERR_INVALID, "types[%d]: transform types need film_mean and film_m2", i);
    d.samples = t.samples;
    d.n = t.n;
    d.mean = t.mean;
    d.m2 = t.m2;
    d.m3 = t.m3;
    d.film_mean = t.film_mean;
    d.film_m2 = t.film_m2;
    d.n_elems = (long long)width * height * t.channels;
    d.stride = d.n_elems;
    d.channels = t.channels;
    d.n_samples = t.n_samples;
    d.transform = t.transform ? 1 : 0;
    d.max_moment = t.max_moment;
    // optional epilogue: the pre-pass of the updated moments under the device's current spec and significance level
    // (statmc_prepass's own PrepassArgs, prepass_impl above)
    d.mean_corr = nullptr;
    d.disc = nullptr;
    d.pre_table = 0;
    d.pre_flags = 0;
    if (t.mean_corr || t.discriminator) {
        if (!t.mean_corr || !t.discriminator) return fail(STATMC_ERR_INVALID, "types[%d]: mean_corr and discriminator come together", i);
        if (t.max_moment < 3) return fail(STATMC_ERR_INVALID, "types[%d]: the pre-pass epilogue needs max_moment 3 (it reads m2 and m3)", i);
        d.mean_corr = t.mean_corr;
        d.disc = t.discriminator;
        d.pre_table = prepass_table(ds);
        d.pre_flags = (ds.spec.dof == STATMC_DOF_WELCH ? 1 : 0) | (ds.spec.small_n == STATMC_SMALL_N_EXCLUDE ? 2 : 0);
    }
    return STATMC_OK;
}

int statmc_accumulate(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types, void *stream) {
    return statmc_accumulate_rows(width, height, types, n_types, 0, height, stream);
}
// Rows [y0, y1) of the film only (per-pixel work: any split into row ranges leaves the same bits).  The multi-GPU step
// accumulates the r rows next to a neighbour first, hands them to the halo exchange and accumulates the rest while the
// exchange runs (statmc_amd/pipeline.py).
int statmc_accumulate_rows(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types, int y0, int y1, void *stream) {
    const int32_t range[2] = {y0, y1};
    return statmc_accumulate_row_ranges(width, height, types, n_types, range, 1, stream);
}
// Several disjoint row ranges in ONE launch (the two border strips of a block: a 20-row launch on its own is bound by the
// latency of its few waves, two of them together cost what one costs).
int statmc_accumulate_row_ranges(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types, const int32_t *ranges,
                                 int n_ranges, void *stream) {
    NEED_READY();
    if (width == 0 || height == 0) return fail(STATMC_ERR_INVALID, "empty image");
    if (n_types < 0 || n_ranges < 0 || (long long)n_types * n_ranges > statmc::kMaxStatTypes)
        return fail(STATMC_ERR_INVALID, "stat types x row ranges must be in [0,%d]", statmc::kMaxStatTypes);
    if (n_types == 0 || n_ranges == 0) return STATMC_OK;
    if (!types || !ranges) return fail(STATMC_ERR_INVALID, "null types or ranges");
    statmc::AccumulateArgs k;
    memset(&k, 0, sizeof(k));
    k.n_types = 0;
    for (int r = 0; r < n_ranges; r++) {
        const int y0 = ranges[2 * r], y1 = ranges[2 * r + 1];
        if (y0 < 0 || y1 > height || y0 > y1) return fail(STATMC_ERR_INVALID, "rows [%d,%d) outside the %d-row image", y0, y1, (int)height);
        for (int q = 0; q < r; q++)
            if (y0 < ranges[2 * q + 1] && ranges[2 * q] < y1) return fail(STATMC_ERR_INVALID, "row ranges %d and %d overlap", q, r);
        if (y0 == y1) continue;
        for (int i = 0; i < n_types; i++) {
            statmc::AccumulateType &d = k.t[k.n_types];
            if (int rc = fill_stat_type(dstate, types[i], i, width, height, true, d)) return rc;
            const long long px0 = (long long)y0 * width, e0 = px0 * d.channels;
            if (d.samples) d.samples += e0;
            d.n += px0;
            d.mean += e0;
            if (d.m2) d.m2 += e0;
            if (d.m3) d.m3 += e0;
            if (d.film_mean) d.film_mean += e0;
            if (d.film_m2) d.film_m2 += e0;
            if (d.mean_corr) { d.mean_corr += e0; d.disc += e0; }
            d.n_elems = (long long)(y1 - y0) * width * d.channels;   // d.stride stays the whole film's plane
            k.n_types++;
        }
    }
    if (k.n_types == 0) return STATMC_OK;
    k.resident_blocks = dstate.acc_resident_blocks;
    k.cus = dstate.cus;
    k.umul = dstate.acc_umul;
    k.dma = dstate.acc_dma;
    k.grid_mode = dstate.acc_grid_mode;
    k.dma_first = dstate.acc_dma_first;
    k.occ = dstate.acc_occ;
    // every type's samples in a STREAM block and its moments in a STATE block of the placed allocator: apart by construction
    k.apart = 1;
    for (int i = 0; i < k.n_types && k.apart; i++)
        k.apart = statmc::placement_role_of(k.t[i].samples) == STATMC_MEM_STREAM && statmc::placement_role_of(k.t[i].mean) == STATMC_MEM_STATE;
    HIP_TRY(statmc::launch_accumulate(k, S(stream)));
    return STATMC_OK;
}

int statmc_accumulate_tiles(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types,
                            const int32_t *tile_bounds, const int64_t *tile_offsets, const int32_t *tile_samples,
                            int n_tiles, void *stream) {
    NEED_READY();
    if (width == 0 || height == 0) return fail(STATMC_ERR_INVALID, "empty image");
    if (n_types < 0 || n_types > statmc::kMaxStatTypes)
        return fail(STATMC_ERR_INVALID, "n_types must be in [0,%d]", statmc::kMaxStatTypes);
    if (n_tiles < 0) return fail(STATMC_ERR_INVALID, "negative tile count");
    if (n_types == 0 || n_tiles == 0) return STATMC_OK;
    if (!types || !tile_bounds || !tile_offsets || !tile_samples) return fail(STATMC_ERR_INVALID, "null pointer");
    statmc::AccumulateTilesArgs k;
    memset(&k, 0, sizeof(k));
    k.n_types = n_types;
    for (int i = 0; i < n_types; i++)
        if (int rc = fill_stat_type(dstate, types[i], i, width, height, false, k.t[i])) return rc;
    k.tile_bounds = tile_bounds;
    k.tile_offsets = reinterpret_cast<const long long *>(tile_offsets);
    k.tile_samples = tile_samples;
    k.n_tiles = n_tiles;
    k.width = width;
    k.height = height;
    k.dma = dstate.acc_dma;
    k.umul = dstate.tiles_umul;
    k.order = dstate.tiles_order;
    k.wg_per_cu = dstate.tiles_wg_per_cu;
    k.dma_first = dstate.acc_dma_first;
    HIP_TRY(statmc::launch_accumulate_tiles(k, S(stream)));
    return STATMC_OK;
}

int statmc_merge_tiles(uint16_t width, uint16_t height, int channels, int transform, const void *tile_pixels,
                       const int32_t *tile_bounds, const int64_t *tile_offsets, int n_tiles, int max_tile_pixels,
                       int32_t *n, float *mean, float *m2, float *m3, float *film_mean, float *film_m2, void *stream) {
    NEED_READY();
    if (channels != 1 && channels != 3) return fail(STATMC_ERR_INVALID, "channels must be 1 or 3");
    if (n_tiles < 0 || max_tile_pixels < 0) return fail(STATMC_ERR_INVALID, "negative tile count");
    if (n_tiles == 0 || max_tile_pixels == 0) return STATMC_OK;
    if (n_tiles > 65535) return fail(STATMC_ERR_UNSUPPORTED, "more than 65535 tiles per call");
    if (!tile_pixels || !tile_bounds || !tile_offsets || !n || !mean || !m2 || !m3)
        return fail(STATMC_ERR_INVALID, "null pointer");
    if (transform && (!film_mean || !film_m2)) return fail(STATMC_ERR_INVALID, "transform merge needs film images");
    statmc::MergeTilesArgs k{tile_pixels, tile_bounds, reinterpret_cast<const long long *>(tile_offsets),
                             n, mean, m2, m3, film_mean, film_m2, width, height, channels, transform ? 1 : 0};
    HIP_TRY(statmc::launch_merge_tiles(k, n_tiles, max_tile_pixels, S(stream)));
    return STATMC_OK;
}

int statmc_film_update(const void *film_pixels, size_t n_pixels, float splat_scale, float scale, float *film_rgb,
                       void *stream) {
    NEED_READY();
    if (n_pixels == 0) return STATMC_OK;
    if (!film_pixels || !film_rgb) return fail(STATMC_ERR_INVALID, "null pointer");
    if (reinterpret_cast<uintptr_t>(film_pixels) & 15) return fail(STATMC_ERR_INVALID, "Film::Pixel array must be 16-byte aligned");
    HIP_TRY(statmc::launch_film_update(film_pixels, (long long)n_pixels, splat_scale, scale, film_rgb, S(stream)));
    return STATMC_OK;
}

int statmc_tile_moments(uint16_t width, uint16_t height, int channels, const float *values, int tile_size,
                        float *out, void *stream) {
    NEED_READY();
    if (channels < 1 || channels > 4) return fail(STATMC_ERR_INVALID, "channels must be 1..4");
    if (tile_size != 8 && tile_size != 16) return fail(STATMC_ERR_INVALID, "tile_size must be 8 or 16");
    if (!values || !out || width == 0 || height == 0) return fail(STATMC_ERR_INVALID, "null pointer / empty image");
    statmc::TileMomentsArgs k{values, out, width, height, channels, tile_size,
                              (width + tile_size - 1) / tile_size, (height + tile_size - 1) / tile_size};
    HIP_TRY(statmc::launch_tile_moments(k, S(stream)));
    return STATMC_OK;
}

int statmc_clock_probe(int64_t *out, int cycles, void *stream) {
    NEED_READY();
    if (!out) return fail(STATMC_ERR_INVALID, "null out");
    if (cycles < 1 || cycles > (1 << 24)) return fail(STATMC_ERR_INVALID, "cycles must be in 1 .. 2^24");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, S(stream), reinterpret_cast<long long *>(out), (long long)cycles);
    HIP_TRY(hipGetLastError());
    return STATMC_OK;
}

// ---- window-sweep split (include/statmc.h): per device, declared
int statmc_set_filter_split(int parts) {
    if (parts < 0 || parts > 64) return fail(STATMC_ERR_INVALID, "parts must be 0 (automatic) .. 64");
    int dev = 0;
    NEED_READY();
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].split = parts;
    return STATMC_OK;
}
int statmc_get_filter_split(void) { return current_state().split; }
int statmc_filter_split_auto(int width, int height, int radius) {
    NEED_READY();
    if (width < 1 || height < 1 || radius < 1 || radius > 20) return fail(STATMC_ERR_INVALID, "need a non-empty image and a radius in 1..20");
    statmc::FilterArgs k;
    memset(&k, 0, sizeof(k));
    k.width = width;
    k.height = height;
    k.rx1 = width;
    k.ry1 = height;
    k.radius = radius;
    statmc::sym_geometry(k);
    return statmc::sym_filter_parts(k, dstate.cus);
}

// ---- test / A-B switches (include/statmc_debug.h; not part of the reference surface).  Per device: they act on the calling
// thread's current device, which must have been set up.
#define STATMC_DEBUG_SET(stmt)                                       \
    do {                                                             \
        int dev = 0;                                                 \
        NEED_READY();                                                \
        HIP_TRY(hipGetDevice(&dev));                                 \
        std::lock_guard<std::mutex> lk(g_mu);                        \
        DeviceState &d = g_dev[dev];                                 \
        stmt;                                                        \
        return STATMC_OK;                                            \
    } while (0)
int statmc_debug_force_filter_variant(int v) {  // 0 auto, 1 generic, 2 runtime-radius one-sided LDS, 3 one-sided r = 20
    STATMC_DEBUG_SET(d.force_variant = v);
}
int statmc_debug_accumulate_resident_blocks(int n) {  // 0 by shape (default), n > 0: n resident workgroups, -1: never a resident grid
    STATMC_DEBUG_SET(d.acc_resident_blocks = n < 0 ? -1 : n);
}
int statmc_debug_last_accumulate_grid(void) { return (int)statmc::last_accumulate_grid(); }
int statmc_debug_accumulate_dma(int on) {   // 1 (default): RGB sample planes stream through LDS-DMA; 0: loads into registers; 3 .. 6: that ring depth (experiment builds)
    STATMC_DEBUG_SET(d.acc_dma = on < 0 ? 1 : on > 6 ? 6 : on == 2 ? 1 : on);
}
int statmc_debug_accumulate_occupancy(int waves_per_simd) {   // experiment builds (-DSTATMC_ACC_OCC_AB=1): 3 = the 168-VGPR build
    STATMC_DEBUG_SET(d.acc_occ = waves_per_simd == 3 ? 3 : 0);
}
int statmc_debug_accumulate_launch(int grid_mode, int dma_first) {   // A/B of the film-major launch shape (round 4)
    STATMC_DEBUG_SET(d.acc_grid_mode = grid_mode < 0 ? -1 : grid_mode == 1 ? 1 : 0; d.acc_dma_first = dma_first ? 1 : 0);
}
int statmc_debug_accumulate_umul(int umul) {   // film-major kernel: 2 = the mean-only feature types prefetch twice as deep
    STATMC_DEBUG_SET(d.acc_umul = umul == 2 ? 2 : 1);
}
int statmc_debug_accumulate_tiles_variant(int umul, int order, int wg_per_cu) {  // experiments (time_accumulate_tiles.py)
    STATMC_DEBUG_SET(d.tiles_umul = umul == 2 ? 2 : 1; d.tiles_order = order < 0 || order > 2 ? 0 : order; d.tiles_wg_per_cu = wg_per_cu < 0 ? 0 : wg_per_cu);
}
int statmc_debug_force_filter_parts(int k) { return statmc_set_filter_split(k < 0 ? 0 : k); }   // the older name of the pin
// non-zero: the library was built with an experiment switch of statmc_sym_experiments.h (never the product build)
int statmc_debug_diagnostic_build(void) { return statmc::sym_diagnostic_bits() | statmc::acc_diagnostic_bits(); }
int statmc_debug_last_filter_parts(void) { return g_last_parts; }
// the tail split of the calling thread's last pair-symmetric launch: parts of the last `*tail_rows` tile rows (0: uniform)
int statmc_debug_last_filter_tail(int *parts_hi, int *tail_rows) {
    if (parts_hi) *parts_hi = g_last_parts_hi;
    if (tail_rows) *tail_rows = g_last_tail_rows;
    return STATMC_OK;
}
// Welch degrees of freedom on the pair-symmetric kernel: how many work items of the calling thread's last launch asked
// for a quantile outside their band of the table and were computed again by the build that reads it in global memory
// (-1: the last launch was not a Welch one).  Waits for the device.
int statmc_debug_welch_far_items(void) {
    if (g_last_redo == nullptr || g_last_redo_n <= 0) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<int> f((size_t)g_last_redo_n);
    if (hipMemcpy(f.data(), g_last_redo, f.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int n = 0;
    for (int v : f) n += v != 0;
    return n;
}
// the partial-sum / patch workspace of the calling thread's current device and stream 0 (diagnostic builds read it back)
int statmc_debug_last_workspace(void **ptr, size_t *bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    *ptr = nullptr;
    *bytes = 0;
    for (auto &kv : g_ws)
        if (kv.first.dev == dev && kv.second.bytes > *bytes) {
            *ptr = kv.second.ptr;
            *bytes = kv.second.bytes;
        }
    return STATMC_OK;
}  // parts per tile of this thread's last window filter
}  // extern "C"
