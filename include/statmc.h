/* statmc.h -- C ABI of libstatmc_hip.so: the MI355X (gfx950) implementation of StatMC's
 * per-pixel statistics accumulation and statistics-gated cross-bilateral filter.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface it replaces
 * (paths relative to the StatMC repository).  Conventions:
 *   - plain C types only; device memory is passed as raw pointers (hipMalloc'ed by the caller,
 *     by statmc_malloc, or by any other HIP allocator in the same process, e.g. PyTorch-ROCm);
 *   - `stream` is a hipStream_t cast to void* (NULL = the default stream); every call is
 *     asynchronous on it, exactly like the reference enqueues on its one cv::cuda::Stream
 *     (src/statistics/estimator.h:326); statmc_synchronize() closes an iteration;
 *   - return value: 0 on success, a negative STATMC_ERR_* otherwise; statmc_last_error()
 *     returns a thread-local message.  (The reference has no error convention at these call
 *     sites -- OpenCV throws)
 *   - images are row-major with interleaved channels, exactly the cv::Mat / GpuMat layout of
 *     src/statistics/buffer.h:19-71: int32 x1 for "n", float32 x1 or x3 otherwise.  `step` is
 *     the row pitch in bytes.  The kernels walk tightly packed rows (step == cols * channels * 4), which is what
 *     the library allocates; filter<T>, pre-pass, window filter and mean-vars also take images with a longer pitch
 *     (a foreign GpuMat: they run on packed twins), the other entry points return STATMC_ERR_UNSUPPORTED for them.
 */
#ifndef STATMC_H
#define STATMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STATMC_OK 0
#define STATMC_ERR_INVALID (-1)     /* bad argument (null pointer, zero size, radius too large) */
#define STATMC_ERR_UNSUPPORTED (-2) /* valid in the reference, not supported by this build */
#define STATMC_ERR_HIP (-3)         /* a HIP runtime call failed; see statmc_last_error() */
#define STATMC_ERR_NO_DEVICE (-4)   /* no gfx950 device / setup not called */

#define STATMC_MAX_BUFFERS 16  /* nBuffers per filter call (reference passes a uchar) */
#define STATMC_MAX_GBUFFERS 8

const char *statmc_last_error(void);

/* Replaces cv::cuda::stat_denoiser::setup()  (src/statistics/estimator.h:280).
 * Selects the device, uploads the Student-t quantile tables. Idempotent per device.
 *
 * All library state is kept PER DEVICE (quantile tables, significance level, filter spec, kernel
 * attributes, workspaces): a process that drives several GPUs -- one Estimator per device, the
 * one-process-eight-devices host design -- calls statmc_setup(d) once for each and every setter
 * below acts on the calling thread's current device (statmc_set_device). */
int statmc_setup(int device);

/* Compute units of the current device (0 before statmc_setup): what the launches are fitted to -- the window-sweep split,
 * and the row bands of the Upload / Denoise / Download pipeline (include/statmc_bands.hpp). */
int statmc_device_cus(void);

/* Makes `device` current for the calling thread (HIP's current device is per thread): a thread
 * other than the one that ran statmc_setup -- e.g. a render worker whose Merge*Tiles call triggers a
 * flush -- calls this before using a device other than 0.  The device must have been set up. */
int statmc_set_device(int device);

/* The reference picks the significance level at compile time by pointing `t_quantiles` at one
 * of three tables (README.md:149,158): 0 -> 0.005 (default), 1 -> 0.002, 2 -> 0.05.
 * Acts on the current device. */
int statmc_set_significance(int alpha_index);
int statmc_get_significance(void);
/* Replaces the built-in table `table` of the current device with the caller's quantiles for
 * dof = 1..n_dof (n_dof <= 4096; larger dof reuse the last entry).  table = alpha_index (0..2) for the
 * two-sided tables t_{1-alpha/2}, 3 + alpha_index for the one-sided ones t_{1-alpha} (see
 * statmc_filter_spec.sides).  The built-in tables are this build's choice; the reference's own
 * `t_quantiles` arrays live in the un-vendored stat_denoiser.cu, and a user who has them can load them
 * here; quantiles == NULL with n_dof == 0 restores the built-in table.  Call after statmc_setup()
 * (a repeated statmc_setup of the same device keeps what was loaded). */
int statmc_set_t_quantiles(int table, const float *quantiles, int n_dof);

/* Copies significance level, filter spec, window-sweep split and the quantile tables (built-in or caller-supplied) of
 * `src_device` to `dst_device`; both must have been set up.  All of these are per-device state: a host that spreads one Estimator's
 * film over several devices (statmc::FilmShards) calls this so that every block is filtered under the same rules. */
int statmc_copy_device_settings(int src_device, int dst_device);

/* ---- filter spec: everything about cv::cuda::stat_denoiser::filter<T> that the reference tree does
 * not fix (its CUDA source is in the un-vendored submodule src/ext/opencv_contrib, .gitmodules:19-21;
 * only the call sites src/statistics/estimator.cpp:437-487 and the buffer meanings README.md:317-325
 * are in the tree).  Every open choice is a field, so that pinning this build to dumps of the CUDA
 * denoiser is a search over specs (tools/fit_spec.py), not a kernel rewrite.  All-zero = this build's
 * default ("spec v2").  `sides` and `small_n` only change the pre-pass.
 * What serves which spec (statmc_last_filter_variant() names the kernel of the calling thread's last filter call;
 * times: one 1080p RGB buffer, r = 20):
 *   every gate x channel rule x border, per-pixel dof, <= 2 RGB + <= 2 one-channel G-buffers   pair-symmetric LDS kernel   1.4 - 1.9 ms
 *   ... float buffers under the asymmetric / centre gate                                       one-sided LDS kernel        2.1 ms per 3 buffers
 *   ... float buffers under the symmetric gate: two per launch (1.4 ms); an odd count >= 3     pair-symmetric + one-sided  ACRR's 5: 3.9 ms
 *       ends with its last three on the one-sided kernel ("sym_r20_f+lds_r20_f", 2.5 ms)
 *   Welch dof (dof = 1), <= 2 RGB G-buffers, any channel rule / border, RGB or float buffers   pair-symmetric Welch build  3.5 ms (1.5 per float buffer)
 *   Welch dof x one-channel G-buffers (depth, material id), RGB or float buffers               eight-plane Welch build     4.2 ms (1.85 per float buffer); block +
 *                                                                                              halo images: the 18-channel layout
 *   a clamped border on a block + halo image (multi-GPU)                                       the pair-symmetric builds + the border kernel, both on the packed image
 *   radius > 20, more than eight feature channels, G-buffers of other channel counts          general kernel ("generic")
 * All of them return the CPU oracle's results to <= 1e-5 (tests/test_gpu_parity.py::test_filter_spec_variants_match_oracle). */
#define STATMC_GATE_SYMMETRIC 0   /* member <=> fma(d, d, -(D_p + D_q)) <= 0, i.e. d^2 <= D_p + D_q        */
#define STATMC_GATE_ASYMMETRIC 1  /* member <=> fma(d, d, -D_q) <= D_p      (this build's spec v1.x)       */
#define STATMC_GATE_CENTRE 2      /* member <=> d * d <= D_p: the neighbour's mean inside the CENTRE pixel's confidence interval
                                     (Moon et al. 2013; what the reference's CUDA source computes under -DMEMFNC=1, README.md:147-150 --
                                     with significance 0.002 and the Box-Cox transform switched off, as that paragraph says) */
#define STATMC_CHANNELS_AND 0     /* every channel of an RGB buffer must pass                              */
#define STATMC_CHANNELS_JOINT 1   /* sum over channels of the left sides <= sum of the right sides        */
#define STATMC_SIDES_TWO 0        /* tabulated quantile t_{1-alpha/2, dof}                                */
#define STATMC_SIDES_ONE 1        /* t_{1-alpha, dof}                                                     */
#define STATMC_DOF_PIXEL 0        /* discriminator = t(n-1)^2 s^2/n per pixel                             */
#define STATMC_DOF_WELCH 1        /* discriminator image = s^2/n; each pair looks t up at floor(Welch-
                                     Satterthwaite dof) and tests d^2 <= t^2 (v_p + v_q)                  */
#define STATMC_BORDER_CLIP 0      /* taps outside the image are skipped                                   */
#define STATMC_BORDER_CLAMP 1     /* tap coordinates are clamped to the image (edge pixels repeat)        */
#define STATMC_SMALL_N_ACCEPT 0   /* n < 2: discriminator +inf, the pixel passes every test               */
#define STATMC_SMALL_N_EXCLUDE 1  /* n < 2: the pixel takes no part in any window                         */
typedef struct statmc_filter_spec {
    int32_t gate, channel_rule, sides, dof, border, small_n;
} statmc_filter_spec;
/* Acts on the current device; applies to every later pre-pass / window-filter call on it. */
int statmc_set_filter_spec(const statmc_filter_spec *spec);
int statmc_get_filter_spec(statmc_filter_spec *spec);
/* The spec and significance level a device has after statmc_setup: include/statmc_pinned_spec.h, which
 * tools/pin_from_dumps.sh rewrites from dumps of the CUDA build (all zero until then).  statmc_reset_filter_spec puts
 * the current device back to it; statmc_pinned_from says where it came from. */
int statmc_reset_filter_spec(void);
const char *statmc_pinned_from(void);

/* ---- window-sweep split (reproducibility across image shapes and devices).  The LDS window filters may sweep the window
 * rows of a work tile with several workgroups ("parts") whose partial sums are added afterwards; how many is chosen per
 * call from the number of tiles of the LOCAL image and the device's CU count, so that the launch fills the chip.  The
 * parts decide how a pixel's 1681 terms are grouped: two calls with a different split agree to <= 1e-6 relative L2, two
 * calls with the same split (and the same film-anchored tile grid, statmc_filter_args::film_x0 / film_y0) agree BIT FOR BIT
 * -- whatever the image shape, the region of interest or the device.  Within one image the split never depends on the
 * region a call filters (bands of the Upload / Denoise / Download pipeline = the whole-image call, bit for bit).
 * A host that needs the blocks of a sharded film and the single-device result to be the same bits pins ONE split on every
 * device involved, the single one included:
 *     statmc_set_filter_split(statmc_filter_split_auto(film_width, film_height, radius)),
 * at the price of a launch that is no longer fitted to the block (a 1920 x 135 strip: + 34 % filter time with the whole
 * film's split of 1).  parts = 0 (default) = automatic: the fitted count, and for tile counts that leave the last round
 * of workgroups mostly empty (1280 x 720: 900 tiles on 256 CUs) more parts for the last tile rows.  Per device; acts on
 * the current device. */
int statmc_set_filter_split(int parts);
int statmc_get_filter_split(void);
/* The split the automatic choice makes on the current device for a whole image of this size (pair-symmetric kernel). */
int statmc_filter_split_auto(int width, int height, int radius);

/* Device memory + copies: the GpuMat role inside Buffer (src/statistics/buffer.h:25,57-63). */
int statmc_malloc(void **dev_ptr, size_t bytes);
int statmc_free(void *dev_ptr);   /* blocks of statmc_malloc and of statmc_malloc_placed alike */

/* ---- Device memory dealt by interference class (MI355X; no counterpart in the reference, whose buffers are plain GpuMats).
 * statmc_accumulate streams a read-once sample arena and read-modify-writes the running moments.  On MI355X every GiB of
 * device memory falls into one of three classes (most likely the three ranks behind every channel of a 12-high HBM3E stack),
 * and a stream that is READ beside WRITES into memory of its own class runs ~ 9 % slower than beside writes into another
 * class (a read-only stream does not care).  With the moments in one class and the
 * sample arenas in the others the 1080p / 256-spp launch of all stat types runs at 0.85 of the HBM peak instead of 0.76
 * (4K / 64 spp: 0.79 instead of 0.68; DESIGN.md section 4.1a, tools/experiments/acc_pool.py fastslow) -- the same kernel, the
 * same bits.  The class travels with the physical memory and HIP does not expose it, so the allocator MEASURES it: one reserved address range per device, backed GiB by GiB, every GiB probed
 * against two GiB of the allocator's own (0.2 ms each; statmc_amd/csrc/statmc_placement.hip).
 *   role STATMC_MEM_STATE   images a kernel reads AND writes per launch: n, mean, m2, m3, film-mean, film-m2 (the first 960 MiB in
 *                           the very GiB every other one is probed against, the rest in slots of its class, A)
 *   role STATMC_MEM_STREAM  read-once inputs: the sample arenas of statmc_accumulate / statmc_accumulate_tiles (all in ONE of the
 *                           other two classes while the card has room: arenas spread over both cost 2 - 3 % of the gain)
 * Blocks are 2-MiB aligned, contiguous in the address space (a block above 2 GiB is GiB slots of one class from anywhere on the
 * card, mapped side by side a second time), freed with statmc_free, and otherwise ordinary device memory.  The first call on
 * a device reserves address space and probes GiB slots until both probe levels have been seen (tens of ms); a GiB slot that
 * holds no live block is idle again (either role may take it), slots of the classes a request cannot use stay backed and idle until
 * statmc_placement_trim.  The search for a class backs at most 3 x the bytes asked for on the device so far (+ 6 GiB;
 * STATMC_PLACEMENT_MAX_GIB=<GiB> sets another budget, never above 60 % of the card) and settles for the other classes after that.
 * Blocks are valid operands of statmc_copy_rect and statmc_halo_exchange across devices: the first copy between two devices grants the
 * owner's blocks to the other device (hipMemSetAccess: hipDeviceEnablePeerAccess does not cover such memory); not IPC-shareable.  Where the probes show no contrast, the device has no virtual-memory management or memory runs short
 * the call still succeeds with memory as it comes (statmc_placement_info says so); STATMC_PLACEMENT=0 in the environment
 * makes it hipMalloc.  Not to be called while a kernel of the caller's runs (the probe competes for the memory system).
 * (ROCm 7.2 / gfx950: hipMemUnmap leaves the shaders' address translation in place -- memory mapped at an address that was mapped
 * before is not what kernels see until the driver rewrites the page tables; the allocator forces that after every unmap, so addresses
 * it uses again -- windows, filled holes -- reach their own memory: tools/microbench/vmm_remap.hip) */
#define STATMC_MEM_STATE 0
#define STATMC_MEM_STREAM 1
int statmc_malloc_placed(void **dev_ptr, size_t bytes, int role);
/* Announces how many bytes the caller is about to ask for in `role` on the current device, in however many blocks: the class search of
 * those calls is budgeted -- and, for STATMC_MEM_STREAM, the arenas' class chosen -- for all of them at once (arena by arena the first
 * 6-GiB arena settles for whichever class has six slots at hand and the later ones for what is left: arenas spread over two classes,
 * 0.77 of the HBM peak instead of 0.805).  Every statmc_malloc_placed of the role counts against it; 0 withdraws it.  Optional. */
int statmc_placement_expect(int role, size_t bytes);
typedef struct statmc_placement_info_t {
    int32_t active;          /* 1: slots are told apart and dealt by class */
    int32_t virtual_memory;  /* 1: the device maps physical allocations into reserved ranges (hipMemCreate / hipMemMap) */
    int32_t slots, probes;   /* GiB slots backed (the allocator's own included), probes run */
    int32_t slots_a, slots_b, slots_c, slots_unclear;   /* by class: A = the allocator's first slot's (STATE), B = its second probe target's (STREAM), C */
    int32_t slots_idle;      /* backed, probed, dealt to no role (yet) */
    int32_t slots_as_they_came[2];   /* per role: slots dealt without the wanted class (no room for better) */
    float fast_probe_ms, slow_probe_ms;
    uint64_t slab_bytes[2], live_bytes[2];   /* per role: bytes of the slots dealt to it / bytes in live blocks */
    int32_t slots_released;  /* holes statmc_placement_trim left in the range (not counted in `slots`) */
    int32_t peer_devices;    /* devices besides the owner that the blocks are mapped for (statmc_copy_rect / statmc_halo_exchange operands) */
    int32_t peak_slots;      /* most GiB slots backed at any one time (what the class searches held before statmc_placement_trim) */
    int32_t rebased;         /* 1: the allocator's reference slot traded places with a slot of another class, because the card's first
                                slots were mostly of the first one's (the moments' home should be the class the card has least of) */
} statmc_placement_info_t;
int statmc_placement_info(statmc_placement_info_t *out);   /* current device */
/* Gives the memory of the idle slots of the current device (backed and probed, dealt to no role: the classes nobody asked for)
 * back to the driver; returns how many, or a negative error.  Synchronises the device.  Later statmc_malloc_placed calls back
 * and probe new slots as they need them (the released addresses first). */
int statmc_placement_trim(void);
/* One character per GiB slot of the current device, NUL-terminated: '#' the allocator's own, a / b / c an idle slot of
 * that class, A / B / C one dealt to a role, S / T one dealt to the state / stream role without the wanted class, '?' unclear,
 * '_' released by statmc_placement_trim. */
int statmc_placement_map(char *out, int capacity);
/* Page-locked host memory for the staging side of statmc_upload / statmc_download (sample arenas
 * of the tile path, dump buffers): copies from it run at the full PCIe rate and stay asynchronous. */
int statmc_malloc_host(void **host_ptr, size_t bytes);
int statmc_free_host(void *host_ptr);
int statmc_memset(void *dev_ptr, int value, size_t bytes, void *stream);
int statmc_upload(void *dev_dst, const void *host_src, size_t bytes, void *stream);   /* Buffer::upload */
int statmc_download(void *host_dst, const void *dev_src, size_t bytes, void *stream); /* Buffer::download */
int statmc_stream_create(void **stream);
/* priority_class 0 normal, > 0 high, < 0 low.  Streams of different classes never share a hardware queue (the runtime
 * keeps one pool of GPU_MAX_HW_QUEUES queues per priority level), so the barrier packets of one cannot hold back the
 * packets of the other: what the band pipeline's copy streams need against its kernel stream (statmc_bands.hpp). */
int statmc_stream_create_with_priority(void **stream, int priority_class);
int statmc_stream_destroy(void *stream);
/* Events: order work across streams without blocking the host -- what lets Estimator::Upload / Denoise / Download
 * (src/statistics/estimator.cpp:409-489) run as a pipeline of row bands on three streams (copies in, kernels, copies
 * out) instead of one after the other.  statmc_stream_wait_event makes everything enqueued on `stream` afterwards
 * wait for the work `event` was recorded behind. */
int statmc_event_create(void **event);
int statmc_event_destroy(void *event);
int statmc_event_record(void *event, void *stream);
int statmc_stream_wait_event(void *stream, void *event);
/* Replaces cv::cuda::stat_denoiser::synchronize(stream)  (src/statistics/estimator.cpp:571-573). */
int statmc_synchronize(void *stream);

/* Device image descriptor: what one cv::cuda::PtrStepSzb entry of the reference's pointer
 * tables carries (src/statistics/estimator.cpp:35-84). */
typedef struct statmc_image {
    void *data;   /* device pointer */
    size_t step;  /* bytes per row: cols * channels * 4 (packed; what the library allocates), or more (a pitched image of the
                   * caller's: filter<T>, pre-pass, window filter and mean-vars run it through a packed twin) */
    int32_t cols; /* width  */
    int32_t rows; /* height */
} statmc_image;

/* Argument block of cv::cuda::stat_denoiser::filter<T>, in the reference's order
 * (src/statistics/estimator.cpp:437-459 for T=float, 465-487 for T=float3).  The reference
 * passes the per-buffer tables as device arrays of PtrStepSzb; here they are host arrays of
 * n_buffers descriptors (they are copied into the kernel arguments). */
typedef struct statmc_filter_args {
    uint8_t n_buffers;        /* floatBufferCounts / rgbBufferCounts [DenoiseGroup] */
    uint16_t width, height;
    float filter_ds_factor;   /* -0.5 / filtersd^2   (estimator.h:259) */
    uint8_t filter_radius;
    uint8_t denoise_film;     /* bool denoiseFilm */
    const statmc_image *n;    /* int32 x1 */
    const statmc_image *mean; /* T */
    const statmc_image *m2;   /* T */
    const statmc_image *m3;   /* T */
    const statmc_image *film; /* T: per-buffer untransformed means (tX-bY-film-mean) */
    statmc_image film_buffer; /* float3 "film" image; colour input of buffer 0 iff denoise_film */
    const statmc_image *g_buffers;     /* n_g_buffers feature-mean images */
    const uint8_t *g_channel_counts;   /* 1 or 3 per G-buffer */
    const float *g_dr_factors;         /* -0.5 / sd_g^2  (estimator.cpp:16) */
    size_t n_g_buffers;
    const statmc_image *mean_corr;     /* out, T */
    const statmc_image *discriminator; /* out, T */
    const statmc_image *film_filtered; /* out, T: tX-bY-film-mean-f */
    statmc_image film_filtered_buffer; /* out, float3 "film-f"; output of buffer 0 iff denoise_film */
    void *stream;
    /* ---- extension (all-zero = reference behaviour: filter the whole image) --------------
     * Output region of interest, used by the multi-GPU block decomposition: outputs are
     * written for x in [roi_x0, roi_x1), y in [roi_y0, roi_y1) only; the window is still
     * clipped to the full [0,width) x [0,height) local image. */
    int32_t roi_x0, roi_y0, roi_x1, roi_y1;
    /* Packed filter inputs (multi-GPU block path; data == NULL = not used).  A [height][width][15]
     * fp32 image holding, per pixel, mean_corr.rgb, discriminator.rgb, colour.rgb, g_buffers[0].rgb,
     * g_buffers[1].rgb -- the layout the halo exchange moves as one message -- or a [height][width][17]
     * image that adds two 1-channel G-buffers behind them (depth, material id: src/statistics/statpath.cpp:828-835,
     * 1096-1130; slots = the call's RGB G-buffers in argument order, then its 1-channel ones; an absent one is 0),
     * or a [height][width][16] image whose channel 15 holds the BITS of the pixel's int32 sample count: the layout for
     * STATMC_DOF_WELCH, whose pair test reads n_p and n_q (two RGB G-buffers; pair-symmetric kernel only; refused under
     * STATMC_DOF_PIXEL, as a 15- or 17-channel image is under STATMC_DOF_WELCH), or a [height][width][18] image with both:
     * channels 15, 16 the two 1-channel G-buffers, channel 17 the count's bits (STATMC_DOF_WELCH with depth / material id
     * among the G-buffers: the eight-plane Welch builds).
     * When set, statmc_window_filter (T = float3, radius <= 20, n_buffers = 1) reads its inputs from it and ignores
     * mean_corr / discriminator / film / g_buffers (g_dr_factors and, for 17 / 18 channels, g_channel_counts still describe
     * the G-buffers: 15 / 16 channels = two RGB; 17 / 18 = up to two RGB + up to two 1-channel, pair-symmetric kernel only).
     * statmc_pack_filter_inputs / statmc_prepass_pack fill the owned block of such an image; the channel count is the
     * image's row pitch / (cols * 4). */
    statmc_image packed_inputs;
    /* Film coordinates of local pixel (0, 0) (multi-GPU block path; 0, 0 = the local image is the film).  The
     * window filter lays its work tiles on a grid fixed in FILM coordinates, so a pixel's sums are formed in the
     * same order whether it is filtered as part of the whole film or of a block + halo image: block-decomposed
     * results equal the single-GPU result bit for bit WHEN BOTH USE THE SAME WINDOW-SWEEP SPLIT (statmc_set_filter_split;
     * the automatic split depends on the local image's shape and the device's CU count), and to <= 1e-6 relative L2
     * under the automatic split. */
    int32_t film_x0, film_y0;
} statmc_filter_args;

/* Replace cv::cuda::stat_denoiser::filter<float> / filter<float3>: pre-pass + window filter. */
int statmc_filter_f32(const statmc_filter_args *args);
int statmc_filter_f32x3(const statmc_filter_args *args);

/* The two halves of filter<T>, separately callable (multi-GPU runs exchange halos between
 * them; the bench times them separately).  channels = 1 or 3 selects T. */
int statmc_prepass(const statmc_filter_args *args, int channels);     /* -> mean_corr, discriminator */
int statmc_window_filter(const statmc_filter_args *args, int channels); /* mean_corr, discriminator -> filtered */

/* Copies the five window-filter inputs of buffer 0 (mean_corr[0], discriminator[0], the colour
 * image -- film_buffer if denoise_film, else film[0] --, g_buffers[0], g_buffers[1]; all
 * width x height x 3) into the 15-channel image `packed` at pixel offset (dst_x0, dst_y0): the
 * owned block inside a block + halo image.  One pass, 60 B read + 60 B written per pixel.  A 17-channel `packed`
 * (row pitch cols * 68) takes up to two RGB and up to two 1-channel G-buffers (g_channel_counts says which); a
 * 16-channel one (row pitch cols * 64) also takes n[0] (channel 15: the count's bits); an 18-channel one (row pitch cols * 72)
 * takes both (the count in channel 17). */
int statmc_pack_filter_inputs(const statmc_filter_args *args, const statmc_image *packed, int dst_x0, int dst_y0);

/* statmc_prepass (T = float3, buffer 0) and statmc_pack_filter_inputs in one pass over the block:
 * reads n, mean, m2, m3, the colour image and the two G-buffers, writes the packed image; mean_corr[0] /
 * discriminator[0] are written as well when those tables are given (the reference keeps them as
 * device images), and skipped when they are NULL. */
int statmc_prepass_pack(const statmc_filter_args *args, const statmc_image *packed, int dst_x0, int dst_y0);
/* ... for n_ranges = 1 or 2 disjoint ascending row ranges {y0, y1} of the block only, in one launch (0 = the whole block). */
int statmc_prepass_pack_rows(const statmc_filter_args *args, const statmc_image *packed, int dst_x0, int dst_y0,
                             const int32_t *ranges, int n_ranges);

/* ---- film blocks on several devices of ONE process (new capability: the reference is single-GPU).
 * The Python side exchanges halos between processes with torch.distributed (RCCL send/recv); this is the same
 * exchange for a C++ host that drives all devices itself -- one Estimator per device -- with device-to-device copies
 * (peer access over xGMI, enabled on demand). */
typedef struct statmc_block {
    int32_t device;      /* HIP device that owns the block (statmc_setup must have run for it) */
    statmc_image packed; /* [block_h + halo rows][block_w + halo columns][15 | 17] fp32 on that device: the block + halo image
                            statmc_prepass_pack fills and statmc_window_filter reads; a side that lies on the film
                            border has no halo */
    void *stream;        /* the block's stream: its pack was enqueued there, the copies into it go there */
} statmc_block;
/* Fills the halo margins of every block's packed image from its neighbours.  blocks[by * gx + bx] = block (bx, by) of
 * a gx x gy grid of equal block_w x block_h blocks; radius = halo width.  Two phases, as in statmc_amd/sharding.py:
 * columns first, then rows over the widened blocks so that the corners ride along; every copy runs on the stream of
 * its destination block and is ordered behind the source block's pack / first phase by events.  Asynchronous.
 * A caller that runs exchange after exchange orders the REWRITE of a block's packed image behind its neighbours' copies
 * out of it itself (an event recorded on each neighbour's stream after this call, awaited before the next pack:
 * statmc_amd/peer.py does). */
int statmc_halo_exchange(const statmc_block *blocks, int gx, int gy, int block_w, int block_h, int radius);
/* ---- ... and between PROCESSES, one per GPU, over RCCL (north_star: "RCCL halo exchange over xGMI ... via a thin C-ABI HIP shim").
 * Every rank owns block (bx, by) = (rank % gx, rank / gx) of the gx x gy grid, packs it (statmc_prepass_pack) on block->stream and
 * calls statmc_halo_exchange_rccl with a communicator of gx * gy ranks whose rank numbers are the block numbers: point-to-point
 * sends and receives with the (at most four) neighbours, columns first, then rows over the widened block so that the corners ride
 * along -- every neighbour one hop over xGMI, no ring, no collective; a row halo is one contiguous message straight out of / into the
 * image, a column halo passes through a staging buffer.  Asynchronous on block->stream (the next statmc_window_filter on that stream
 * sees the halo); only `packed`, `device` and `stream` of *block are used.  `nccl_comm` is an ncclComm_t: the caller's own
 * (ncclCommInitRank of the RCCL the process links) or one made by statmc_rccl_comm_create.  RCCL is bound at run time (dlsym /
 * dlopen of librccl.so.1); STATMC_ERR_UNSUPPORTED where it cannot be found.  Python ranks do the same exchange through
 * torch.distributed (statmc_amd/sharding.py). */
int statmc_halo_exchange_rccl(const statmc_block *block, int gx, int gy, int block_w, int block_h, int radius, void *nccl_comm, int rank);
int statmc_rccl_available(void);                        /* 1: an RCCL library was found and bound */
int statmc_rccl_unique_id(void *id128);                 /* ncclGetUniqueId: 128 bytes, made by one rank, handed to the others by the host's own means */
int statmc_rccl_comm_create(void **nccl_comm, int n_ranks, int rank, const void *id128);   /* ncclCommInitRank on the current device (collective) */
int statmc_rccl_comm_destroy(void *nccl_comm);

/* Rectangle copy between device images of any two devices of the process (block cut / block paste of the sharded
 * path).  elem_bytes = bytes per pixel; runs on `stream`, a stream of either device (peer access is enabled in both
 * directions on first use). */
int statmc_copy_rect(const statmc_image *dst, int dst_device, int dst_x, int dst_y, const statmc_image *src,
                     int src_device, int src_x, int src_y, int width, int height, int elem_bytes, void *stream);

/* Replaces cv::cuda::stat_denoiser::calculateMeanVars<T> (commented-out call,
 * src/statistics/estimator.cpp:501-521) and its CPU stand-in (estimator.cpp:524-568):
 * film_var = film_m2 / ((n-1)*n).  row_n_quirk != 0 reproduces the CPU loop reading n once
 * per row (estimator.cpp:540,558). */
int statmc_calculate_mean_vars(uint8_t n_buffers, uint16_t width, uint16_t height, int channels,
                               const statmc_image *n, const statmc_image *film_m2,
                               const statmc_image *film_var, int row_n_quirk, void *stream);

/* ---- accumulation: the GPU form of StatTile<T>::Add[Transform]SampleM{1,2,3}
 * (src/statistics/estimator.h:162-232) + Estimator::Merge[Transform]Tile
 * (src/statistics/estimator.cpp:341-407) for a whole batch of samples per pixel. ---------- */
typedef struct statmc_stat_type {
    int32_t channels;    /* 1 or 3                      (StatTypeConfig::nChannels) */
    int32_t transform;   /* Box-Cox(0.5) on the sample  (StatTypeConfig::transform) */
    int32_t max_moment;  /* 1, 2 or 3                   (StatTypeConfig::maxMoment) */
    int32_t n_samples;   /* samples per pixel in this batch */
    const float *samples; /* device, [n_samples][height][width][channels] */
    int32_t *n;          /* device state images, updated in place */
    float *mean, *m2, *m3;
    float *film_mean, *film_m2; /* transform types only; non-transform types alias mean/m2
                                   (estimator.cpp:127-137) and may pass NULL here */
    /* Optional (both or neither; max_moment 3): the accumulation's epilogue also writes the Johnson-corrected mean and the
     * discriminator of the UPDATED moments -- exactly what statmc_prepass computes from n / mean / m2 / m3 under the current device's
     * filter spec and significance level, the same bits, from the registers that hold the new moments -- so that a host whose
     * statistics live on one device goes from statmc_accumulate straight to statmc_window_filter: one launch and a 40 B/px read
     * fewer per iteration.  NULL (zero-initialised descriptors): off.  Spec or significance level changed since: call statmc_prepass. */
    float *mean_corr, *discriminator;
} statmc_stat_type;

int statmc_accumulate(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types,
                      void *stream);
/* The same for rows [y0, y1) of the film only (the descriptors still describe the whole film: samples
 * [n_samples][height][width][channels], whole state images).  Per-pixel work, so any split of a batch into row ranges
 * leaves the same bits; the multi-GPU step accumulates the rows next to a neighbour first and the rest while their
 * halo exchange runs. */
int statmc_accumulate_rows(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types, int y0, int y1,
                           void *stream);
/* ... and for n_ranges disjoint row ranges {y0, y1} in one launch (n_types x n_ranges <= 16). */
int statmc_accumulate_row_ranges(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types,
                                 const int32_t *ranges, int n_ranges, void *stream);

/* The same accumulation fed tile by tile, the way StatPathIntegrator::Render produces samples
 * (src/statistics/statpath.cpp:132-190: 16 x 16 tiles; 355-371: every sample of a pixel is handed to
 * the tile of every stat type; 381-388: Merge*Tiles once per tile and iteration).  Every type's
 * `samples` is an arena in which tile k owns the block starting at float offset
 * tile_offsets[k] * channels, laid out [tile_samples[k]][y1-y0][x1-x0][channels]: tile_samples[k]
 * samples for each pixel of the tile (the same count for every type; 0 = tile untouched).
 * tile_bounds = {x0,y0,x1,y1} per tile, tiles disjoint and inside the image.  All three tables are
 * device arrays.  `n_samples` of the types is ignored.  Blocks whose offset, origin and width are
 * multiples of 4 pixels (16 x 16 tiles of an image whose width is) take the vector path. */
int statmc_accumulate_tiles(uint16_t width, uint16_t height, const statmc_stat_type *types, int n_types,
                            const int32_t *tile_bounds, const int64_t *tile_offsets,
                            const int32_t *tile_samples, int n_tiles, void *stream);

/* Scatter of reference-layout AoS tiles (StatTilePixel<T>, estimator.h:104-124: 64 B for
 * T=float, 128 B for T=Vec3) that were accumulated on the host into the planar device images:
 * Estimator::MergeTile / MergeTransformTile.  tile_bounds = {x0,y0,x1,y1} per tile (device,
 * int32 x4), tile_offsets = index of each tile's first pixel in tile_pixels (device). */
int statmc_merge_tiles(uint16_t width, uint16_t height, int channels, int transform,
                       const void *tile_pixels, const int32_t *tile_bounds,
                       const int64_t *tile_offsets, int n_tiles, int max_tile_pixels,
                       int32_t *n, float *mean, float *m2, float *m3, float *film_mean,
                       float *film_m2, void *stream);

/* Film::UpdateImage on the device (src/core/film.cpp:188-222): reads the reference's AoS
 * Film::Pixel array {float xyz[3]; float filterWeightSum; float splatXYZ[3]; float pad} (32 B per
 * pixel, src/core/film.h:72-78) and writes the interleaved RGB "film" image the filter denoises:
 * XYZ -> RGB, / weight sum, clamp >= 0, + splat_scale * splat RGB, * scale. */
int statmc_film_update(const void *film_pixels, size_t n_pixels, float splat_scale, float scale, float *film_rgb,
                       void *stream);

/* Tile-local pooled moments of an image by wavefront-level Welford/Chan merges: for every
 * tile_size x tile_size tile writes {count, mean, M2} per channel of `values`
 * (out: [tiles_y][tiles_x][channels][3] fp32).  tile_size in {8, 16}. */
int statmc_tile_moments(uint16_t width, uint16_t height, int channels, const float *values,
                        int tile_size, float *out, void *stream);

/* Introspection used by tests/bench: name of the window-filter kernel variant the last
 * statmc_window_filter call dispatched ("lds_r20", "lds_rt", "generic", ...). */
const char *statmc_last_filter_variant(void);
int statmc_version(void);

/* Measurement aid (bench.py "shader_clock"; no counterpart in the reference): one wave counts `cycles` shader clocks
 * (s_memtime) against the constant 100 MHz clock (s_memrealtime) and writes {shader clocks, 10 ns ticks} to
 * out[0..1] (device, int64).  Enqueued right behind a kernel it reports the clock the power manager held for that
 * kernel.  The loop ends after `cycles` shader clocks (1 .. 2^24). */
int statmc_clock_probe(int64_t *out, int cycles, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STATMC_H */
