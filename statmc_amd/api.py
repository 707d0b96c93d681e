"""ctypes binding of libstatmc_hip.so (include/statmc.h) + thin helpers that marshal torch
tensors (device memory, streams) into the C structs.  Plumbing only: every number is produced
by the HIP kernels behind the C ABI.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))

STATMC_OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE = -1, -2, -3, -4
DOF_PIXEL, DOF_WELCH = 0, 1      # statmc_filter_spec.dof (include/statmc.h)
MAX_BUFFERS, MAX_GBUFFERS = 16, 8


class StatmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("statmc error %d: %s" % (code, msg))
        self.code = code


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("step", C.c_size_t), ("cols", C.c_int32), ("rows", C.c_int32)]


class FilterArgs(C.Structure):
    _fields_ = [
        ("n_buffers", C.c_uint8), ("width", C.c_uint16), ("height", C.c_uint16),
        ("filter_ds_factor", C.c_float), ("filter_radius", C.c_uint8), ("denoise_film", C.c_uint8),
        ("n", C.POINTER(Image)), ("mean", C.POINTER(Image)), ("m2", C.POINTER(Image)),
        ("m3", C.POINTER(Image)), ("film", C.POINTER(Image)), ("film_buffer", Image),
        ("g_buffers", C.POINTER(Image)), ("g_channel_counts", C.POINTER(C.c_uint8)),
        ("g_dr_factors", C.POINTER(C.c_float)), ("n_g_buffers", C.c_size_t),
        ("mean_corr", C.POINTER(Image)), ("discriminator", C.POINTER(Image)),
        ("film_filtered", C.POINTER(Image)), ("film_filtered_buffer", Image),
        ("stream", C.c_void_p),
        ("roi_x0", C.c_int32), ("roi_y0", C.c_int32), ("roi_x1", C.c_int32), ("roi_y1", C.c_int32),
        ("packed_inputs", Image),
        ("film_x0", C.c_int32), ("film_y0", C.c_int32),
    ]


class FilterSpec(C.Structure):
    """statmc_filter_spec: what the reference tree leaves open about filter<T>.  All-zero = default."""
    _fields_ = [("gate", C.c_int32), ("channel_rule", C.c_int32), ("sides", C.c_int32), ("dof", C.c_int32),
                ("border", C.c_int32), ("small_n", C.c_int32)]

    def __init__(self, gate=0, channel_rule=0, sides=0, dof=0, border=0, small_n=0):
        super().__init__(gate, channel_rule, sides, dof, border, small_n)

    def as_tuple(self):
        return (self.gate, self.channel_rule, self.sides, self.dof, self.border, self.small_n)


class PlacementInfo(C.Structure):
    _fields_ = [("active", C.c_int32), ("virtual_memory", C.c_int32), ("slots", C.c_int32), ("probes", C.c_int32),
                ("slots_a", C.c_int32), ("slots_b", C.c_int32), ("slots_c", C.c_int32), ("slots_unclear", C.c_int32), ("slots_idle", C.c_int32),
                ("slots_as_they_came", C.c_int32 * 2), ("fast_probe_ms", C.c_float), ("slow_probe_ms", C.c_float),
                ("slab_bytes", C.c_uint64 * 2), ("live_bytes", C.c_uint64 * 2), ("slots_released", C.c_int32), ("peer_devices", C.c_int32),
                ("peak_slots", C.c_int32), ("rebased", C.c_int32)]


MEM_STATE, MEM_STREAM = 0, 1


class StatType(C.Structure):
    _fields_ = [
        ("channels", C.c_int32), ("transform", C.c_int32), ("max_moment", C.c_int32),
        ("n_samples", C.c_int32), ("samples", C.c_void_p), ("n", C.c_void_p),
        ("mean", C.c_void_p), ("m2", C.c_void_p), ("m3", C.c_void_p),
        ("film_mean", C.c_void_p), ("film_m2", C.c_void_p),
        ("mean_corr", C.c_void_p), ("discriminator", C.c_void_p),
    ]


EXPORTS = [
    "statmc_last_error", "statmc_setup", "statmc_device_cus", "statmc_set_device", "statmc_set_significance", "statmc_get_significance", "statmc_set_t_quantiles",
    "statmc_set_filter_spec", "statmc_get_filter_spec", "statmc_reset_filter_spec", "statmc_pinned_from", "statmc_copy_device_settings",
    "statmc_set_filter_split", "statmc_get_filter_split", "statmc_filter_split_auto",
    "statmc_malloc", "statmc_free", "statmc_malloc_placed", "statmc_placement_expect", "statmc_placement_info", "statmc_placement_map", "statmc_placement_trim", "statmc_malloc_host", "statmc_free_host", "statmc_memset", "statmc_upload", "statmc_download",
    "statmc_stream_create", "statmc_stream_create_with_priority", "statmc_stream_destroy", "statmc_synchronize",
    "statmc_event_create", "statmc_event_destroy", "statmc_event_record", "statmc_stream_wait_event",
    "statmc_filter_f32", "statmc_filter_f32x3", "statmc_prepass", "statmc_window_filter", "statmc_pack_filter_inputs", "statmc_prepass_pack", "statmc_prepass_pack_rows",
    "statmc_halo_exchange", "statmc_halo_exchange_rccl", "statmc_rccl_available", "statmc_rccl_unique_id", "statmc_rccl_comm_create", "statmc_rccl_comm_destroy", "statmc_copy_rect", "statmc_calculate_mean_vars", "statmc_accumulate", "statmc_accumulate_rows", "statmc_accumulate_row_ranges", "statmc_accumulate_tiles", "statmc_merge_tiles", "statmc_tile_moments", "statmc_film_update",
    "statmc_last_filter_variant", "statmc_version", "statmc_clock_probe",
]

_lib = None


def library_path():
    return _build.SO


def load():
    """Load libstatmc_hip.so.  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_build.SO):
        raise RuntimeError(
            "libstatmc_hip.so is missing (%s). Build it with `python -m statmc_amd.build` or "
            "__graft_entry__.build(); statmc_amd has no CPU/PyTorch fallback." % _build.SO)
    if _build.SO == _build.DEFAULT_SO and _build.stale():
        raise RuntimeError("libstatmc_hip.so was built from other sources than the ones in statmc_amd/csrc "
                           "(content hash mismatch): rebuild with `python -m statmc_amd.build` before using it")
    lib = C.CDLL(_build.SO)
    lib.statmc_last_error.restype = C.c_char_p
    lib.statmc_last_filter_variant.restype = C.c_char_p
    lib.statmc_pinned_from.restype = C.c_char_p
    lib.statmc_setup.argtypes = [C.c_int]
    lib.statmc_set_significance.argtypes = [C.c_int]
    lib.statmc_set_t_quantiles.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.statmc_set_filter_spec.argtypes = [C.POINTER(FilterSpec)]
    lib.statmc_get_filter_spec.argtypes = [C.POINTER(FilterSpec)]
    lib.statmc_copy_device_settings.argtypes = [C.c_int, C.c_int]
    lib.statmc_set_filter_split.argtypes = [C.c_int]
    lib.statmc_filter_split_auto.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.statmc_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.statmc_free.argtypes = [C.c_void_p]
    lib.statmc_malloc_placed.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]
    lib.statmc_placement_expect.argtypes = [C.c_int, C.c_size_t]
    lib.statmc_placement_info.argtypes = [C.POINTER(PlacementInfo)]
    lib.statmc_malloc_host.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.statmc_free_host.argtypes = [C.c_void_p]
    lib.statmc_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    lib.statmc_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.statmc_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.statmc_stream_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.statmc_stream_create_with_priority.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.statmc_stream_destroy.argtypes = [C.c_void_p]
    lib.statmc_event_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.statmc_event_destroy.argtypes = [C.c_void_p]
    lib.statmc_event_record.argtypes = [C.c_void_p, C.c_void_p]
    lib.statmc_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
    lib.statmc_synchronize.argtypes = [C.c_void_p]
    lib.statmc_filter_f32.argtypes = [C.POINTER(FilterArgs)]
    lib.statmc_filter_f32x3.argtypes = [C.POINTER(FilterArgs)]
    lib.statmc_prepass.argtypes = [C.POINTER(FilterArgs), C.c_int]
    lib.statmc_window_filter.argtypes = [C.POINTER(FilterArgs), C.c_int]
    lib.statmc_pack_filter_inputs.argtypes = [C.POINTER(FilterArgs), C.POINTER(Image), C.c_int, C.c_int]
    lib.statmc_prepass_pack.argtypes = [C.POINTER(FilterArgs), C.POINTER(Image), C.c_int, C.c_int]
    lib.statmc_prepass_pack_rows.argtypes = [C.POINTER(FilterArgs), C.POINTER(Image), C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]
    lib.statmc_halo_exchange.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.statmc_halo_exchange_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.statmc_rccl_unique_id.argtypes = [C.c_void_p]
    lib.statmc_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
    lib.statmc_rccl_comm_destroy.argtypes = [C.c_void_p]
    lib.statmc_copy_rect.argtypes = [C.POINTER(Image), C.c_int, C.c_int, C.c_int, C.POINTER(Image), C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.statmc_set_device.argtypes = [C.c_int]
    lib.statmc_calculate_mean_vars.argtypes = [C.c_uint8, C.c_uint16, C.c_uint16, C.c_int,
                                               C.POINTER(Image), C.POINTER(Image), C.POINTER(Image),
                                               C.c_int, C.c_void_p]
    lib.statmc_accumulate.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(StatType), C.c_int, C.c_void_p]
    lib.statmc_accumulate_rows.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(StatType), C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.statmc_accumulate_row_ranges.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(StatType), C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p]
    lib.statmc_accumulate_tiles.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(StatType), C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_void_p]
    lib.statmc_merge_tiles.argtypes = [C.c_uint16, C.c_uint16, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p]
    lib.statmc_tile_moments.argtypes = [C.c_uint16, C.c_uint16, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p]
    lib.statmc_film_update.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.statmc_clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.statmc_debug_force_filter_variant.argtypes = [C.c_int]
    lib.statmc_debug_force_filter_parts.argtypes = [C.c_int]
    lib.statmc_debug_last_filter_parts.restype = C.c_int
    lib.statmc_debug_accumulate_resident_blocks.argtypes = [C.c_int]
    lib.statmc_debug_accumulate_dma.argtypes = [C.c_int]
    if lib.statmc_debug_diagnostic_build() and os.environ.get("STATMC_ALLOW_DIAGNOSTIC_BUILD") != "1":
        raise RuntimeError("%s was built with STATMC_SYM_* experiment switches (bits %#x): its filter results are not the "
                           "product's.  Set STATMC_ALLOW_DIAGNOSTIC_BUILD=1 to load it anyway (tools/experiments only)."
                           % (_build.SO, lib.statmc_debug_diagnostic_build()))
    _lib = lib
    return lib


def check(rc):
    if rc != STATMC_OK:
        raise StatmcError(rc, load().statmc_last_error().decode())
    return rc


_setup_done = set()


def setup(device=0):
    if device not in _setup_done:
        check(load().statmc_setup(int(device)))
        _setup_done.add(device)


def set_filter_spec(spec=None, **fields):
    """Filter spec of the current device: a FilterSpec, or its fields by keyword; no argument = the pinned default
    (and the pinned significance level)."""
    if spec is None and not fields:     # what a freshly set-up device has (include/statmc_pinned_spec.h)
        check(load().statmc_reset_filter_spec())
        return
    spec = spec if spec is not None else FilterSpec(**fields)
    check(load().statmc_set_filter_spec(C.byref(spec)))


def get_significance():
    """The current device's significance-level index (statmc_get_significance: 0 = 0.005, 1 = 0.002, 2 = 0.05)."""
    return int(load().statmc_get_significance())


def get_filter_spec():
    spec = FilterSpec()
    check(load().statmc_get_filter_spec(C.byref(spec)))
    return spec


def last_filter_variant():
    return load().statmc_last_filter_variant().decode()


def force_filter_variant(v):
    """0 auto, 1 generic (global-memory) kernel, 2 runtime-radius one-sided LDS kernel, 3 one-sided r = 20 LDS kernel.
    (include/statmc_debug.h; per device, like the two switches below)"""
    check(load().statmc_debug_force_filter_variant(int(v)))


def accumulate_resident_blocks(n):
    """0: by shape (default); n > 0: the accumulate kernel runs as n resident workgroups; -1: never a resident grid."""
    check(load().statmc_debug_accumulate_resident_blocks(int(n)))


def accumulate_dma(on):
    """1 (default): the RGB sample planes of the accumulation stream through LDS-DMA; 0: loads into registers (A/B, tests)."""
    check(load().statmc_debug_accumulate_dma(int(on)))


def force_filter_parts(k):
    """0 automatic; k >= 1: the LDS kernels sweep the window with k workgroups per tile (= set_filter_split)."""
    check(load().statmc_set_filter_split(int(k)))


def set_filter_split(parts):
    """Window-sweep split of the current device: 0 = automatic (fitted to the local image and the device), k >= 1 = pinned.
    Calls with the same split reproduce each other bit for bit whatever the image shape (include/statmc.h)."""
    check(load().statmc_set_filter_split(int(parts)))


def get_filter_split():
    return int(load().statmc_get_filter_split())


def filter_split_auto(width, height, radius):
    """What the automatic choice picks on the current device for a whole image of this size."""
    k = int(load().statmc_filter_split_auto(int(width), int(height), int(radius)))
    if k < 0:
        check(k)
    return k


# ------------------------------------------------------------------ torch marshalling
def _torch():
    import torch
    return torch


def current_stream_handle():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def image_of(t):
    """Describe a [H, W] or [H, W, C] device tensor as a statmc_image: packed, or with a row pitch (a view such as
    padded[:, :W] of a wider tensor -- what a cv::cuda::GpuMat allocated with a pitch looks like)."""
    assert t.is_cuda and t.element_size() == 4
    h, w = t.shape[0], t.shape[1]
    c = t.shape[2] if t.dim() == 3 else 1
    inner = (t.stride(1) == c and t.stride(2) == 1) if t.dim() == 3 else t.stride(1) == 1
    assert inner and t.stride(0) >= w * c, "device images must have packed pixels and a row pitch of at least one row"
    return Image(C.c_void_p(t.data_ptr()), t.stride(0) * 4, w, h)


def _img_array(tensors):
    arr = (Image * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = image_of(t)
    return arr


def make_filter_args(n, mean, m2, m3, film, mean_corr, disc, film_filtered, g_buffers, g_sds=None,
                     g_dr=None, filter_sd=10.0, radius=20, denoise_film=False, film_buffer=None,
                     film_filtered_buffer=None, roi=None, stream=None, keep=None, packed=None, film_origin=None,
                     packed_g_channels=None):
    """Build a statmc_filter_args from lists of per-buffer device tensors (reference argument
    order, estimator.cpp:437-459).  Returns (args, keepalive).  packed: optional [H, W, 15 | 16 | 17]
    block + halo tensor the window filter reads instead of the separate images (16: + the sample count, for Welch degrees
    of freedom; 17: packed_g_channels
    names the channel count of every G-buffer in it, e.g. [3, 3, 1, 1], one g_sds / g_dr entry each)."""
    tables = [mean_corr, mean, film, film_filtered, n]
    nb = max(len(t) for t in tables) if packed is None else 1
    ref = next(t[0] for t in tables if t) if packed is None else packed
    h, w = ref.shape[0], ref.shape[1]
    a = FilterArgs()
    ka = []
    a.n_buffers, a.width, a.height = nb, w, h
    a.filter_ds_factor = -0.5 / (filter_sd * filter_sd)   # estimator.h:259
    a.filter_radius = radius
    a.denoise_film = 1 if denoise_film else 0
    for name, lst in (("n", n), ("mean", mean), ("m2", m2), ("m3", m3), ("film", film),
                      ("mean_corr", mean_corr), ("discriminator", disc), ("film_filtered", film_filtered)):
        if lst:
            arr = _img_array(lst)
            ka.append(arr)
            setattr(a, name, arr)
    if film_buffer is not None:
        a.film_buffer = image_of(film_buffer)
    if film_filtered_buffer is not None:
        a.film_filtered_buffer = image_of(film_filtered_buffer)
    ng = len(g_buffers)
    if ng:
        garr = _img_array(g_buffers)
        gch = (C.c_uint8 * ng)(*[(g.shape[2] if g.dim() == 3 else 1) for g in g_buffers])
        if g_dr is None:
            g_dr = [-0.5 / (sd * sd) for sd in g_sds]      # estimator.cpp:16
        gdr = (C.c_float * ng)(*g_dr)
        ka += [garr, gch, gdr]
        a.g_buffers, a.g_channel_counts, a.g_dr_factors = garr, gch, gdr
    a.n_g_buffers = ng
    if packed is not None:
        pch = packed.shape[2]
        a.packed_inputs = Image(C.c_void_p(packed.data_ptr()), w * pch * 4, w, h)
        if g_dr is None:
            g_dr = [-0.5 / (sd * sd) for sd in g_sds]
        gch_list = list(packed_g_channels) if packed_g_channels is not None else [3] * len(g_dr)
        assert len(gch_list) == len(g_dr)
        gdr = (C.c_float * len(g_dr))(*g_dr)
        gch = (C.c_uint8 * len(g_dr))(*gch_list)
        ka += [gdr, gch]
        a.g_dr_factors, a.g_channel_counts, a.n_g_buffers = gdr, gch, len(g_dr)
    a.stream = stream if stream is not None else current_stream_handle()
    if roi is not None:
        a.roi_x0, a.roi_y0, a.roi_x1, a.roi_y1 = roi
    if film_origin is not None:   # film coordinates of local pixel (0, 0): block + halo images of the multi-GPU path
        a.film_x0, a.film_y0 = film_origin
    ka.append((n, mean, m2, m3, film, mean_corr, disc, film_filtered, g_buffers, film_buffer,
               film_filtered_buffer, packed))
    return a, ka


def prepass(args, channels):
    check(load().statmc_prepass(C.byref(args), channels))


def window_filter(args, channels):
    check(load().statmc_window_filter(C.byref(args), channels))


def pack_filter_inputs(args, packed, dst_x0, dst_y0):
    """Owned block of the five filter inputs -> [Hp, Wp, 15 | 16 | 17] block + halo tensor at (dst_x0, dst_y0)."""
    img = image_of(packed)
    check(load().statmc_pack_filter_inputs(C.byref(args), C.byref(img), dst_x0, dst_y0))


def prepass_pack(args, packed, dst_x0, dst_y0, rows=None):
    """Pre-pass of buffer 0 + pack of the five filter inputs in one pass (mean_corr / discriminator are
    also written to their own images when the args carry them).  rows: one or two (y0, y1) ranges of the block -- only
    those rows, in one launch."""
    img = image_of(packed)
    if rows is None:
        check(load().statmc_prepass_pack(C.byref(args), C.byref(img), dst_x0, dst_y0))
    else:
        flat = (C.c_int32 * (2 * len(rows)))(*[int(v) for r in rows for v in r])
        check(load().statmc_prepass_pack_rows(C.byref(args), C.byref(img), dst_x0, dst_y0, flat, len(rows)))


def filter_f32x3(args):
    check(load().statmc_filter_f32x3(C.byref(args)))


def filter_f32(args):
    check(load().statmc_filter_f32(C.byref(args)))


class _PlacedBlock:
    """A statmc_malloc_placed block seen by torch through __cuda_array_interface__; freed (statmc_free) when the last
    tensor that aliases it goes away."""

    def __init__(self, nbytes, role, shape, typestr):
        p = C.c_void_p()
        check(load().statmc_malloc_placed(C.byref(p), max(int(nbytes), 1), int(role)))
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if self.ptr:
                load().statmc_free(C.c_void_p(self.ptr))
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass
        self.ptr = None


def empty_placed(shape, dtype, device, role):
    """torch tensor on `device` in memory placed by HBM rank (include/statmc.h: statmc_malloc_placed): role MEM_STATE for
    the images a launch reads and writes (the running moments), MEM_STREAM for read-once sample arenas.  Uninitialised."""
    import torch
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1", torch.int64: "<i8"}[dtype]
    n = 1
    for d in shape:
        n *= int(d)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        blk = _PlacedBlock(n * torch.empty((), dtype=dtype).element_size(), role, shape if n else (0,), typestr)
        if n == 0:
            return torch.empty(shape, dtype=dtype, device=device)
        return torch.as_tensor(blk, device=device)


def placement_expect(role, nbytes, device=None):
    """Announces the bytes about to be asked for in `role` (include/statmc.h: the class search is budgeted, and the arenas' class chosen, for all
    of them at once)."""
    import torch
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    with torch.cuda.device(idx):
        check(load().statmc_placement_expect(int(role), int(nbytes)))


def zeros_placed(shape, dtype, device, role):
    return empty_placed(shape, dtype, device, role).zero_()


def placement_info():
    info = PlacementInfo()
    check(load().statmc_placement_info(C.byref(info)))
    out = {k: (list(getattr(info, k)) if k in ("slots_as_they_came", "slab_bytes", "live_bytes") else getattr(info, k)) for k, _ in PlacementInfo._fields_}
    buf = C.create_string_buffer(1024)
    check(load().statmc_placement_map(buf, 1024))
    out["map"] = buf.value.decode()
    return out


class RcclComm:
    """An RCCL communicator made through the C ABI (statmc_rccl_unique_id / statmc_rccl_comm_create): what a multi-process C++
    host uses for statmc_halo_exchange_rccl.  rank 0 makes the id (RcclComm.unique_id()), the host hands its 128 bytes to the
    other ranks by its own means, every rank constructs RcclComm(n_ranks, rank, id) on its device (collective)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(load().statmc_rccl_unique_id(buf))
        return buf.raw

    def __init__(self, n_ranks, rank, id128):
        assert len(id128) == 128
        self.n_ranks, self.rank = n_ranks, rank
        self.handle = C.c_void_p()
        check(load().statmc_rccl_comm_create(C.byref(self.handle), n_ranks, rank, C.c_char_p(id128)))

    def destroy(self):
        if self.handle:
            check(load().statmc_rccl_comm_destroy(self.handle))
            self.handle = C.c_void_p()


def halo_exchange_rccl(packed, device_index, gx, gy, block_w, block_h, radius, comm, stream=None):
    """statmc_halo_exchange_rccl on this rank's block + halo image `packed` ([rows, cols, channels] device tensor)."""
    from .peer import Block
    blk = Block()
    blk.device = device_index
    blk.packed = image_of(packed)
    blk.stream = stream if stream is not None else current_stream_handle()
    check(load().statmc_halo_exchange_rccl(C.byref(blk), gx, gy, block_w, block_h, radius, comm.handle, comm.rank))


def placement_trim():
    """Releases the idle slots of the current device's placed allocator (statmc_placement_trim); returns how many."""
    n = load().statmc_placement_trim()
    if n < 0:
        check(n)
    return n


def make_stat_type(samples, state, transform, max_moment, prepass_into=None):
    """samples: [S, H, W, C] device tensor; state: dict of device tensors n/mean/m2/m3/film_mean/film_m2.
    prepass_into = (mean_corr, discriminator): the accumulation's epilogue also writes the pre-pass of the updated moments there
    (statmc_stat_type::mean_corr / discriminator; max_moment 3)."""
    t = StatType()
    c = samples.shape[3] if samples.dim() == 4 else 1
    t.channels, t.transform, t.max_moment = c, int(bool(transform)), int(max_moment)
    t.n_samples = samples.shape[0]
    t.samples = samples.data_ptr()
    t.n = state["n"].data_ptr()
    t.mean = state["mean"].data_ptr()
    t.m2 = state["m2"].data_ptr() if state.get("m2") is not None else None
    t.m3 = state["m3"].data_ptr() if state.get("m3") is not None else None
    t.film_mean = state["film_mean"].data_ptr() if state.get("film_mean") is not None else None
    t.film_m2 = state["film_m2"].data_ptr() if state.get("film_m2") is not None else None
    if prepass_into is not None:
        t.mean_corr, t.discriminator = prepass_into[0].data_ptr(), prepass_into[1].data_ptr()
    return t


def make_stat_type_arena(arena, channels, state, transform, max_moment, prepass_into=None):
    """Stat type whose samples arrive tile by tile (accumulate_tiles): `arena` is a flat fp32 device tensor."""
    t = make_stat_type(arena.view(1, 1, -1, 1), state, transform, max_moment, prepass_into=prepass_into)
    t.channels, t.n_samples = int(channels), 0
    return t


def accumulate(width, height, stat_types, stream=None, rows=None):
    """rows = (y0, y1): only those rows of the film (statmc_accumulate_rows); rows = [(y0, y1), ...]: several disjoint
    ranges in one launch (statmc_accumulate_row_ranges)."""
    arr = (StatType * max(len(stat_types), 1))(*stat_types)
    st = stream if stream is not None else current_stream_handle()
    if rows is None:
        check(load().statmc_accumulate(width, height, arr, len(stat_types), st))
    elif len(rows) == 2 and not hasattr(rows[0], "__len__"):
        check(load().statmc_accumulate_rows(width, height, arr, len(stat_types), int(rows[0]), int(rows[1]), st))
    else:
        flat = (C.c_int32 * (2 * len(rows)))(*[int(v) for r in rows for v in r])
        check(load().statmc_accumulate_row_ranges(width, height, arr, len(stat_types), flat, len(rows), st))


def accumulate_tiles(width, height, stat_types, tile_bounds, tile_offsets, tile_samples, stream=None):
    """stat_types: make_stat_type(arena, state, ...) per type, `arena` a flat device tensor holding the
    tiles' sample blocks; tile_bounds int32 [n,4], tile_offsets int64 [n] (pixel-samples),
    tile_samples int32 [n]: device tensors."""
    arr = (StatType * max(len(stat_types), 1))(*stat_types)
    check(load().statmc_accumulate_tiles(width, height, arr, len(stat_types), tile_bounds.data_ptr(),
                                         tile_offsets.data_ptr(), tile_samples.data_ptr(), tile_bounds.shape[0],
                                         stream if stream is not None else current_stream_handle()))


def calculate_mean_vars(n, film_m2, film_var, row_n_quirk=True, stream=None):
    h, w = n[0].shape
    c = film_m2[0].shape[2] if film_m2[0].dim() == 3 else 1
    check(load().statmc_calculate_mean_vars(len(n), w, h, c, _img_array(n), _img_array(film_m2),
                                            _img_array(film_var), int(row_n_quirk),
                                            stream if stream is not None else current_stream_handle()))


def merge_tiles(width, height, channels, transform, tile_pixels, tile_bounds, tile_offsets, max_tile_pixels,
                state, stream=None):
    fm = state["film_mean"].data_ptr() if transform else None
    f2 = state["film_m2"].data_ptr() if transform else None
    check(load().statmc_merge_tiles(width, height, channels, int(bool(transform)), tile_pixels.data_ptr(),
                                    tile_bounds.data_ptr(), tile_offsets.data_ptr(), tile_bounds.shape[0],
                                    max_tile_pixels, state["n"].data_ptr(), state["mean"].data_ptr(),
                                    state["m2"].data_ptr(), state["m3"].data_ptr(), fm, f2,
                                    stream if stream is not None else current_stream_handle()))


def film_update(film_pixels, n_pixels, film_rgb, splat_scale=1.0, scale=1.0, stream=None):
    """film_pixels: uint8 device tensor holding n_pixels Film::Pixel structs (32 B each)."""
    check(load().statmc_film_update(film_pixels.data_ptr(), n_pixels, float(splat_scale), float(scale), film_rgb.data_ptr(),
                                    stream if stream is not None else current_stream_handle()))


def tile_moments(values, tile_size, out, stream=None):
    h, w = values.shape[0], values.shape[1]
    c = values.shape[2] if values.dim() == 3 else 1
    check(load().statmc_tile_moments(w, h, c, values.data_ptr(), tile_size, out.data_ptr(),
                                     stream if stream is not None else current_stream_handle()))


def clock_probe(out, cycles=200000, stream=None):
    """out: int64 device tensor of 2 elements <- {shader clocks counted, 10 ns ticks they took}."""
    assert out.is_cuda and out.numel() >= 2 and out.element_size() == 8
    check(load().statmc_clock_probe(out.data_ptr(), int(cycles), stream if stream is not None else current_stream_handle()))
