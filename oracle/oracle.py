"""ctypes binding of the CPU oracle (oracle/libstatmc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (statmc_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libstatmc_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return _SO


class FilterSpec(C.Structure):
    """oracle_filter_spec: the choices SURVEY.md App. B leaves open.  All-zero = default (spec v2)."""
    _fields_ = [("gate", C.c_int32), ("channel_rule", C.c_int32), ("sides", C.c_int32), ("dof", C.c_int32),
                ("border", C.c_int32), ("small_n", C.c_int32)]

    def __init__(self, gate=0, channel_rule=0, sides=0, dof=0, border=0, small_n=0):
        super().__init__(gate, channel_rule, sides, dof, border, small_n)

    def as_tuple(self):
        return (self.gate, self.channel_rule, self.sides, self.dof, self.border, self.small_n)


GATE_SYMMETRIC, GATE_ASYMMETRIC, GATE_CENTRE = 0, 1, 2
CHANNELS_AND, CHANNELS_JOINT = 0, 1
SIDES_TWO, SIDES_ONE = 0, 1
DOF_PIXEL, DOF_WELCH = 0, 1
BORDER_CLIP, BORDER_CLAMP = 0, 1
SMALL_N_ACCEPT, SMALL_N_EXCLUDE = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        _lib.oracle_box_cox.restype = C.c_float
        _lib.oracle_box_cox.argtypes = [C.c_float, C.c_float]
        _lib.oracle_t_quantile.restype = C.c_float
        _lib.oracle_t_quantile.argtypes = [C.c_int, C.c_int]
        _lib.oracle_add_sample.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int]
        _lib.oracle_accumulate_image.argtypes = [C.c_int] * 6 + [f32p, i32p] + [f32p] * 5 + [C.c_int] * 2
        _lib.oracle_accumulate_tile_stream.argtypes = [C.c_int] * 6 + [f32p, i32p] + [f32p] * 5 + [C.c_int] * 2
        _lib.oracle_merge_tile.argtypes = [C.c_void_p] + [C.c_int] * 6 + [i32p] + [f32p] * 5
        _lib.oracle_mean_vars.argtypes = [C.c_int] * 3 + [i32p, f32p, f32p, C.c_int]
        _lib.oracle_prepass.argtypes = [C.c_int] * 4 + [i32p] + [f32p] * 5
        _lib.oracle_filter.argtypes = ([C.c_int] * 3 + [C.c_float, C.c_int] + [f32p] * 3 +
                                       [C.c_int, C.POINTER(f32p), C.POINTER(C.c_int), f32p, f32p] +
                                       [C.c_int] * 5)
        _lib.oracle_set_t_quantiles.argtypes = [C.c_int, f32p, C.c_int]
        _lib.oracle_set_fp_contract.argtypes = [C.c_int]
        _lib.oracle_prepass_spec.argtypes = [C.c_int] * 4 + [C.POINTER(FilterSpec), i32p] + [f32p] * 5
        _lib.oracle_filter_spec_run.argtypes = ([C.c_int] * 3 + [C.c_float, C.c_int, C.c_int, C.POINTER(FilterSpec), i32p] +
                                                [f32p] * 3 + [C.c_int, C.POINTER(f32p), C.POINTER(C.c_int), f32p, f32p] +
                                                [C.c_int] * 5)
        _lib.oracle_film_update.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, f32p]
        _lib.oracle_num_threads.restype = C.c_int
        _lib.oracle_tile_moments.argtypes = [C.c_int] * 3 + [f32p, C.c_int, f32p]
    return _lib


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def num_threads():
    return lib().oracle_num_threads()


def set_fp_contract(on):
    """True: the accumulation / film arithmetic fuses what clang -O3 -march=native -ffp-contract=on (the
    reference's own build recipe) fuses in StatTile<Float> and Film::UpdateImage; False (default): no FMA."""
    lib().oracle_set_fp_contract(int(bool(on)))


def box_cox(v, lam=0.5):
    return lib().oracle_box_cox(float(v), float(lam))


def set_t_quantiles(alpha_index, quantiles):
    """quantiles: float32 array for dof 1..len, or None to restore the built-in table.  alpha_index is the table
    slot: 0..2 two-sided, 3..5 one-sided."""
    if quantiles is None:
        lib().oracle_set_t_quantiles(int(alpha_index), None, 0)
    else:
        q = np.ascontiguousarray(quantiles, dtype=np.float32)
        lib().oracle_set_t_quantiles(int(alpha_index), _f(q), len(q))


def t_quantile(alpha_index, dof):
    return lib().oracle_t_quantile(int(alpha_index), int(dof))


TILE_PIXEL_DTYPE = {
    1: np.dtype({"names": ["n", "mean", "m2", "m3", "film_mean", "film_m2"],
                 "formats": ["<u8"] + ["<f4"] * 5, "offsets": [0, 8, 12, 16, 20, 24], "itemsize": 64}),
    3: np.dtype({"names": ["n", "mean", "m2", "m3", "film_mean", "film_m2"],
                 "formats": ["<u8"] + [("<f4", (3,))] * 5, "offsets": [0, 8, 20, 32, 44, 56],
                 "itemsize": 128}),
}


def add_samples_to_pixel(samples, channels, transform, max_moment):
    """Run a 1-D sample sequence through one AoS StatTilePixel; returns the structured pixel."""
    px = np.zeros(1, dtype=TILE_PIXEL_DTYPE[channels])
    samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, channels)
    for s in samples:
        lib().oracle_add_sample(px.ctypes.data, channels, _f(np.ascontiguousarray(s)),
                                int(transform), int(max_moment))
    return px[0]


def new_state(h, w, channels):
    z = lambda: np.zeros((h, w, channels), np.float32)
    return dict(n=np.zeros((h, w), np.int32), mean=z(), m2=z(), m3=z(), film_mean=z(), film_m2=z())


def accumulate(state, samples, transform, max_moment, tile_size=16, threads=0):
    """samples: [S, H, W, C] float32.  Updates `state` (see new_state) in place."""
    S, h, w, c = samples.shape
    samples = np.ascontiguousarray(samples, dtype=np.float32)
    lib().oracle_accumulate_image(w, h, c, int(transform), int(max_moment), S, _f(samples),
                                  _i(state["n"]), _f(state["mean"]), _f(state["m2"]), _f(state["m3"]),
                                  _f(state["film_mean"]), _f(state["film_m2"]), tile_size, threads)
    return state


def to_tile_major(samples, tile_size=16):
    """[S, H, W, C] -> flat [tile][pixel][S][C] (tiles and the pixels inside a tile in row-major order): the order
    StatPathIntegrator::Render produces samples in (statpath.cpp:132,255,294-375)."""
    S, h, w, c = samples.shape
    parts = []
    for y0 in range(0, h, tile_size):
        for x0 in range(0, w, tile_size):
            blk = samples[:, y0:y0 + tile_size, x0:x0 + tile_size]          # [S, th, tw, C]
            parts.append(np.ascontiguousarray(blk.transpose(1, 2, 0, 3)).reshape(-1))
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)


def accumulate_tile_stream(state, samples_tile_major, n_samples, transform, max_moment, tile_size=16, threads=0):
    """The accumulation fed tile-major, pixel-major samples (to_tile_major): the same bits as accumulate()."""
    h, w = state["n"].shape
    c = state["mean"].shape[2] if state["mean"].ndim == 3 else 1
    smp = np.ascontiguousarray(samples_tile_major, dtype=np.float32)
    assert smp.size == n_samples * h * w * c
    lib().oracle_accumulate_tile_stream(w, h, c, int(transform), int(max_moment), int(n_samples), _f(smp),
                                        _i(state["n"]), _f(state["mean"]), _f(state["m2"]), _f(state["m3"]),
                                        _f(state["film_mean"]), _f(state["film_m2"]), tile_size, threads)
    return state


def merge_tile(tile_pixels, channels, x0, y0, x1, y1, state, transform=True):
    h, w = state["n"].shape
    fm = _f(state["film_mean"]) if transform else None
    f2 = _f(state["film_m2"]) if transform else None
    lib().oracle_merge_tile(tile_pixels.ctypes.data, channels, x0, y0, x1, y1, w, _i(state["n"]),
                            _f(state["mean"]), _f(state["m2"]), _f(state["m3"]), fm, f2)


def mean_vars(n, film_m2, row_n_quirk=True):
    h, w = n.shape
    c = film_m2.shape[2] if film_m2.ndim == 3 else 1
    out = np.empty_like(film_m2)
    lib().oracle_mean_vars(w, h, c, _i(n), _f(film_m2), _f(out), int(row_n_quirk))
    return out


FILM_PIXEL_DTYPE = np.dtype({"names": ["xyz", "filter_weight_sum", "splat_xyz", "pad"],
                             "formats": [("<f4", (3,)), "<f4", ("<f4", (3,)), "<f4"], "offsets": [0, 12, 16, 28],
                             "itemsize": 32})


def film_update(pixels, splat_scale=1.0, scale=1.0):
    """pixels: structured array of FILM_PIXEL_DTYPE, any shape; returns float32 [..., 3]."""
    assert pixels.dtype == FILM_PIXEL_DTYPE and pixels.flags["C_CONTIGUOUS"]
    out = np.empty(pixels.shape + (3,), np.float32)
    lib().oracle_film_update(pixels.ctypes.data, pixels.size, float(splat_scale), float(scale), _f(out))
    return out


def tile_moments(values, tile_size):
    """{count, mean, M2} per tile and channel: float32 [tiles_y, tiles_x, C, 3]."""
    values = np.ascontiguousarray(values, dtype=np.float32)
    h, w = values.shape[:2]
    c = values.shape[2] if values.ndim == 3 else 1
    out = np.zeros((-(-h // tile_size), -(-w // tile_size), c, 3), np.float32)
    lib().oracle_tile_moments(w, h, c, _f(values), int(tile_size), _f(out))
    return out


def default_spec():
    """The pinned spec (include/statmc_pinned_spec.h; all zero = spec v2 until tools/pin_from_dumps.sh has run)."""
    s = FilterSpec()
    lib().oracle_default_spec(C.byref(s))
    return s


def default_significance():
    return int(lib().oracle_default_significance())


def prepass(n, mean, m2, m3, alpha_index=None, spec=None):
    h, w = n.shape
    c = mean.shape[2] if mean.ndim == 3 else 1
    mc, disc = np.empty_like(mean), np.empty_like(mean)
    spec = spec if spec is not None else default_spec()
    alpha_index = default_significance() if alpha_index is None else alpha_index
    lib().oracle_prepass_spec(w, h, c, alpha_index, C.byref(spec), _i(n), _f(mean), _f(m2), _f(m3), _f(mc), _f(disc))
    return mc, disc


def filter_image(mean_corr, disc, colour, g_buffers, g_dr, ds, radius, roi=None, threads=0, spec=None, n=None,
                 alpha_index=None):
    """g_buffers: list of [H, W, Cg] (or [H, W]) float32 arrays; g_dr: list of -0.5/sd^2.
    spec: FilterSpec (default: the pinned spec); n: the int32 sample counts, needed in Welch mode only."""
    h, w = mean_corr.shape[:2]
    c = mean_corr.shape[2] if mean_corr.ndim == 3 else 1
    gs = [np.ascontiguousarray(g, dtype=np.float32) for g in g_buffers]
    ng = len(gs)
    gptrs = (C.POINTER(C.c_float) * max(ng, 1))(*[_f(g) for g in gs])
    gch = (C.c_int * max(ng, 1))(*[(g.shape[2] if g.ndim == 3 else 1) for g in gs])
    gdr = np.asarray(list(g_dr) + ([] if ng else [0.0]), dtype=np.float32)
    out = np.zeros_like(colour)
    x0, y0, x1, y1 = roi if roi is not None else (0, 0, w, h)
    spec = spec if spec is not None else default_spec()
    alpha_index = default_significance() if alpha_index is None else alpha_index
    assert spec.dof == DOF_PIXEL or n is not None, "Welch mode reads the sample counts"
    lib().oracle_filter_spec_run(w, h, c, float(ds), int(radius), int(alpha_index), C.byref(spec),
                                 _i(n) if n is not None else None, _f(mean_corr), _f(disc), _f(colour),
                                 ng, gptrs, gch, _f(gdr), _f(out), x0, y0, x1, y1, threads)
    return out
