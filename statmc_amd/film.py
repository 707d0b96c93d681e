"""Device-side film statistics + the accumulate -> pre-pass -> window-filter pass, expressed
with the reference's buffer names (t<type>-b<bounce>-{n,mean,m2,m3,film-mean,film-m2,mean-corr,
discriminator,film-mean-f}, "film", "film-f"; estimator.cpp:101-161).  Holds torch tensors only
as HBM allocations; all arithmetic happens in libstatmc_hip.so.
"""
import torch

from . import api

# (channels, transform, max_moment, filter sd) of the shipped denoise configuration
# (statpath.cpp:1027-1160; scenes/render-denoise.pbrt:19-22) plus the two float G-buffers (their sds: this build's
# choice for the synthetic scene -- the reference takes them from the scene file's `filterbuffersds`, no default).
STAT_TYPES = {
    "radiance":   dict(channels=3, transform=True,  max_moment=3, sd=None),
    "normal":     dict(channels=3, transform=False, max_moment=1, sd=0.1),
    "albedo":     dict(channels=3, transform=False, max_moment=1, sd=0.02),
    "depth":      dict(channels=1, transform=False, max_moment=1, sd=1.0),
    "materialid": dict(channels=1, transform=False, max_moment=1, sd=0.5),
}


def new_state(height, width, channels, device, transform=True, placed=False):
    """placed: the images come from statmc_malloc_placed(STATMC_MEM_STATE) -- one HBM rank, apart from the sample arenas
    (api.empty_placed(..., api.MEM_STREAM)): include/statmc.h, "Device memory placed by HBM rank"."""
    if placed:
        z = lambda: api.zeros_placed((height, width, channels), torch.float32, device, api.MEM_STATE)
        n = api.zeros_placed((height, width), torch.int32, device, api.MEM_STATE)
    else:
        z = lambda: torch.zeros(height, width, channels, dtype=torch.float32, device=device)
        n = torch.zeros(height, width, dtype=torch.int32, device=device)
    st = dict(n=n, mean=z(), m2=z(), m3=z())
    if transform:
        st["film_mean"], st["film_m2"] = z(), z()
    else:  # non-transform types alias mean/m2 (estimator.cpp:127-137)
        st["film_mean"], st["film_m2"] = None, None
    return st


class FilmStats:
    """One GPU's block of the film: per-type running moments and the filter's work images."""

    def __init__(self, width, height, device, types=("radiance", "normal", "albedo"),
                 filter_sd=10.0, radius=20, g_buffers=("normal", "albedo"), g_sds=None, placed=False, fused_prepass=False):
        """fused_prepass: whole-film accumulations also write the radiance type's mean-corr / discriminator in their epilogue
        (statmc_stat_type::mean_corr / discriminator: the bits of statmc_prepass), and prepass() has nothing left to do while
        that result is current -- same filter spec and significance level, no row-range accumulation, no reset since."""
        api.setup(device.index if device.index is not None else 0)
        self.width, self.height, self.device = width, height, device
        self.types = list(types)
        self.placed = placed
        self.state = {t: new_state(height, width, STAT_TYPES[t]["channels"], device,
                                   STAT_TYPES[t]["transform"], placed=placed) for t in self.types}
        self.filter_sd, self.radius = filter_sd, radius
        self.g_names = list(g_buffers)
        self.g_sds = list(g_sds) if g_sds is not None else [STAT_TYPES[g]["sd"] for g in self.g_names]
        # the filter's work images are written by one kernel and read by the next: with the moments (placed: state role)
        z3 = (lambda: api.zeros_placed((height, width, 3), torch.float32, device, api.MEM_STATE)) if placed else \
             (lambda: torch.zeros(height, width, 3, dtype=torch.float32, device=device))
        self.mean_corr, self.disc, self.film, self.film_f = z3(), z3(), z3(), z3()
        self.fused_prepass = bool(fused_prepass)
        self._prepass_current = None      # (filter spec, significance) under which the epilogue's result was written

    def reset(self):
        self._prepass_current = None
        for st in self.state.values():
            for v in st.values():
                if v is not None:
                    v.zero_()

    def accumulate(self, samples, rows=None):
        """samples: {type: [S, H, W, C]}; one kernel launch covers every stat type.  rows = (y0, y1): only those rows."""
        fuse = self.fused_prepass and rows is None and "radiance" in samples
        sts = [api.make_stat_type(samples[t], self.state[t], STAT_TYPES[t]["transform"], STAT_TYPES[t]["max_moment"],
                                  prepass_into=(self.mean_corr, self.disc) if (fuse and t == "radiance") else None)
               for t in self.types if t in samples]
        api.accumulate(self.width, self.height, sts, rows=rows)
        self._prepass_current = self._prepass_key() if fuse else None

    def _prepass_key(self):
        return (api.get_filter_spec().as_tuple(), api.get_significance())

    def g_buffer(self, name):
        return self.state[name]["mean"]  # film-mean == mean for non-transform types

    def filter_args(self, roi=None, colour=None, out=None, rows=None):
        """rows = (y0, y1): the call sees those rows of every image as images of their own (per-pixel stages only)."""
        rad = self.state["radiance"]
        colour = colour if colour is not None else rad["film_mean"]
        out = out if out is not None else self.film_f
        cut = (lambda t: t) if rows is None else (lambda t: t[rows[0]:rows[1]])
        args, keep = api.make_filter_args(
            n=[cut(rad["n"])], mean=[cut(rad["mean"])], m2=[cut(rad["m2"])], m3=[cut(rad["m3"])], film=[cut(colour)],
            mean_corr=[cut(self.mean_corr)], disc=[cut(self.disc)], film_filtered=[cut(out)],
            g_buffers=[cut(self.g_buffer(g)) for g in self.g_names], g_sds=self.g_sds,
            filter_sd=self.filter_sd, radius=self.radius, denoise_film=False, roi=roi)
        return args, keep

    def prepass(self):
        if self._prepass_current is not None and self._prepass_current == self._prepass_key():
            return          # the last accumulation's epilogue wrote mean-corr / discriminator already (same bits)
        self._prepass_current = None
        args, keep = self.filter_args()
        api.prepass(args, 3)

    def window_filter(self, roi=None):
        args, keep = self.filter_args(roi=roi)
        api.window_filter(args, 3)

    def denoise(self):
        """Estimator::Denoise for the RGB radiance buffer: pre-pass + window filter."""
        args, keep = self.filter_args()
        api.filter_f32x3(args)
        return self.film_f
