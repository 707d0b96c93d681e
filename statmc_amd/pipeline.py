"""One rank's share of the hot path: accumulate -> pre-pass -> [halo exchange] -> window filter on
its block of the film.  Used by bench.py and by the multi-rank tests; torch supplies device
memory and torch.distributed (RCCL) the halo exchange, every number comes out of libstatmc_hip.so.
"""
import torch

from . import api, film, sharding

HALO_IMAGES = ("mean_corr", "disc", "colour", "normal", "albedo")


class BlockPipeline:
    def __init__(self, layout, device, types, filter_sd=10.0, radius=20, via_host=False):
        self.layout, self.device, self.via_host = layout, device, via_host
        self.fs = film.FilmStats(layout.bw, layout.bh, device, types=types, filter_sd=filter_sd, radius=radius)
        self.filter_sd, self.radius = filter_sd, radius
        self.multi = layout.world > 1
        if self.multi:
            # block + halo: one 15-channel exchange buffer, five padded float3 filter inputs, padded output
            self.packed = layout.new_padded(15, device)
            self.pad = {k: layout.new_padded(3, device) for k in HALO_IMAGES}
            self.out_pad = layout.new_padded(3, device)

    def accumulate(self, samples):
        self.fs.accumulate(samples)

    def prepass(self):
        self.fs.prepass()

    def exchange(self):
        """Pack the five filter inputs of the owned block, fetch the r-pixel border from the
        neighbours (two-phase RCCL send/recv), unpack into the padded images."""
        fs, L = self.fs, self.layout
        rad = fs.state["radiance"]
        inner = L.interior(self.packed)
        for i, src in enumerate((fs.mean_corr, fs.disc, rad["film_mean"], fs.g_buffer("normal"), fs.g_buffer("albedo"))):
            inner[..., 3 * i:3 * i + 3].copy_(src)
        sharding.exchange_halo(L, self.packed, via_host=self.via_host)
        for i, k in enumerate(HALO_IMAGES):
            self.pad[k].copy_(self.packed[..., 3 * i:3 * i + 3])

    def window_filter(self):
        """Returns the filtered owned block ([bh, bw, 3] view)."""
        if not self.multi:
            self.fs.window_filter()
            return self.fs.film_f
        L = self.layout
        a, keep = api.make_filter_args(
            n=[], mean=[], m2=[], m3=[], film=[self.pad["colour"]], mean_corr=[self.pad["mean_corr"]],
            disc=[self.pad["disc"]], film_filtered=[self.out_pad], g_buffers=[self.pad["normal"], self.pad["albedo"]],
            g_sds=self.fs.g_sds, filter_sd=self.filter_sd, radius=self.radius, roi=L.roi)
        api.window_filter(a, 3)
        return L.interior(self.out_pad)

    def denoise(self):
        self.prepass()
        if self.multi:
            self.exchange()
        return self.window_filter()
