"""One rank's share of the hot path: accumulate -> pre-pass -> [halo exchange] -> window filter on
its block of the film.  Used by bench.py and by the multi-rank tests; torch supplies device
memory and torch.distributed (RCCL) the halo exchange, every number comes out of libstatmc_hip.so.
"""
import torch

from . import api, film, sharding



class BlockPipeline:
    def __init__(self, layout, device, types, filter_sd=10.0, radius=20, via_host=False):
        self.layout, self.device, self.via_host = layout, device, via_host
        self.fs = film.FilmStats(layout.bw, layout.bh, device, types=types, filter_sd=filter_sd, radius=radius)
        self.filter_sd, self.radius = filter_sd, radius
        self.multi = layout.world > 1
        if self.multi:
            # block + halo: one 15-channel image (mean-corr, discriminator, colour, normal, albedo per
            # pixel) is what the pack kernel writes, the halo exchange moves and the filter reads
            self.packed = layout.new_padded(15, device)
            self.out_pad = layout.new_padded(3, device)

    def accumulate(self, samples):
        self.fs.accumulate(samples)

    def prepass(self):
        """Single GPU: the pre-pass.  Multi-GPU: pre-pass and pack of the owned block in one HIP pass --
        the five filter inputs go straight into the block + halo image, mean-corr / discriminator are
        still written out as the reference's device images."""
        if not self.multi:
            self.fs.prepass()
            return
        L = self.layout
        args, keep = self.fs.filter_args()
        api.prepass_pack(args, self.packed, L.pl, L.pt)

    def exchange(self):
        """Fetch the r-pixel border of the block + halo image from the neighbours (RCCL send/recv)."""
        sharding.exchange_halo(self.layout, self.packed, via_host=self.via_host)

    def window_filter(self):
        """Returns the filtered owned block ([bh, bw, 3] view)."""
        if not self.multi:
            self.fs.window_filter()
            return self.fs.film_f
        L = self.layout
        ox, oy = L.origin
        # film_origin: the filter's work tiles sit on a grid fixed in film coordinates, so every pixel's sums are
        # formed in the same order as in a whole-film run (bit-identical results, tests/test_multirank_gpu.py)
        a, keep = api.make_filter_args(
            n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[self.out_pad],
            g_buffers=[], g_sds=self.fs.g_sds, filter_sd=self.filter_sd, radius=self.radius, roi=L.roi,
            packed=self.packed, film_origin=(ox - L.pl, oy - L.pt))
        api.window_filter(a, 3)
        return L.interior(self.out_pad)

    def denoise(self):
        self.prepass()
        if self.multi:
            self.exchange()
        return self.window_filter()

    def gather_film(self, block, film_f=None, dst=0):
        """The "final gather" of SURVEY 8e: every rank's filtered block -> the whole film-f image on rank `dst`."""
        return sharding.gather_blocks(self.layout, block, film_f, dst=dst, via_host=self.via_host)
