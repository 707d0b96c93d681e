"""One rank's share of the hot path: accumulate -> pre-pass -> [halo exchange] -> window filter on
its block of the film.  Used by bench.py and by the multi-rank tests; torch supplies device
memory and torch.distributed (RCCL) the halo exchange, every number comes out of libstatmc_hip.so.
"""
import torch

from . import api, film, sharding



class BlockPipeline:
    def __init__(self, layout, device, types, filter_sd=10.0, radius=20, via_host=False, reproducible=False,
                 g_buffers=("normal", "albedo"), g_sds=None, placed=False, fused_prepass=False):
        """placed: the running moments come from statmc_malloc_placed(STATMC_MEM_STATE) (include/statmc.h; the caller
        allocates its sample arenas with api.empty_placed(..., api.MEM_STREAM)).
        reproducible: pin the window-sweep split of this device to the uniform split a single device would choose for the
        WHOLE film (statmc_filter_split_auto -> statmc_set_filter_split), so that the assembled blocks equal the one-device
        result bit for bit -- PROVIDED the one-device run pins the same split: its automatic choice may add a tail split for
        tile counts that leave the last round of workgroups mostly empty (1280 x 720: 900 tiles on 256 CUs), which
        statmc_filter_split_auto does not report (include/statmc.h, "window-sweep split").  Default: the split fitted to the
        block (faster on strips, <= 1e-6 from the one-device result)."""
        self.layout, self.device, self.via_host = layout, device, via_host
        # fused_prepass (one block = the whole film only): the accumulation's epilogue writes mean-corr / discriminator, prepass()
        # has nothing left to launch (film.FilmStats); blocks of a sharded film pre-pass and pack in one pass of their own
        self.fs = film.FilmStats(layout.bw, layout.bh, device, types=types, filter_sd=filter_sd, radius=radius,
                                 g_buffers=g_buffers, g_sds=g_sds, placed=placed, fused_prepass=fused_prepass and layout.world == 1)
        self.g_channels = [film.STAT_TYPES[g]["channels"] for g in self.fs.g_names]
        if reproducible:
            fw, fh = layout.film_size
            api.set_filter_split(api.filter_split_auto(fw, fh, radius))
        self.filter_sd, self.radius = filter_sd, radius
        self.multi = layout.world > 1
        self.side = None   # stream of the interior's accumulation in the overlapped order
        if self.multi:
            # block + halo: one 15-channel image (mean-corr, discriminator, colour, normal, albedo per
            # pixel) is what the pack kernel writes, the halo exchange moves and the filter reads; with 1-channel
            # G-buffers (depth, material id) it has 17 channels, under Welch degrees of freedom (the device's filter spec
            # at this moment) 16 / 18: + the sample count
            welch = api.get_filter_spec().dof == api.DOF_WELCH
            self.packed = layout.new_padded(sharding.block_image_channels(self.g_channels, welch), device)
            self.out_pad = layout.new_padded(3, device)

    def accumulate(self, samples, rows=None):
        self.fs.accumulate(samples, rows=rows)

    def prepass(self, rows=None):
        """Single GPU: the pre-pass.  Multi-GPU: pre-pass and pack of the owned block (or of its rows [y0, y1)) in one
        HIP pass -- the five filter inputs go straight into the block + halo image, mean-corr / discriminator are
        still written out as the reference's device images."""
        if not self.multi:
            self.fs.prepass()
            return
        L = self.layout
        args, keep = self.fs.filter_args()
        if rows is not None and not hasattr(rows[0], "__len__"):
            rows = [rows]                                  # one (y0, y1) range; a list holds one or two of them: one launch
        api.prepass_pack(args, self.packed, L.pl, L.pt, rows=rows)

    def exchange(self):
        """Fetch the r-pixel border of the block + halo image from the neighbours (RCCL send/recv)."""
        sharding.exchange_halo(self.layout, self.packed, via_host=self.via_host)

    def exchange_rccl(self, comm):
        """The same exchange through the C ABI's RCCL entry point (statmc_halo_exchange_rccl; comm: api.RcclComm): what a
        multi-process C++ host calls.  Asynchronous on the current stream."""
        L = self.layout
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        api.halo_exchange_rccl(self.packed, idx, L.gx, L.gy, L.bw, L.bh, self.radius, comm)

    # ---- the halo exchange behind the accumulation (row-strip grids): the rows a neighbour needs are accumulated,
    # pre-passed and sent first; the rest of the block is accumulated while they travel.  Per-pixel stages, so the bits are
    # those of the one-piece order.
    def border_rows(self):
        """Row ranges of the owned block that go to a neighbour: [(0, r)] and / or [(bh - r, bh)]."""
        return self.layout.border_rows()

    def interior_rows(self):
        return self.layout.interior_rows()

    def accumulate_and_denoise(self, samples, overlap=True):
        """One iteration on this rank's block: accumulate -> pre-pass -> halo exchange -> window filter.  With `overlap`
        (row-strip grids) the rows a neighbour needs go first and travel while the rest is accumulated."""
        border = self.border_rows() if overlap else []
        if not border:
            self.accumulate(samples)
            return self.denoise()
        in_flight = self.border_first(samples)
        self.interior_beside(samples)
        self.join_interior(in_flight)
        return self.window_filter()

    # The border chain -- both strips accumulated in one launch (a few hundred waves, bound by their own latency: 0.13 ms
    # for 30 % of an N = 8 block), pre-passed and packed, the sends and receives issued -- goes first on the current
    # stream.  The rest of the block is accumulated on a side stream that waits only for what preceded the border chain, so
    # it starts a few microseconds behind the border launch, takes the part of the machine that launch leaves free and all
    # of it when the strips are done.  (The other way round -- interior on the main stream, border chain on a high-priority
    # side stream, forked from the same event or not -- lets the interior's workgroups occupy every slot first: stream
    # priority does not reorder the dispatch, the strips finish WITH the interior and nothing is hidden; rocprofv3
    # timelines, tools/experiments/block_step_trace.py.  N = 8, middle rank, per step without the exchange E: one piece
    # 0.78 ms + E; strips first, then the rest 0.87 + max(0, E - 0.37); this order 0.83 + max(0, E - 0.10).)
    def border_first(self, samples_or_fn, exchange=True):
        """samples_or_fn: the samples of this batch, or a callable(rows) that accumulates them (bench.py's pooled slices).
        exchange=False leaves the sends and receives out (one-process timing scripts that stand in for one rank)."""
        main = torch.cuda.current_stream(self.device)
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.device)
            self.fork = torch.cuda.Event()
        self.fork.record(main)                          # what both chains have to wait for: the previous iteration
        border = self.border_rows()

        if callable(samples_or_fn):
            samples_or_fn(border)
        else:
            self.accumulate(samples_or_fn, rows=border)
        self.prepass(rows=border)                       # both strips in one launch, too
        return self.exchange_start() if exchange else sharding.HaloInFlight()

    def interior_beside(self, samples_or_fn, timed=None):
        """Accumulation and pre-pass of the rows that need no neighbour, on the side stream.  timed(name, fn, *args): the
        caller's per-kernel event timing (bench.py), applied on the side stream."""
        self.side.wait_event(self.fork)
        rows = self.interior_rows()
        run = timed if timed is not None else (lambda name, fn, *a: fn(*a))
        with torch.cuda.stream(self.side):
            if callable(samples_or_fn):
                run("accumulate", samples_or_fn, rows)
            else:
                run("accumulate", self.accumulate, samples_or_fn, rows)
            run("prepass", self.prepass, rows)

    def join_side(self):
        """Orders the current stream behind the interior's chain (side stream)."""
        torch.cuda.current_stream(self.device).wait_stream(self.side)

    def join_interior(self, in_flight):
        """Orders the current stream behind the interior's chain and the receives."""
        self.join_side()
        in_flight.wait()

    def exchange_start(self):
        return sharding.exchange_halo_start(self.layout, self.packed, via_host=self.via_host)

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
            packed=self.packed, film_origin=(ox - L.pl, oy - L.pt), packed_g_channels=self.g_channels)
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
