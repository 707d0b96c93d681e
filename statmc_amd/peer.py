"""ONE process, every device: the blocks of a film driven through the C ABI alone -- per block
statmc_accumulate_row_ranges -> statmc_prepass_pack_rows -> statmc_halo_exchange (device-to-device copies, peer access
over xGMI) -> statmc_window_filter -- in the same border-first order as the one-process-per-GPU path (pipeline.py),
without torch.distributed and without RCCL.  This is the second way for an N > 1 run to exist (bench.py --backend peer,
and the fallback of the nccl leg) and the Python twin of statmc::FilmShards (include/statmc_denoiser.hpp), extended from
the filter to the whole step.

Host cost matters here: one thread enqueues for all devices, so every call of a step is prepared once (ctypes argument
blocks, row-range tables, stream handles) and a step is a flat walk over those calls.  torch supplies device memory,
streams and timing events; every number comes out of libstatmc_hip.so.
"""
import ctypes as C

import torch

from . import api, film, sharding


class Block(C.Structure):     # statmc_block (include/statmc.h)
    _fields_ = [("device", C.c_int32), ("packed", api.Image), ("stream", C.c_void_p)]


def _ranges(rows):
    flat = (C.c_int32 * (2 * len(rows)))(*[int(v) for r in rows for v in r])
    return flat, len(rows)


class _Call:
    """One prepared C call: fn(*args) on `device`."""
    __slots__ = ("device", "fn", "args", "keep")

    def __init__(self, device, fn, args, keep=None):
        self.device, self.fn, self.args, self.keep = device, fn, args, keep


class PeerBlock:
    """Block b of the grid, resident on its device: statistics, block + halo image, two streams."""

    def __init__(self, layout, device_index, types, filter_sd, radius, g_buffers=("normal", "albedo"), g_sds=None, placed=False):
        self.layout, self.device_index = layout, int(device_index)
        self.dev = torch.device("cuda", self.device_index)
        with torch.cuda.device(self.dev):      # (placed: statmc_malloc_placed acts on the current device)
            self.fs = film.FilmStats(layout.bw, layout.bh, self.dev, types=types, filter_sd=filter_sd, radius=radius,
                                     g_buffers=g_buffers, g_sds=g_sds, placed=placed)
        self.g_channels = [film.STAT_TYPES[g]["channels"] for g in self.fs.g_names]
        api.check(api.load().statmc_set_device(self.device_index))
        # (laid out for the device's filter spec at this moment: 16 / 18 channels under Welch degrees of freedom)
        self.channels = sharding.block_image_channels(self.g_channels, api.get_filter_spec().dof == api.DOF_WELCH)
        self.packed = layout.new_padded(self.channels, self.dev)
        self.out_pad = layout.new_padded(3, self.dev)
        self.main = torch.cuda.Stream(device=self.dev)
        self.side = torch.cuda.Stream(device=self.dev)
        self.h_main, self.h_side = C.c_void_p(self.main.cuda_stream), C.c_void_p(self.side.cuda_stream)
        lib = api.load()
        self.ev = {}
        for name in ("fork", "join", "halo_done"):
            e = C.c_void_p()
            api.check(lib.statmc_set_device(self.device_index))
            api.check(lib.statmc_event_create(C.byref(e)))
            self.ev[name] = e

    def border_rows(self):
        return self.layout.border_rows()

    def interior_rows(self):
        return self.layout.interior_rows()


class PeerFilm:
    """The blocks of one film on the devices of one process.  devices[b] = HIP device of block b (the same index
    several times: blocks share a device -- what a 1-GPU box can rehearse).

    step(batches): batches = [{type: [S, bh, bw, C] tensor} per block] or a list of such (several accumulate launches
    per step: pooled sample slices, the reference's progressive schedule is the caller's loop)."""

    def __init__(self, world, block_w, block_h, radius, devices, types, filter_sd=10.0, grid=None, overlap=True,
                 g_buffers=("normal", "albedo"), g_sds=None, placed=False):
        """placed: every block's running moments come from statmc_malloc_placed(STATMC_MEM_STATE) on its device."""
        self.lib = api.load()
        self.world, self.radius, self.filter_sd, self.types = world, radius, filter_sd, list(types)
        assert len(devices) == world
        for d in sorted(set(devices)):
            api.setup(d)
        if len(set(devices)) > 1:      # every block is filtered under the rules of block 0's device
            for d in sorted(set(devices)):
                if d != devices[0]:
                    api.check(self.lib.statmc_copy_device_settings(devices[0], d))
        self.blocks = [PeerBlock(sharding.BlockLayout(b, world, block_w, block_h, radius, grid=grid), devices[b], types,
                                 filter_sd, radius, g_buffers=g_buffers, g_sds=g_sds, placed=placed) for b in range(world)]
        L0 = self.blocks[0].layout
        self.gx, self.gy, self.bw, self.bh = L0.gx, L0.gy, block_w, block_h
        self.overlap = bool(overlap) and all(len(b.border_rows()) > 0 for b in self.blocks) and world > 1
        self.film_size = L0.film_size
        self._blocks_c = (Block * world)()
        for b, blk in enumerate(self.blocks):
            self._blocks_c[b].device = blk.device_index
            self._blocks_c[b].packed = api.image_of(blk.packed)
            self._blocks_c[b].stream = blk.h_main
        self._filter_calls = [self._prepare_filter(blk) for blk in self.blocks]
        for d in sorted(set(devices)):      # the allocations above were zeroed on the devices' default streams
            torch.cuda.synchronize(torch.device("cuda", d))

    # ---- prepared calls
    def _prepare_filter(self, blk):
        L = blk.layout
        ox, oy = L.origin
        g_sds = blk.fs.g_sds
        a, keep = api.make_filter_args(
            n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[blk.out_pad],
            g_buffers=[], g_sds=g_sds, filter_sd=self.filter_sd, radius=self.radius, roi=L.roi,
            packed=blk.packed, film_origin=(ox - L.pl, oy - L.pt), stream=blk.h_main, packed_g_channels=blk.g_channels)
        return _Call(blk.device_index, self.lib.statmc_window_filter, (C.byref(a), 3), (a, keep))

    def _prepare_accumulate(self, blk, samples, rows, stream):
        sts = [api.make_stat_type(samples[t], blk.fs.state[t], film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"])
               for t in blk.fs.types if t in samples]
        arr = (api.StatType * max(len(sts), 1))(*sts)
        flat, n = _ranges(rows)
        return _Call(blk.device_index, self.lib.statmc_accumulate_row_ranges,
                     (blk.layout.bw, blk.layout.bh, arr, len(sts), flat, n, stream), (arr, flat, samples))

    def _prepare_pack(self, blk, rows, stream):
        L = blk.layout
        a, keep = blk.fs.filter_args()
        a.stream = stream
        img = api.image_of(blk.packed)
        if rows is None:
            return _Call(blk.device_index, self.lib.statmc_prepass_pack, (C.byref(a), C.byref(img), L.pl, L.pt), (a, keep, img))
        flat, n = _ranges(rows)
        return _Call(blk.device_index, self.lib.statmc_prepass_pack_rows, (C.byref(a), C.byref(img), L.pl, L.pt, flat, n), (a, keep, img, flat))

    def prepare_step(self, per_block_batches, overlap=None):
        """per_block_batches[b] = list of sample dicts (one accumulate launch each, in order).  overlap: None = this film's
        default order; False = the plain order (whole block, then the exchange).  Returns a plan for run(); everything the
        step needs is marshalled here, once."""
        overlap = self.overlap if overlap is None else (bool(overlap) and self.overlap)
        plan = {"overlap": overlap, "acc_main": [], "pack_main": [], "acc_side": [], "pack_side": [], "filter": self._filter_calls}
        for blk, batches in zip(self.blocks, per_block_batches):
            if overlap:
                border, interior = blk.border_rows(), [blk.interior_rows()]
                plan["acc_main"].append([self._prepare_accumulate(blk, s, border, blk.h_main) for s in batches])
                plan["pack_main"].append(self._prepare_pack(blk, border, blk.h_main))
                plan["acc_side"].append([self._prepare_accumulate(blk, s, interior, blk.h_side) for s in batches])
                plan["pack_side"].append(self._prepare_pack(blk, interior, blk.h_side))
            else:
                whole = [(0, blk.layout.bh)]
                plan["acc_main"].append([self._prepare_accumulate(blk, s, whole, blk.h_main) for s in batches])
                plan["pack_main"].append(self._prepare_pack(blk, None, blk.h_main))
        return plan

    # ---- one step
    def run(self, plan, timer=None):
        """Enqueues one step on every block's streams and returns (no synchronisation).  timer(block_index, name, stream)
        is called at the marks of a block's step -- "start", "accumulated", "border" (= packed), "exchange", "joined",
        "filter" on the main stream; "interior_start", "interior_accumulated", "interior" on the side stream -- bench.py
        records timing events there."""
        lib, blocks = self.lib, self.blocks
        set_device, record, wait = lib.statmc_set_device, lib.statmc_event_record, lib.statmc_stream_wait_event
        mark = timer if timer is not None else (lambda b, name, stream: None)
        overlap = plan["overlap"]
        # the block + halo image of a block is about to be rewritten: its neighbours' copies OUT of it (previous step, on
        # THEIR streams) must have passed
        for b, blk in enumerate(blocks):
            set_device(blk.device_index)
            L = blk.layout
            for nb in (L.left, L.right, L.up, L.down):
                if nb is not None:
                    wait(blk.h_main, blocks[nb].ev["halo_done"])
            mark(b, "start", blk.main)
            if overlap:
                record(blk.ev["fork"], blk.h_main)
            for c in plan["acc_main"][b]:
                api.check(c.fn(*c.args))
            mark(b, "accumulated", blk.main)
            c = plan["pack_main"][b]
            api.check(c.fn(*c.args))
            mark(b, "border", blk.main)
            if overlap:     # the rest of the block right behind it, on the side stream (it waits for the fork only)
                wait(blk.h_side, blk.ev["fork"])
                mark(b, "interior_start", blk.side)
                for c in plan["acc_side"][b]:
                    api.check(c.fn(*c.args))
                mark(b, "interior_accumulated", blk.side)
                c = plan["pack_side"][b]
                api.check(c.fn(*c.args))
                mark(b, "interior", blk.side)
                record(blk.ev["join"], blk.h_side)
        # copies on the destination blocks' main streams, each behind its source block's pack (events inside)
        api.check(lib.statmc_halo_exchange(self._blocks_c, self.gx, self.gy, self.bw, self.bh, self.radius))
        for b, blk in enumerate(blocks):
            set_device(blk.device_index)
            record(blk.ev["halo_done"], blk.h_main)
            mark(b, "exchange", blk.main)
            if overlap:
                wait(blk.h_main, blk.ev["join"])
                mark(b, "joined", blk.main)
            c = plan["filter"][b]
            api.check(c.fn(*c.args))
            mark(b, "filter", blk.main)

    def synchronize(self):
        for blk in self.blocks:
            self.lib.statmc_set_device(blk.device_index)
            api.check(self.lib.statmc_synchronize(blk.h_main))
            api.check(self.lib.statmc_synchronize(blk.h_side))

    def reset(self):
        for blk in self.blocks:
            with torch.cuda.device(blk.dev), torch.cuda.stream(blk.main):
                blk.fs.reset()

    def filtered_block(self, b):
        blk = self.blocks[b]
        return blk.layout.interior(blk.out_pad)

    def gather(self, film_f=None, dst_device=None):
        """SURVEY 8e's final gather: every block's filtered pixels -> one [fh, fw, 3] image on `dst_device` (default: block
        0's), by statmc_copy_rect on the source blocks' main streams; returns it (not synchronised)."""
        fw, fh = self.film_size
        d = self.blocks[0].device_index if dst_device is None else int(dst_device)
        if film_f is None:
            film_f = torch.empty(fh, fw, 3, dtype=torch.float32, device=torch.device("cuda", d))
        dst = api.image_of(film_f)
        for blk in self.blocks:
            L = blk.layout
            ox, oy = L.origin
            src = api.image_of(blk.out_pad)
            self.lib.statmc_set_device(blk.device_index)
            api.check(self.lib.statmc_copy_rect(C.byref(dst), d, ox, oy, C.byref(src), blk.device_index, L.pl, L.pt, L.bw, L.bh, 12, blk.h_main))
        return film_f
