"""Block decomposition of the film over the GPUs of one node and the halo exchange the window
filter needs (new capability: the reference is single-GPU).

Every rank owns one block of every per-pixel image.  Accumulation and the pre-pass are purely
per-pixel, so they run on the owned block with no communication.  The window filter reads a
(2r+1)^2 neighbourhood, so before it runs each rank fetches an r-pixel border of the five filter
inputs (mean-corr, discriminator, colour, normal, albedo = 15 floats / pixel) from its
neighbours: a two-phase exchange (left/right columns first, then top/bottom rows of the already
widened block, which carries the corners along) = at most 4 point-to-point messages per rank
over RCCL send/recv (xGMI is point-to-point, every neighbour is one hop).
"""
import torch
import torch.distributed as dist

GRIDS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}  # 2-D blocks


def row_strips(world_size):
    """1 x N grid: every block spans the film width, so a rank has at most two neighbours and its
    halo messages are whole contiguous rows (no staging copies, one exchange phase)."""
    return (1, world_size)


def grid_for(world_size):
    if world_size in GRIDS:
        return GRIDS[world_size]
    return (world_size, 1)  # column strips for any other count


class BlockLayout:
    """Rank (bx, by) of a gx x gy grid of equal blocks (block_w x block_h pixels each)."""

    def __init__(self, rank, world_size, block_w, block_h, radius, grid=None):
        self.rank, self.world = rank, world_size
        self.gx, self.gy = grid if grid is not None else grid_for(world_size)
        assert self.gx * self.gy == world_size
        self.bx, self.by = rank % self.gx, rank // self.gx
        self.bw, self.bh, self.r = block_w, block_h, radius
        # a halo comes from ONE ring of neighbours: a block narrower than the radius would need the next ring too
        if (self.gx > 1 and block_w < radius) or (self.gy > 1 and block_h < radius):
            raise ValueError("block %dx%d is smaller than the filter radius %d: the halo exchange reaches one "
                             "ring of neighbours only" % (block_w, block_h, radius))
        self.left = self._nb(-1, 0)
        self.right = self._nb(1, 0)
        self.up = self._nb(0, -1)
        self.down = self._nb(0, 1)
        # halo widths actually present (no neighbour -> true film border -> no padding)
        self.pl = radius if self.left is not None else 0
        self.pr = radius if self.right is not None else 0
        self.pt = radius if self.up is not None else 0
        self.pb = radius if self.down is not None else 0
        self.pw, self.ph = self.bw + self.pl + self.pr, self.bh + self.pt + self.pb

    def _nb(self, dx, dy):
        x, y = self.bx + dx, self.by + dy
        if 0 <= x < self.gx and 0 <= y < self.gy:
            return y * self.gx + x
        return None

    @property
    def film_size(self):
        return self.gx * self.bw, self.gy * self.bh

    @property
    def origin(self):
        """Film coordinates of the block's first pixel."""
        return self.bx * self.bw, self.by * self.bh

    @property
    def roi(self):
        """Owned pixels inside the padded local image: (x0, y0, x1, y1)."""
        return (self.pl, self.pt, self.pl + self.bw, self.pt + self.bh)

    # Row-strip grids: the rows a neighbour needs (its halo) are accumulated, pre-passed and sent first, the rest of the
    # block while they travel (pipeline.py, peer.py).  Blocks too short to have an interior keep the plain order.
    def border_rows(self):
        """Row ranges of the owned block that go to a neighbour: [(0, r)] and / or [(bh - r, bh)]; [] = plain order."""
        out = []
        if self.world > 1 and self.gx == 1 and self.bh >= 2 * self.r + 8:
            if self.up is not None:
                out.append((0, self.r))
            if self.down is not None:
                out.append((self.bh - self.r, self.bh))
        return out

    def interior_rows(self):
        b = self.border_rows()
        lo = self.r if any(y0 == 0 for y0, _ in b) else 0
        hi = self.bh - (self.r if any(y1 == self.bh for _, y1 in b) else 0)
        return (lo, hi)

    def new_padded(self, channels, device, dtype=torch.float32):
        return torch.zeros(self.ph, self.pw, channels, dtype=dtype, device=device)

    def interior(self, padded):
        return padded[self.pt:self.pt + self.bh, self.pl:self.pl + self.bw]


def block_image_channels(g_channels, welch=False):
    """Channels per pixel of a block + halo image: 15 (mean-corr, discriminator, colour, two RGB G-buffers), 17 with
    1-channel G-buffers (depth, material id); under Welch degrees of freedom (STATMC_DOF_WELCH) + the sample count, which the
    pair test reads: 16 and 18 (channels 15, 16 the 1-channel G-buffers, channel 17 the count: the eight-plane Welch builds)."""
    if welch:
        return 16 if list(g_channels) == [3, 3] else 18
    # exactly two RGB G-buffers: the 15-channel image; every other set of up to two RGB and two 1-channel images: 17 channels,
    # absent slots zero (what statmc::FilmShards does on the C++ side)
    return 15 if list(g_channels) == [3, 3] else 17


def exchange_halo(layout, padded, group=None, via_host=False):
    """Fill the halo margins of `padded` ([ph, pw, C], interior already written) from the
    neighbouring ranks.  Works on any backend (nccl == RCCL on ROCm, gloo on CPU).  via_host
    stages the messages through host memory (gloo with device tensors: test setups only)."""
    L, r = layout, layout.r
    if L.world == 1 or r == 0:
        return

    def phase(pairs):
        ops, recvs = [], []
        for peer, send_view, recv_view in pairs:
            if peer is None:
                continue
            # whole rows of the padded image are contiguous: sent from and received into place
            # (the row-strip grids bench.py uses never take the staging copies below)
            in_place = not via_host and send_view.is_contiguous() and recv_view.is_contiguous()
            sbuf = send_view if in_place else send_view.contiguous()
            rbuf = recv_view if in_place else torch.empty_like(recv_view, memory_format=torch.contiguous_format)
            if via_host:
                sbuf, rbuf = sbuf.cpu(), rbuf.cpu()
            ops.append(dist.P2POp(dist.isend, sbuf, peer, group=group))
            ops.append(dist.P2POp(dist.irecv, rbuf, peer, group=group))
            if not in_place:
                recvs.append((recv_view, rbuf))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for view, buf in recvs:
            view.copy_(buf)  # (host -> device when via_host)

    y0, y1 = L.pt, L.pt + L.bh
    x0, x1 = L.pl, L.pl + L.bw
    # phase 1: columns (owned rows only)
    phase([
        (L.left, padded[y0:y1, x0:x0 + r], padded[y0:y1, 0:x0]),
        (L.right, padded[y0:y1, x1 - r:x1], padded[y0:y1, x1:x1 + L.pr]),
    ])
    # phase 2: rows over the full padded width (corners ride along)
    phase([
        (L.up, padded[y0:y0 + r, :], padded[0:y0, :]),
        (L.down, padded[y1 - r:y1, :], padded[y1:y1 + L.pb, :]),
    ])


class HaloInFlight:
    """A halo exchange that has been started (exchange_halo_start); wait() makes the current stream wait for it."""

    def __init__(self, reqs=()):
        self.reqs = list(reqs)

    def wait(self):
        for req in self.reqs:
            req.wait()
        self.reqs = []


def exchange_halo_start(layout, padded, group=None, via_host=False):
    """exchange_halo in two halves for row-strip grids: the border rows of `padded` must be written, the sends and
    receives are issued and the call returns; whatever is enqueued next (the accumulation of the rows that need no
    halo) runs beside the transfers; HaloInFlight.wait() orders the current stream behind them.  Whole rows of the
    padded image travel in place (no staging copies).  2-D grids and host-staged test setups (gloo) finish the exchange
    here and return a handle with nothing to wait for."""
    L, r = layout, layout.r
    if L.world == 1 or r == 0:
        return HaloInFlight()
    if L.gx > 1 or via_host:
        exchange_halo(layout, padded, group=group, via_host=via_host)
        return HaloInFlight()
    y0, y1 = L.pt, L.pt + L.bh
    ops = []
    for peer, send_view, recv_view in ((L.up, padded[y0:y0 + r, :], padded[0:y0, :]),
                                       (L.down, padded[y1 - r:y1, :], padded[y1:y1 + L.pb, :])):
        if peer is None:
            continue
        assert send_view.is_contiguous() and recv_view.is_contiguous()
        ops.append(dist.P2POp(dist.isend, send_view, peer, group=group))
        ops.append(dist.P2POp(dist.irecv, recv_view, peer, group=group))
    return HaloInFlight(dist.batch_isend_irecv(ops) if ops else ())


def gather_blocks(layout, block, film_f=None, dst=0, group=None, via_host=False):
    """Assemble the ranks' filtered blocks ([bh, bw, C] each) into the whole film on rank `dst`: SURVEY 8e's final
    gather of film-f (12 B/px).  Row strips land in place -- a strip is a contiguous slab of the film -- and 2-D blocks
    go through one staging tensor per block.  Returns the film on rank dst (film_f if given), None elsewhere."""
    L = layout
    if L.world == 1:
        return block
    dev = block.device
    src = block.contiguous()
    if via_host:
        src = src.cpu()
    if L.rank != dst:
        dist.gather(src, None, dst=dst, group=group)
        return None
    fw, fh = L.film_size
    c = block.shape[2]
    if film_f is None:
        film_f = torch.empty(fh, fw, c, dtype=block.dtype, device=dev)
    in_place = L.gx == 1 and not via_host
    if in_place:
        parts = [film_f[r * L.bh:(r + 1) * L.bh] for r in range(L.world)]
    else:
        parts = [torch.empty(L.bh, L.bw, c, dtype=block.dtype, device=src.device) for _ in range(L.world)]
    dist.gather(src, parts, dst=dst, group=group)
    if not in_place:
        for r, p in enumerate(parts):
            bx, by = r % L.gx, r // L.gx
            film_f[by * L.bh:(by + 1) * L.bh, bx * L.bw:(bx + 1) * L.bw].copy_(p)
    return film_f
