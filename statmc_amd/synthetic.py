"""Seeded synthetic sample streams standing in for the reference's scenes (whose geometry is
not in the tree: scripts/_download-scenes.sh).  Shape of the data follows SURVEY.md section 8d:
a piecewise-smooth ground truth (Voronoi regions with linear shading gradients and hard edges),
per-region constant albedo / normal / depth / material id, log-normal radiance noise, 20 % of
zero-radiance paths (Box-Cox -> -2) and rare fireflies.

Written with torch ops only so the same code runs on the host (small parity cases, fed to the
CPU oracle) and on the GPU (full-size benchmark streams, generated in place in HBM).
"""
import math

import torch

FEATURES = ("radiance", "normal", "albedo", "depth", "materialid")
CHANNELS = {"radiance": 3, "normal": 3, "albedo": 3, "depth": 1, "materialid": 1}


class Scene:
    """Ground truth of one synthetic film."""

    def __init__(self, width, height, n_regions=12, seed=1, device="cpu", x_offset=0, y_offset=0,
                 full_width=None, full_height=None):
        g = torch.Generator(device="cpu").manual_seed(seed)
        fw, fh = full_width or width, full_height or height
        sites = torch.rand(n_regions, 2, generator=g) * torch.tensor([fw, fh], dtype=torch.float32)
        albedo = 0.05 + 0.9 * torch.rand(n_regions, 3, generator=g)
        normal = torch.randn(n_regions, 3, generator=g)
        normal = normal / normal.norm(dim=1, keepdim=True)
        depth = 1.0 + 49.0 * torch.rand(n_regions, generator=g)
        grad = (torch.rand(n_regions, 2, generator=g) - 0.5) * 2.0 / max(fw, fh)
        base = 0.3 + 1.7 * torch.rand(n_regions, generator=g)
        ys = torch.arange(height, dtype=torch.float32) + y_offset
        xs = torch.arange(width, dtype=torch.float32) + x_offset
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        d2 = (xx[..., None] - sites[:, 0]) ** 2 + (yy[..., None] - sites[:, 1]) ** 2
        region = d2.argmin(dim=2)                                   # [H, W]
        irr = base[region] + grad[region][..., 0] * (xx - sites[region][..., 0]) \
            + grad[region][..., 1] * (yy - sites[region][..., 1])
        irr = irr.clamp_min(0.05)
        self.width, self.height = width, height
        self.device = torch.device(device)
        self.region = region.to(self.device)
        self.albedo = albedo[region].to(self.device)                # [H, W, 3]
        self.normal = normal[region].to(self.device)
        self.depth = depth[region].to(self.device)                  # [H, W]
        self.materialid = region.to(torch.float32).to(self.device)
        self.irradiance = irr.to(self.device)
        self.truth = self.albedo * self.irradiance[..., None]       # noise-free radiance

    def samples(self, n_samples, seed, sigma=1.0, zero_frac=0.2, firefly_frac=5e-4, jitter=0.01,
                features=FEATURES):
        """Returns {feature: [S, H, W, C] float32 tensor} on self.device."""
        dev = self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        h, w = self.height, self.width
        out = {}
        if "radiance" in features:
            noise = torch.randn(n_samples, h, w, 3, generator=g, device=dev)
            noise.mul_(sigma).sub_(0.5 * sigma * sigma).exp_()
            noise.mul_(self.truth.unsqueeze(0)).mul_(1.0 / (1.0 - zero_frac))
            u = torch.rand(n_samples, h, w, 1, generator=g, device=dev)
            noise.mul_((u >= zero_frac).to(torch.float32))
            noise.mul_(1.0 + 999.0 * (u > 1.0 - firefly_frac).to(torch.float32))
            out["radiance"] = noise
        for name in ("normal", "albedo"):
            if name in features:
                base = getattr(self, name)
                j = torch.randn(n_samples, h, w, 3, generator=g, device=dev).mul_(jitter)
                out[name] = j.add_(base.unsqueeze(0))
        for name in ("depth", "materialid"):
            if name in features:
                base = getattr(self, name)
                if name == "depth":
                    j = torch.randn(n_samples, h, w, generator=g, device=dev).mul_(jitter)
                    out[name] = j.add_(base.unsqueeze(0)).unsqueeze(-1)
                else:
                    out[name] = base.unsqueeze(0).expand(n_samples, h, w).contiguous().unsqueeze(-1)
        return out


def sample_schedule(spp_total, pixelsamples=4):
    """Per-iteration batch sizes of the reference's exponential schedule
    (statpath.cpp:272-279): 4, 4, 8, 16, ... cumulative 4*2^(i-1)."""
    batches, cum = [], 0
    i = 1
    while cum < spp_total:
        b = pixelsamples if i <= 2 else pixelsamples << (i - 2)
        b = min(b, spp_total - cum)
        batches.append(b)
        cum += b
        i += 1
    return batches
