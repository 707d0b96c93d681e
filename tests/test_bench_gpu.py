"""bench.py end to end on a small film: the one-GPU line with its secondary legs, and the N > 1 strong-scaling mode
(ONE film cut into blocks) with two ranks that share the test box's single GPU (gloo rendezvous, halos via the host)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--film", "512x256", "--spp", "8", "--steps", "3", "--warmup", "1"]


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    return json.loads(lines[0])


def test_single_gpu_line(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + ["--cpu-acc-rows", "8"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out)
    assert r["metric"] == "denoised_mpixels_per_s" and r["unit"] == "Mpixels/s" and r["n_gpus"] == 1 and r["steps"] == 3
    assert r["value"] > 0 and r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert r["config"]["film"] == "512x256" and r["config"]["filter_variant"] == "sym_r20"
    assert abs(r["value"] - 512 * 256 / r["ms_per_step"] / 1e3) < 1e-2 * r["value"]
    for k in ("roofline", "roofline_filter"):
        assert r[k]["bound"] == "hbm" and r[k]["unit"] == "GB/s" and abs(r[k]["frac"] - r[k]["achieved"] / r[k]["peak"]) < 1e-3
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] >= 1 and r["cpu_baseline"]["value"] > 0
    # secondary legs: the reference's own bracket through the C++ host side, tile-fed accumulation, copy rates
    assert r["cuda_time_bracket"]["cuda_time_bracket_ms"] > 0 and r["cuda_time_bracket"]["iterations"] == 4
    assert r["tile_fed_accumulate"]["achieved_GBs"] > 0
    assert r["host_copies"]["upload_bytes_per_px"] == 76 and r["host_copies"]["download_bytes_per_px"] == 12


@pytest.mark.parametrize("grid,blocks", [("rows", "1x2"), ("blocks", "2x1")])
def test_strong_scaling_two_ranks_share_the_device(gpu, grid, blocks):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--grid", grid] + COMMON
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["film"] == "512x256" and r["config"]["block_grid"] == blocks      # one film, two blocks
    assert abs(r["value"] - 512 * 256 / r["ms_per_step"] / 1e3) < 1e-2 * r["value"]      # film pixels per step time
    assert r["kernels"]["halo_exchange"]["avg_ms"] > 0
    assert "cpu_baseline" not in r
