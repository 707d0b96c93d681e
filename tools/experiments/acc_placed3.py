"""Round 5: is "apart" transitive?  One stat type (normal: a pure stream + 16 B/px of state), eight placed arenas of 5.93 GiB
(they land in different runs of class-B slots), three placed states (different class-A slots: a 900 MiB spacer between
them), every pair timed.  A uniform table = the classes are what the allocator thinks they are.
python tools/experiments/acc_placed3.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film

dev = torch.device("cuda:0")
api.setup(0)
W, H, S = 1920, 1080, 256
states, spacers = [], []
for k in range(3):
    states.append(film.FilmStats(W, H, dev, types=["normal"], placed=True))
    spacers.append(api.empty_placed((900 << 18,), torch.float32, dev, api.MEM_STATE))      # 900 MiB: the next state starts in another slot
arenas = []
for k in range(8):
    a = api.empty_placed((S, H, W, 3), torch.float32, dev, api.MEM_STREAM)
    for s0 in range(0, S, 32):
        a[s0:s0 + 32].uniform_()
    arenas.append(a)
info = api.placement_info()
base = None
print("map:", info["map"], flush=True)
print("state blocks at (GiB from the first):", [round((st.state["normal"]["mean"].data_ptr() - states[0].state["normal"]["mean"].data_ptr()) / 2 ** 30, 2) for st in states])
print("arena blocks at (GiB from the first state):", [round((a.data_ptr() - states[0].state["normal"]["mean"].data_ptr()) / 2 ** 30, 2) for a in arenas], flush=True)
for rnd in range(2):
    for si, st in enumerate(states):
        row = []
        for a in arenas:
            smp = {"normal": a}
            st.accumulate(smp)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    st.accumulate(smp)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 6)
            row.append("%.2f" % (12 * S * W * H / best / 1e9))
        print("state %d: TB/s per arena: %s" % (si, " ".join(row)), flush=True)
