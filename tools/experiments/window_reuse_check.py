"""A window block at the addresses a freed window left: do the two live blocks share memory?  (statmc_placement.hip, flush_translations;
tests/test_placement_gpu.py::test_a_window_at_addresses_another_window_left_reaches_its_own_memory runs the same steps.)
python tools/experiments/window_reuse_check.py [library.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import build
if len(sys.argv) > 1:
    build.SO = os.path.abspath(sys.argv[1])
from statmc_amd import api
api.setup(0)
dev = torch.device("cuda:0")
G = 1 << 28
# six one-slot blocks, every other one freed: three idle slots that do not lie side by side -- a 3-GiB block over them is a WINDOW
x = [api.empty_placed((G,), torch.float32, dev, api.MEM_STREAM) for _ in range(6)]
for k in (0, 2, 4):
    x[k] = None
a = api.empty_placed((3 * G,), torch.float32, dev, api.MEM_STREAM)
a.fill_(1.0)
pa = a.data_ptr()
torch.cuda.synchronize()
del a
b = api.empty_placed((G,), torch.float32, dev, api.MEM_STREAM)          # takes the first slot the window gave back
c = api.empty_placed((3 * G,), torch.float32, dev, api.MEM_STREAM)      # a window again: other slots, the first window's addresses
print("second window at the first one's addresses:", c.data_ptr() == pa, api.placement_info()["map"])
b.fill_(2.0)
c.fill_(3.0)
torch.cuda.synchronize()
for name, t, v in (("b", b, 2.0), ("c", c, 3.0)):
    bad = 0
    for k in range(0, t.numel(), G // 2):
        bad += int((t[k:k + G // 2] != v).sum().item())
    print("block %s: %d of %d values are not %.1f" % (name, bad, t.numel(), v))
