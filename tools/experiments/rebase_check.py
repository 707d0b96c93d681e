"""Does the reference slot's trade (statmc_placement.hip: rebase) reach the shaders?  STATMC_PLACEMENT_FORCE_REBASE=1 STATMC_PLACEMENT_DEBUG=1
python tools/experiments/rebase_check.py -- the allocator pokes a word into both slots before the trade and reads both addresses after it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api
api.setup(0)
dev = torch.device("cuda:0")
a = api.empty_placed((1 << 20,), torch.float32, dev, api.MEM_STATE)
b = api.empty_placed((3 << 28,), torch.float32, dev, api.MEM_STREAM)
i = api.placement_info()
print("rebased", i["rebased"], "map", i["map"], "probes", i["probes"])
