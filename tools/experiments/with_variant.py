"""Runs another script of this directory against a variant library: python tools/experiments/with_variant.py NAME script.py [args ...]
(NAME = tools/experiments/variants/NAME.so, built by build_variant.sh; the library is bound before the script imports statmc_amd.api)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1")
build.SO = os.path.join(ROOT, "tools", "experiments", "variants", sys.argv[1] + ".so")
assert os.path.exists(build.SO), build.SO
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
print("== variant", os.path.basename(build.SO), flush=True)
runpy.run_path(script, run_name="__main__")
