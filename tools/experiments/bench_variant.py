"""Scratch: bench.py with a variant library.  usage: bench_variant.py path/to/variant.so [bench.py options]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
