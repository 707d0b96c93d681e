"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/statmc.h
declares, and refuses to compute without a device (no silent fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from statmc_amd import api, build
    build.build()
    return api.load()


def declared_symbols(header="statmc.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(statmc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    """Every symbol a C header under include/ declares: the boundary (statmc.h) and the test / A-B switches (statmc_debug.h)."""
    syms = declared_symbols()
    assert len(syms) >= 20
    dbg = declared_symbols("statmc_debug.h")
    assert len(dbg) >= 8 and all(s.startswith("statmc_debug_") for s in dbg)
    missing = [s for s in syms + dbg if not hasattr(lib, s)]
    assert not missing, missing
    # nothing of the debug surface leaks into the boundary header, and no experiment transport is left in it
    assert not [s for s in syms if s.startswith("statmc_debug_") or "by_kernel" in s]


def test_python_export_list_matches_header():
    from statmc_amd import api
    assert sorted(api.EXPORTS) == declared_symbols()


def test_struct_layouts_match_header(lib):
    """ctypes mirrors of statmc_image / statmc_filter_args / statmc_stat_type vs the C compiler."""
    import subprocess
    import tempfile
    from statmc_amd import api
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "statmc.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(statmc_image), sizeof(statmc_filter_args),
         sizeof(statmc_stat_type), offsetof(statmc_filter_args, film_buffer),
         offsetof(statmc_filter_args, n_g_buffers), offsetof(statmc_filter_args, stream),
         offsetof(statmc_filter_args, roi_x0), offsetof(statmc_stat_type, samples));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    fa, st = api.FilterArgs, api.StatType
    assert got == [C.sizeof(api.Image), C.sizeof(fa), C.sizeof(st), fa.film_buffer.offset,
                   fa.n_g_buffers.offset, fa.stream.offset, fa.roi_x0.offset, st.samples.offset]


def test_no_device_means_error_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from statmc_amd import api
    assert lib.statmc_setup(0) == api.ERR_NO_DEVICE
    assert b"device" in lib.statmc_last_error().lower()
    a = api.FilterArgs()
    a.width, a.height = 8, 8
    assert lib.statmc_filter_f32x3(C.byref(a)) == api.ERR_NO_DEVICE       # setup() was never successful
    assert lib.statmc_accumulate(8, 8, None, 0, None) == api.ERR_NO_DEVICE
    with pytest.raises(api.StatmcError):
        api.setup(0)


def test_rccl_entry_points_check_their_arguments_without_a_device(lib):
    """statmc_halo_exchange_rccl and the communicator helpers (include/statmc.h): exported, RCCL bound at run time (no link-time
    dependency), argument errors reported before anything touches a device."""
    import subprocess
    from statmc_amd import api, build
    needed = subprocess.check_output(["readelf", "-d", build.SO], text=True)
    assert "rccl" not in needed.lower(), "libstatmc_hip.so must not link RCCL: it is bound by dlsym / dlopen on first use"
    assert lib.statmc_rccl_available() in (0, 1)
    assert lib.statmc_halo_exchange_rccl(None, 1, 2, 64, 64, 20, None, 0) == api.ERR_INVALID
    from statmc_amd.peer import Block
    blk = Block()
    assert lib.statmc_halo_exchange_rccl(C.byref(blk), 1, 2, 64, 8, 20, None, 0) == api.ERR_INVALID      # block smaller than the radius
    assert lib.statmc_halo_exchange_rccl(C.byref(blk), 1, 2, 64, 64, 20, None, 5) == api.ERR_INVALID     # rank outside the grid
    assert lib.statmc_halo_exchange_rccl(C.byref(blk), 1, 1, 64, 64, 20, None, 0) == 0                   # one block: nothing to exchange
    assert lib.statmc_halo_exchange_rccl(C.byref(blk), 1, 2, 64, 64, 20, None, 0) == api.ERR_INVALID     # no communicator
    assert lib.statmc_rccl_unique_id(None) == api.ERR_INVALID
    assert lib.statmc_rccl_comm_create(None, 2, 0, None) == api.ERR_INVALID


def test_product_does_not_import_oracle():
    """The product path may not import, include or load the checker (oracle/) in any form."""
    offenders = []
    for base in ("statmc_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                    continue
                for line in open(os.path.join(dirpath, f), errors="ignore"):
                    code = line.split("//")[0].split("#")[0] if not line.lstrip().startswith("#include") else line
                    if re.search(r"(import|include|CDLL|dlopen).*oracle", code):
                        offenders.append((f, line.strip()))
    assert not offenders, offenders


def test_per_device_state_needs_setup(lib):
    """Every setting lives in the state of a device that statmc_setup() has prepared: without one, the setters refuse
    (no process-global fallback that a second device would silently inherit)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from statmc_amd import api
    spec = api.FilterSpec(gate=1)
    assert lib.statmc_set_filter_spec(C.byref(spec)) == api.ERR_NO_DEVICE
    assert lib.statmc_set_filter_spec(None) == api.ERR_INVALID
    bad = api.FilterSpec(border=7)
    assert lib.statmc_set_filter_spec(C.byref(bad)) == api.ERR_INVALID
    assert lib.statmc_set_significance(1) == api.ERR_NO_DEVICE
    assert lib.statmc_set_significance(9) == api.ERR_INVALID
    assert lib.statmc_set_device(0) == api.ERR_NO_DEVICE          # never set up
    got = api.FilterSpec(gate=1, border=1)
    assert lib.statmc_get_filter_spec(C.byref(got)) == 0 and got.as_tuple() == (0,) * 6     # the default spec
    assert lib.statmc_get_significance() == 0
    q = (C.c_float * 4)(1, 2, 3, 4)
    assert lib.statmc_set_t_quantiles(0, q, 4) == api.ERR_NO_DEVICE


def test_filter_spec_struct_layout(lib):
    import subprocess
    import tempfile
    from statmc_amd import api
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "statmc.h"
int main(void) { printf("%zu %zu %zu %zu\n", sizeof(statmc_filter_spec), offsetof(statmc_filter_spec, small_n),
                        offsetof(statmc_filter_args, film_x0), sizeof(statmc_filter_args)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert got == [C.sizeof(api.FilterSpec), api.FilterSpec.small_n.offset, api.FilterArgs.film_x0.offset, C.sizeof(api.FilterArgs)]


def test_stale_library_is_refused(lib, tmp_path, monkeypatch):
    """The library records a hash of the sources it was built from; a library from other sources is not loaded."""
    from statmc_amd import api, build
    assert not build.stale()
    fake = tmp_path / "stamp"
    fake.write_text("0" * 64 + "\n")
    monkeypatch.setattr(build, "STAMP", str(fake))
    assert build.stale() and build.needs_build()
    monkeypatch.setattr(api, "_lib", None)
    with pytest.raises(RuntimeError, match="other sources"):
        api.load()


def test_placement_fast_level_is_the_bottom_of_a_tight_cluster(lib):
    """The level the placed allocator's class thresholds are multiples of (statmc_placement.hip: fast_level_of; pure arithmetic, no
    device): the smallest probe with two companions within 2 %.  Three cards this round broke simpler rules -- one probe a few per
    cent too fast (the smallest alone), one fast slot followed by six of slot 0's class (the third-smallest of all: a slow one, every
    slot read "apart from slot 0": profiles/r06_bench_x.json), slots that straddle two classes at 1.045 x the fast level (the
    third-smallest of the probes below slot 0's own)."""
    import ctypes as C
    f = lib.statmc_debug_placement_fast_level
    f.restype = C.c_float
    f.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float]

    def level(probes, self_ms=0.2):
        arr = (C.c_float * max(1, len(probes)))(*probes)
        return round(float(f(arr, len(probes), C.c_float(self_ms))), 4)

    slow, fast = 0.195, 0.177
    assert level([fast] + [slow, 0.194, 0.1955]) == 0.194                   # no fast cluster yet: the tight cluster is the slow one (no contrast to it: the calibration goes on)
    assert level([fast, 0.178] + [slow] * 2) == 0.0                          # two fast slots, two slow: no three within 2 % anywhere
    assert level([fast, 0.178, 0.179] + [slow] * 6) == fast                  # three: the bottom of the cluster
    assert level([0.170, fast, 0.178, 0.179] + [slow] * 9) == fast           # one probe a few per cent too fast has no companions
    assert level([0.1927, 0.1962, 0.1949, 0.1943, 0.1914, 0.1886, 0.1909, 0.1890, 0.1805]) == 0.1886   # (gpurun_out r06_bench_z3, slots 1 - 14: straddling slots ...)
    assert level([0.1927, 0.1962, 0.1949, 0.1886, 0.1890, 0.1805, 0.1886, 0.1815, 0.1800, 0.1808]) == 0.18   # ... until the fast cluster is there)
    assert level([fast, slow, 0.0, slow]) == 0.0                             # (0 = not probed)
    assert level([]) == 0.0
