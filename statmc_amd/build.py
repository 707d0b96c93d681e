"""Builds libstatmc_hip.so for gfx950 with hipcc, in-tree (statmc_amd/libstatmc_hip.so)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libstatmc_hip.so")
DEFAULT_SO = SO
SOURCES = ["statmc_pointwise.hip", "statmc_filter.hip", "statmc_filter_sym.hip", "statmc_placement.hip", "statmc_abi.hip", "statmc_rccl.hip"]
HEADERS = ["statmc_device.h", "statmc_filter_common.h", "statmc_sym_experiments.h", "t_quantiles.h", os.path.join("..", "..", "include", "statmc.h"),
           os.path.join("..", "..", "include", "statmc_pinned_spec.h")]
# -ffp-contract=off: every fp32 op rounds once, in source order, like the CPU oracle build.
# -DSTATMC_PRODUCT_BUILD: any STATMC_SYM_* experiment switch next to it is a compile error (statmc_sym_experiments.h).
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-DSTATMC_PRODUCT_BUILD=1"]
# The hot loops are written in the order they should issue (stage by stage across a lane's pixels,
# loads ahead of the folds); the pre-RA machine scheduler only loses against that order: window filter
# 2.44 -> 2.40 ms, radiance accumulation 1.40 -> 1.30 ms in A/B builds on one box.
KERNEL_FLAGS = ["-mllvm", "-enable-misched=0"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libstatmc_hip.so cannot be built")


STAMP = DEFAULT_SO + ".src"   # hash of the sources + flags the library was built from (git-ignored, ships with the .so)


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS + KERNEL_FLAGS).encode())
    return h.hexdigest()


def stale():
    """True when libstatmc_hip.so exists but was built from other sources than the ones in csrc/ (mtimes do not
    survive a snapshot copy of the tree, a content hash does)."""
    if not os.path.exists(SO):
        return False
    try:
        return open(STAMP).read().strip() != source_hash()
    except OSError:
        return True


def needs_build():
    """A missing library or one built from other sources.  The rebuild goes to a temporary file and is renamed into
    place, so a process that has the old library mapped keeps a consistent image."""
    return not os.path.exists(SO) or stale()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = KERNEL_FLAGS if src not in ("statmc_abi.hip", "statmc_placement.hip", "statmc_rccl.hip") else []
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    tmp = SO + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    subprocess.check_call(cmd)
    os.replace(tmp, SO)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return SO


ROOT = os.path.dirname(HERE)
DENOISE_BIN = os.path.join(ROOT, "tools", "bin", "statmc_denoise")
RENDER_SIM_BIN = os.path.join(ROOT, "tools", "bin", "statmc_render_sim")
CV_ADAPTOR_BIN = os.path.join(ROOT, "tools", "bin", "test_cv_adaptor")   # tests/cpp/test_cv_adaptor.cpp: include/statmc_cv.hpp in use
TOOLS = {DENOISE_BIN: "statmc_denoise.cpp", RENDER_SIM_BIN: "statmc_render_sim.cpp",
         CV_ADAPTOR_BIN: os.path.join("..", "tests", "cpp", "test_cv_adaptor.cpp")}


TOOLS_STAMP = os.path.join(os.path.dirname(DENOISE_BIN), ".src")


def tools_hash():
    """Content hash of what the host tools are compiled from: their sources and every header under include/."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(ROOT, "include")
    files = [os.path.join(ROOT, "tools", src) for src in TOOLS.values()]
    files += [os.path.join(inc, f) for f in sorted(os.listdir(inc))]
    for path in files:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()


def tools_stale():
    try:
        return open(TOOLS_STAMP).read().strip() != tools_hash()
    except OSError:
        return True


def build_tools(force=False):
    """g++ build of the C++ host side (include/statmc_denoiser.hpp + tools/*.cpp: the offline
    denoise driver and the render-loop harness), linked against libstatmc_hip.so."""
    # Binaries built from other sources than the ones in the tree are rebuilt (content hash: mtimes do not survive a
    # snapshot copy of the tree); each goes to a temporary file and is renamed into place.
    if not force and all(os.path.exists(b) for b in TOOLS) and not tools_stale():
        return DENOISE_BIN
    if not os.path.exists(SO):
        build()
    os.makedirs(os.path.dirname(DENOISE_BIN), exist_ok=True)
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), "lib")
    for binary, src in TOOLS.items():
        tmp = binary + ".tmp%d" % os.getpid()
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-ffp-contract=off", "-pthread",
                               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", src), "-o", tmp,
                               "-L", HERE, "-lstatmc_hip", "-L", rocm_lib, "-Wl,-rpath,$ORIGIN/../../statmc_amd",
                               "-Wl,-rpath," + rocm_lib])
        os.replace(tmp, binary)
    with open(TOOLS_STAMP, "w") as f:
        f.write(tools_hash() + "\n")
    return DENOISE_BIN


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))
    print(build_tools(force=True))
