#!/bin/bash
# Does the reference REALLY compile against the adaptor?    Container only: /root/reference does not exist on
# the GPU box, and nothing of the reference enters this repository -- the work happens in a scratch directory.
#   1. copy the reference's src/ to a scratch tree, apply patches/000*.patch
#   2. g++ -std=c++17 -fsyntax-only, with the -D list CMake would pass, over the translation units that
#      touch the cv:: names: statistics/{estimator,buffer,statpath}.cpp, core/{film,api,integrator}.cpp --
#      include path: include/ (statmc_cv.hpp), the patched tree, and a logging stub for <glog/logging.h>
#      (tests/cpp/stubs; glog's submodule directory is empty in the checkout)
#   3. compile estimator.cpp + buffer.cpp to objects and LINK them with tests/cpp/ref_link_main.cpp against
#      libstatmc_hip.so alone (no OpenCV, no CUDA).  Link only: the binary is not run here.
#   4. (when STATTILE_IN / STATTILE_OUT are set) build tests/cpp/ref_stattile_main.cpp -- the reference's own StatTile<T>
#      on include/statmc_cv.hpp's cv::Vec, nothing else of the reference -- twice: g++ (no contraction: the survey's
#      probe build) and AMD clang -O3 -march=x86-64-v3 (-ffp-contract=on, the reference's own recipe, scripts/_build.sh),
#      and RUN both on the sample file STATTILE_IN -> ${STATTILE_OUT}.gcc / .clang (CPU only, no GPU call).
#      tests/test_host_cpu.py compares them with the oracle bit for bit.  This tests PRODUCT code (the adaptor's operators);
#      it is not a pin of the oracle (the build uses a logging stub and this repository's cv:: stand-in).
# Exit 77 = no reference checkout (skipped).
set -e
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/src/statistics" ] || { echo "no reference checkout at $REF: skipped"; exit 77; }
[ -f "$ROOT/statmc_amd/libstatmc_hip.so" ] || { echo "libstatmc_hip.so is not built"; exit 1; }
S=$(mktemp -d)
trap 'rm -rf "$S"' EXIT
cp -r "$REF/src" "$S/src"
cp "$REF/CMakeLists.txt" "$S/"
cd "$S"
git init -q . && git add -A >/dev/null 2>&1 && git -c user.email=x@x -c user.name=x commit -q -m reference
for p in "$ROOT"/patches/000*.patch; do git apply "$p"; done
FLAGS=(-std=c++17 -DPBRT_IS_LINUX -DPBRT_HAVE_ALLOCA_H -DPBRT_HAVE_MEMORY_H -DPBRT_HAVE_HEX_FP_CONSTANTS
       -DPBRT_HAVE_BINARY_CONSTANTS -DPBRT_HAVE_CONSTEXPR -DPBRT_CONSTEXPR=constexpr -DPBRT_HAVE_ALIGNAS -DPBRT_HAVE_ALIGNOF
       -DPBRT_HAVE_ITIMER -DPBRT_HAVE_NONPOD_IN_UNIONS -DPBRT_HAVE_MMAP "-DPBRT_NOINLINE=__attribute__((noinline))"
       -DPBRT_HAVE_POSIX_MEMALIGN -DPBRT_THREAD_LOCAL=thread_local -DNDEBUG
       -I"$ROOT/tests/cpp/stubs" -I"$ROOT/include" -Isrc -Isrc/core -Isrc/display)
TUS="statistics/estimator.cpp statistics/buffer.cpp statistics/statpath.cpp core/film.cpp core/api.cpp core/integrator.cpp"
fail=0
pids=()
for f in $TUS; do
  ( g++ "${FLAGS[@]}" -fsyntax-only "src/$f" > "$(echo $f | tr / _).log" 2>&1 && echo "syntax ok   src/$f" || { echo "SYNTAX FAIL src/$f"; grep -m5 error "$(echo $f | tr / _).log"; exit 1; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || fail=1; done
[ $fail = 0 ] || exit 1
g++ "${FLAGS[@]}" -O1 -c src/statistics/estimator.cpp -o estimator.o &
g++ "${FLAGS[@]}" -O1 -c src/statistics/buffer.cpp -o buffer.o &
g++ "${FLAGS[@]}" -O1 -c "$ROOT/tests/cpp/ref_link_main.cpp" -o main.o &
wait
# The one import neither object file nor library resolves is pbrtv4::DisplayStatic -- the tev viewer IPC of src/display
# (buffer.cpp:56-71; SURVEY section 2: out of scope, supplied by pbrt's own link line): tolerated by name, nothing else is.
LINK=(g++ main.o estimator.o buffer.o -L"$ROOT/statmc_amd" -lstatmc_hip -Wl,-rpath,"$ROOT/statmc_amd")
"${LINK[@]}" -o strict.out 2> link.log || true
undef=$(grep -o "undefined reference to \`[^']*'" link.log | sort -u | grep -v 'pbrtv4::DisplayStatic' || true)
if [ -n "$undef" ]; then echo "unresolved beyond the tev viewer:"; echo "$undef"; exit 1; fi
"${LINK[@]}" -Wl,--unresolved-symbols=ignore-all -o ref_estimator_on_statmc
echo "linked      main.o + estimator.o + buffer.o -> libstatmc_hip.so: $(nm -u ref_estimator_on_statmc | grep -c ' statmc_') statmc_* imports; left to pbrt's link line: $(grep -c "undefined reference to .pbrtv4::DisplayStatic" link.log) reference(s) to pbrtv4::DisplayStatic"
if nm -u ref_estimator_on_statmc | grep -qiE "opencv|cuda[A-Z]|_ZN2cv"; then echo "an OpenCV / CUDA import is left"; exit 1; fi
echo "reference compiles and links against the adaptor"
if [ -n "${STATTILE_IN:-}" ] && [ -n "${STATTILE_OUT:-}" ]; then
  CLANG=/opt/rocm/lib/llvm/bin/clang++
  g++ "${FLAGS[@]}" -O2 "$ROOT/tests/cpp/ref_stattile_main.cpp" -o stattile_gcc -L"$ROOT/statmc_amd" -lstatmc_hip -Wl,-rpath,"$ROOT/statmc_amd" -Wl,--unresolved-symbols=ignore-all &
  if [ -x "$CLANG" ]; then
    "$CLANG" "${FLAGS[@]}" -O3 -march=x86-64-v3 -ffp-contract=on -Wno-everything "$ROOT/tests/cpp/ref_stattile_main.cpp" -o stattile_clang -L"$ROOT/statmc_amd" -lstatmc_hip -Wl,-rpath,"$ROOT/statmc_amd" -Wl,--unresolved-symbols=ignore-all &
  fi
  wait
  ./stattile_gcc "$STATTILE_IN" "$STATTILE_OUT.gcc"
  echo "ran         the reference's StatTile<Float> / StatTile<Vec3> on statmc_cv.hpp's cv::Vec (g++, no contraction)"
  if [ -x stattile_clang ]; then
    ./stattile_clang "$STATTILE_IN" "$STATTILE_OUT.clang"
    echo "ran         ... and built by clang -O3 -march=x86-64-v3 -ffp-contract=on (the reference's own recipe)"
  fi
fi
