#!/bin/bash
# Do patches/000*.patch apply to the reference checkout?  Container only: /root/reference does not exist on the GPU box.
# Copies the touched files into a scratch git repository (nothing from the reference enters this repository), runs
# `git apply --check` and `git apply` for every patch in order, and checks the result.
set -e
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/src/statistics" ] || { echo "no reference checkout at $REF: skipped"; exit 77; }
S=$(mktemp -d)
trap 'rm -rf "$S"' EXIT
mkdir -p "$S/src/statistics" "$S/src/core"
cp "$REF/CMakeLists.txt" "$S/"
cp "$REF"/src/statistics/*.h "$REF"/src/statistics/*.cpp "$S/src/statistics/"
cp "$REF/src/core/film.h" "$REF/src/core/film.cpp" "$S/src/core/"
cd "$S"
git init -q . && git add -A && git -c user.email=x@x -c user.name=x commit -q -m reference
for p in "$ROOT"/patches/000*.patch; do
  git apply --check "$p"
  git apply "$p"
  echo "applied $(basename "$p")"
done
if grep -rn "opencv2/" src/statistics src/core/film.h src/core/film.cpp; then echo "an OpenCV include is left"; exit 1; fi
grep -q 'statmc_cv.hpp' src/statistics/statpbrt.h
grep -q 'statmc_hip' CMakeLists.txt
# every cv:: name the patched sources still use must exist in the adaptor
missing=0
for name in $(grep -rhoE "cv::(cuda::)?(stat_denoiser::)?[A-Za-z_0-9]+" src/statistics src/core/film.h src/core/film.cpp | sort -u); do
  leaf=${name##*::}
  grep -qE "\b$leaf\b" "$ROOT/include/statmc_cv.hpp" || { echo "statmc_cv.hpp lacks $name"; missing=1; }
done
[ $missing = 0 ]
echo "patches apply; $(git diff --stat HEAD | tail -1)"
