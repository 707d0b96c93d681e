#!/bin/bash
# tools/pin_from_dumps.sh <dir> [options]: pin the filter spec of this build to a directory of dumps of the CUDA StatMC
# (inputs <stem>-<spp>-*.pfm incl. the reference's own film-f / t0-b0-mean-corr / t0-b0-discriminator outputs).
# Prints the per-channel L2 table of all 64 specs x 3 significance levels, writes the winner as the new default
# (include/statmc_pinned_spec.h), rebuilds and regenerates tests/golden/.  See tools/pin_from_dumps.py.
exec python3 "$(dirname "$0")/pin_from_dumps.py" "$@"
