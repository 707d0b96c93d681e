/* statmc_pinned_spec.h -- the filter spec and significance level a freshly set-up device (and the CPU oracle) start
 * with.  REWRITTEN by tools/pin_from_dumps.sh from dumps of the CUDA build (the un-vendored stat_denoiser.cu,
 * /root/reference/.gitmodules:19-21): the spec that reproduces them becomes the default of library and oracle alike.
 * All zero = this build's own spec v2: nothing has been pinned yet.
 * Field order = statmc_filter_spec: gate, channel_rule, sides, dof, border, small_n. */
#ifndef STATMC_PINNED_SPEC_H
#define STATMC_PINNED_SPEC_H
#define STATMC_PINNED_SPEC {0, 0, 0, 0, 0, 0}
#define STATMC_PINNED_SIGNIFICANCE 0
#define STATMC_PINNED_FROM "nothing pinned: spec v2 defaults (parity with the CUDA denoiser unpinned)"
#endif
