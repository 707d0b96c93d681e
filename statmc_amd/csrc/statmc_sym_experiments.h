// statmc_sym_experiments.h -- diagnostic and experiment switches of the pair-symmetric window filter
// (statmc_filter_sym.hip).  None of them belongs in the product: several give WRONG RESULTS (timing-only ablations,
// clock stamps and counters written into the patches).  They exist for tools/experiments/*.sh|py, which build variant
// libraries with -D flags (tools/experiments/build_variant.sh).  The product build (statmc_amd/build.py) defines
// STATMC_PRODUCT_BUILD: any switch set there is a compile error, so a stray -D cannot ship a wrong filter; the
// library also reports what it was built with (statmc_debug_diagnostic_build) and statmc_amd.api refuses a
// diagnostic build unless STATMC_ALLOW_DIAGNOSTIC_BUILD=1.
#pragma once

// timing-only ablation builds (tools/experiments/ablate_sym.sh): bit mask, results are wrong with any bit set
#ifndef STATMC_SYM_ABLATE
#define STATMC_SYM_ABLATE 0
#endif
// per-step housekeeping after the sweep instead of before (measured slower)
#ifndef STATMC_SYM_HK_END
#define STATMC_SYM_HK_END 0
#endif
// hand-placed ds_read_b128 one phase ahead of the arithmetic (1.57 ms against 1.43)
#ifndef STATMC_SYM_PIPE
#define STATMC_SYM_PIPE 0
#endif
// every wave sums the shader clocks it spends per step in housekeeping / sweep / barrier and leaves them in the last
// float4s of its item's patch (tools/experiments/stamps_sym.py): results are wrong
#ifndef STATMC_SYM_STAMPS
#define STATMC_SYM_STAMPS 0
#endif
// uneven split of the window columns between the two waves of a row (every value but 0 measured slower)
#ifndef STATMC_SYM_SPLIT
#define STATMC_SYM_SPLIT 0
#endif
// every wave counts its read groups and those without a member pair (tools/experiments/count_sym.py): results are wrong
#ifndef STATMC_SYM_COUNT
#define STATMC_SYM_COUNT 0
#endif
// s_setprio: 1 = the half-1 waves at raised issue priority (1.49 ms against 1.43), 2 = the half-0 waves during housekeeping (no change)
#ifndef STATMC_SYM_PRIO
#define STATMC_SYM_PRIO 0
#endif

// timing-only ablations of the Welch gate (round 4): 1 = no table gather (a constant quantile), 2 = no NaN select on the
// quotient, 4 = no quotient at all (nu = s^2): results are wrong with any bit set
#ifndef STATMC_SYM_WELCH_ABLATE
#define STATMC_SYM_WELCH_ABLATE 0
#endif

#define STATMC_SYM_DIAGNOSTIC_BITS                                                                                        \
    ((STATMC_SYM_ABLATE ? 1 : 0) | (STATMC_SYM_HK_END ? 2 : 0) | (STATMC_SYM_PIPE ? 4 : 0) | (STATMC_SYM_STAMPS ? 8 : 0) | \
     (STATMC_SYM_SPLIT ? 16 : 0) | (STATMC_SYM_COUNT ? 32 : 0) | (STATMC_SYM_PRIO ? 64 : 0) | (STATMC_SYM_WELCH_ABLATE ? 256 : 0))

#if defined(STATMC_PRODUCT_BUILD) && STATMC_SYM_ABLATE + STATMC_SYM_HK_END + STATMC_SYM_PIPE + STATMC_SYM_STAMPS + STATMC_SYM_SPLIT * STATMC_SYM_SPLIT + STATMC_SYM_COUNT + STATMC_SYM_PRIO + STATMC_SYM_WELCH_ABLATE != 0
#error "a STATMC_SYM_* diagnostic switch is set in the product build (statmc_amd/build.py): it would ship a wrong or slower filter"
#endif
