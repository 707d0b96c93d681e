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
// per-step housekeeping after the sweep (1) or before it (0) in EVERY build; -1 (the product): per build, as measured --
// hk_at_end() in statmc_filter_sym.hip (after the sweep in the one-buffer six-plane builds, before it with eight planes / two float buffers)
#define STATMC_SYM_HK_END_DEFAULT (-1)
#ifndef STATMC_SYM_HK_END
#define STATMC_SYM_HK_END STATMC_SYM_HK_END_DEFAULT
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

// round 6 (get the sweep off the per-step barrier), each measured and written up in HISTORY.md 4.3d:
// GSPLIT = G > 0: the two waves of a row split the window at a read-group boundary instead of at dx = 0 -- half 0 sweeps
// groups [0, G), half 1 groups [G, 11): no group is evaluated twice (84 instead of 86 (tap pair, pixel) units per lane and row)
// -1 (the product since round 6): G = 6 in every build, gsplit_of() in statmc_filter_sym.hip (half 0, the older wave of its SIMD pair,
// which also keeps house, takes 46 of the 84 units, half 1 38; one G for all builds because the split decides the order of the sums and
// blocks of a sharded film may run another build than the whole film).  1080p RGB, back to back, one box: dx = 0 split 1.440 ms |
// G = 5 1.42 | 6 1.383 | 7 1.340 | 8 1.442 (profiles/r06_ab*.log); per build: profiles/r06_modes.log.
// 0 = the window split at dx = STATMC_SYM_SPLIT (rounds 2 - 5); G > 0 = that G in every build.
#define STATMC_SYM_GSPLIT_DEFAULT (-1)
#ifndef STATMC_SYM_GSPLIT
#define STATMC_SYM_GSPLIT STATMC_SYM_GSPLIT_DEFAULT
#endif
// the same group-aligned split in the runtime-radius builds (r < 20, Welch)
#define STATMC_SYM_GSPLIT_RT_DEFAULT 1
#ifndef STATMC_SYM_GSPLIT_RT
#define STATMC_SYM_GSPLIT_RT STATMC_SYM_GSPLIT_RT_DEFAULT
#endif
// (Round 6 also built, measured and removed again -- source at commit e762ea4, logs profiles/r06_ab1.log .. r06_ab3.log, HISTORY.md 4.3d:
// STATMC_SYM_HK_HALF, the housekeeping on the other / on alternating waves of a SIMD pair (slower); STATMC_SYM_FLAGS, per-wave progress
// words in LDS instead of the per-step barrier (no change); STATMC_SYM_GSPLIT_MASK, a split finer than a read group (slower).)

// timing-only ablations of the Welch gate (round 4): 1 = no table gather (a constant quantile), 2 = no NaN select on the
// quotient, 4 = no quotient at all (nu = s^2): results are wrong with any bit set
#ifndef STATMC_SYM_WELCH_ABLATE
#define STATMC_SYM_WELCH_ABLATE 0
#endif

#define STATMC_SYM_DIAGNOSTIC_BITS                                                                                        \
    ((STATMC_SYM_ABLATE ? 1 : 0) | (STATMC_SYM_HK_END != STATMC_SYM_HK_END_DEFAULT ? 2 : 0) | (STATMC_SYM_PIPE ? 4 : 0) | (STATMC_SYM_STAMPS ? 8 : 0) | \
     (STATMC_SYM_SPLIT ? 16 : 0) | (STATMC_SYM_COUNT ? 32 : 0) | (STATMC_SYM_PRIO ? 64 : 0) | (STATMC_SYM_WELCH_ABLATE ? 256 : 0) | \
     (STATMC_SYM_GSPLIT != STATMC_SYM_GSPLIT_DEFAULT ? 512 : 0) | \
     (STATMC_SYM_GSPLIT_RT != STATMC_SYM_GSPLIT_RT_DEFAULT ? 8192 : 0))

#if defined(STATMC_PRODUCT_BUILD) && STATMC_SYM_DIAGNOSTIC_BITS != 0
#error "a STATMC_SYM_* diagnostic switch is set in the product build (statmc_amd/build.py): it would ship a wrong or slower filter"
#endif
