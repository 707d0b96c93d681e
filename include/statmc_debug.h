/* statmc_debug.h -- test, A/B and diagnostic switches of libstatmc_hip.so.  NOT part of the drop-in boundary
 * (include/statmc.h): nothing here has a counterpart in the reference, and a product host never calls it.  tests/,
 * bench.py's secondary legs and tools/experiments do.
 *
 * Like all library state the switches are PER DEVICE: they act on the calling thread's current device (statmc_set_device),
 * which must have been set up, and return STATMC_ERR_NO_DEVICE otherwise. */
#ifndef STATMC_DEBUG_H
#define STATMC_DEBUG_H

#include "statmc.h"

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Window-filter kernel of the device: 0 automatic; 1 window_filter_generic (global memory); 2 the one-sided LDS kernel's
 * runtime-radius build; 3 its compile-time r = 20 build (the kernel the pair-symmetric one replaced). */
int statmc_debug_force_filter_variant(int variant);
/* Older name of statmc_set_filter_split (include/statmc.h). */
int statmc_debug_force_filter_parts(int parts);
/* Parts per tile of the calling thread's last window-filter call. */
int statmc_debug_last_filter_parts(void);
/* ... and its tail split: the last *tail_rows tile rows of the image swept with *parts_hi parts (0, 0: uniform). */
int statmc_debug_last_filter_tail(int *parts_hi, int *tail_rows);

/* Film-major accumulation: n > 0 runs it as n resident workgroups; 0 = by shape (default: the large interleaved grid, or one
 * workgroup per compute unit on 1080p-sized films of placed buffers from 256 samples per launch up); -1 = never a resident grid. */
int statmc_debug_accumulate_resident_blocks(int n);
/* Workgroups of the calling thread's last statmc_accumulate launch (which launch shape the sizes chose: DESIGN.md 4.1). */
int statmc_debug_last_accumulate_grid(void);
/* 1 (default): RGB sample planes stream through LDS-DMA; 0: loads into registers (same bits). */
int statmc_debug_accumulate_dma(int on);
/* Film-major launch shape: grid_mode -1 = chosen by the batch length (default); 1 = one pass per workgroup, stat types
 * round-robin; 0 = capped grid with slots per type and a grid-stride walk.  dma_first 1 = the first rows of the LDS-DMA
 * ring are requested before the state loads (A/B; default 0). */
int statmc_debug_accumulate_launch(int grid_mode, int dma_first);
/* Experiment builds (-DSTATMC_ACC_OCC_AB=1): 3 = the film-major kernel compiled for three waves per SIMD (168 VGPRs; default 2
 * waves, 202 VGPRs).  Other builds ignore it. */
int statmc_debug_accumulate_occupancy(int waves_per_simd);
/* 2: the mean-only feature types of the film-major kernel prefetch twice as deep (default 1). */
int statmc_debug_accumulate_umul(int umul);
/* Tile-fed accumulation: prefetch depth of the mean-only types (1 | 2, default 2), item order, workgroups per CU. */
int statmc_debug_accumulate_tiles_variant(int umul, int order, int wg_per_cu);

/* The placed allocator's probe (statmc_amd/csrc/statmc_placement.hip) on memory of the caller's: streams [stream_ptr, + stream_bytes)
 * while every fourth step read-modify-writes 16 bytes inside [rmw_ptr, + rmw_bytes) -- the words there change.  Best of five, ms.
 * Two buffers in the same interference class: ~ 9 % slower than two in different ones (1-GiB stream, 64-MiB window). */
int statmc_debug_interference_probe(const void *stream_ptr, size_t stream_bytes, void *rmw_ptr, size_t rmw_bytes, float *ms);
/* The role of the statmc_malloc_placed block that holds `ptr` (any address inside it; window blocks included), -1 when `ptr` is
 * in none, the block was dealt without the wanted class, or the device tells no classes apart: what statmc_accumulate asks. */
int statmc_debug_placement_role(const void *ptr);
/* The placed allocator's fast level for a list of probe times against slot 0 (ms; <= 0 = not probed) and slot 0's probe against
 * itself (0 = unknown): what its class thresholds are multiples of.  Pure arithmetic, no device call (tests/test_abi_cpu.py). */
float statmc_debug_placement_fast_level(const float *probes_ms, int n, float self_ms);

/* Non-zero: the library was built with a timing-only / diagnostic switch (statmc_sym_experiments.h); its results are
 * not the product's and statmc_amd.api refuses to load it. */
int statmc_debug_diagnostic_build(void);
/* Welch degrees of freedom on the pair-symmetric kernel: the number of work items of the calling thread's last launch
 * that left their band of the quantile table and were computed again from the table in global memory (0 for a film of
 * uniform sample count; -1: the last pair-symmetric launch was not a Welch one).  Waits for the device. */
int statmc_debug_welch_far_items(void);
/* Largest filter workspace of the current device (diagnostic builds read their counters back from it). */
int statmc_debug_last_workspace(void **ptr, size_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* STATMC_DEBUG_H */
