/*
 * spada_probe.h -- MEASUREMENT entry points of libspada_spgemm.so.  Not part of the drop-in boundary (include/spada_ffi.h): nothing a
 * caller of the engine needs, no counterpart in the reference.  They exist so that the figures in profiles/ can be reproduced with
 * one call (scripts/probe_floor.py).
 */
#ifndef SPADA_PROBE_H
#define SPADA_PROBE_H

#include "spada_ffi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The floor under the task kernel: expand + scale only over the task list the context's LAST pipeline run has left (call
 * spada_dev_spgemm_fused or _symbolic first), static task assignment, descriptors and entry records prefetched a task ahead, no
 * ticket, no chain, no accumulator.  write = 0: the products are folded into a word nobody reads; write = 1: 12 bytes per product
 * stored to a scratch buffer of the probe's own (tasks x 2048 entries).  wgs_per_cu: workgroups of 512 threads per CU (1 .. 4).
 * reps launches are timed with HIP events on the engine stream; *ms_best / *ms_mean are per launch.  *tasks_skipped = tasks of the
 * older range path, which the probe leaves out. */
int spada_dev_probe_floor(spada_ctx *ctx, int write, uint32_t wgs_per_cu, uint32_t reps, double *ms_best, double *ms_mean,
                          uint64_t *tasks, uint64_t *tasks_skipped);

/* What spada_dev_csr_upload spent on the two arrays it derives from the matrix and keeps with it -- `rowid` (row of every entry, 4 B x
 * nnz) and `rext` (first / last column of every row, 8 B x rows): input pre-processing that no SpGEMM call's time contains.  Both are
 * built on the device (two small kernels: device_ms by HIP events); host_ms is 0 since round 6 (rounds 2 - 5 built rowid on the host). */
int spada_dev_csr_aux_cost(const spada_dev_csr *m, double *host_ms, double *device_ms, uint64_t *bytes);

/* Where the scatter's scratch arrays lie (spada_engine.hip, place_scratch): a context that keeps scattering into the same two arrays probes
 * them with the scatter's store pattern and tries other places for the column array (then, if none made a difference, for the value array), keeping the fastest.  *blocks_tried = blocks probed
 * besides the one the arrays started in (0: no choice was made -- arrays too small, too few runs, SPADA_PLACE=0); *probe_ms_first /
 * *probe_ms_kept = the probe's time where the arrays were / where they are now.  The two regimes are 2.5 and 3.3 ms apart on MI355X. */
int spada_dev_scratch_placement(const spada_ctx *ctx, uint32_t *blocks_tried, float *probe_ms_first, float *probe_ms_kept);

#ifdef __cplusplus
}
#endif
#endif
