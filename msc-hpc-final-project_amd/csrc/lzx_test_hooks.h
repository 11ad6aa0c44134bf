// lzx_test_hooks.h -- test-only entry of liblzx.so (NOT part of the boundary in include/lzx.h; the drop-in classes never call it).
// Forces the table shapes that large graphs get by themselves onto small test graphs, so that tests/ exercise the product
// library's own kernels in every shape: "pb_reduce" (minimum run length of a reduced run; 0 = every run plain), "pb_target"
// (values per gather item), "pb_unit" (entries per scatter unit), "pb_column_band" (8192 | 16384 | 18432: the last on one rank only), "pb_run_align", "pb_taper",
// "pb_dyn_share" (per cent of the gather pass left to its dynamic tail; 0 = every item dealt by the host), "pb_gather_grid" (at most this many gather workgroups),
// "pb_gather_nt" (the gather pass's stream loads non-temporal 1 / cached 0, whatever the stream's size),
// "pb_group" / "pb_group_force" (small row bands gathered one wavefront each), "narrow_slices", "tie_sort", "long_row",
// "item_len", "exchange_at_world_1" (a 1-rank RCCL communicator runs the several-rank loop, collectives included),
// "pb_carry_scan" (reduced steps: a row that spans lanes summed by the fixed-order cross-lane scan 1 / through LDS carry slots 0),
// "isolated_rows" (0: rows without an edge updated elementwise instead of as one scalar recurrence), "unnormalised_basis" (0: the
// resident basis holds q_j instead of u_j), "fuse_staged" (0: staged-columns kernel in a launch of its own; 1: behind the
// scatter units of the shared launch instead of ahead of them), "start_vector_scan" (0: lzx_lanczos_prepare_f64 always uploads x0 and
// sums its squares in one serial chain; default: one look at x0 first -- a constant vector is filled on the device, a sum that is
// exact in any order is formed by several threads), "defer_finish" (0: the blocked SpMV always launches k_pb_finish; default: in the lazy loop
// k_lazy_update adds the totals of multi-item gather bands where it reads v and the launch is left out).
#pragma once
#include <stdint.h>
#include "lzx.h"
#ifdef __cplusplus
extern "C"
#endif
int lzx_test_set_shape(lzx_handle h, const char *name, int64_t value);
// what shape the tables of the handle's graph took: "gather_items_dealt" / "gather_items_drawn" (static lists / dynamic tail of
// the gather pass), "gather_workgroups", "placement_tried" / "placement_kept" / "placement_us_<i>" (option placement_trials: candidates of
// the value stream timed at the last hand-over, the one kept, the SpMV time of candidate i in microseconds), "start_vector_was_constant"
// (the last prepared x0 was filled on the device instead of uploaded), "finish_launched" / "finish_deferrable" (the graph's blocked SpMV has a
// k_pb_finish launch / the lazy loop may leave it out)
#ifdef __cplusplus
extern "C"
#endif
int lzx_test_get_shape(lzx_handle h, const char *name, int64_t *value);
// One rank's share checked without its peers (a rank of an in-process group whose other handles hold no graph; C5's rank
// share on the one-GPU test box): this handle's LOCAL SpMV of x = 1 -- the row sums of its own rows, integers, exactly the
// rows' degrees when its tables are right -- into v_local[0 .. rows_local), and the position of every vertex of the caller's
// order in the full-length layout (rank * n_loc_pad + local row) into layout_pos[0 .. n), so that the caller can tell which
// vertex a local row is.  Voids a prepared decomposition, like lzx_bench_spmv.
#ifdef __cplusplus
extern "C"
#endif
int lzx_test_rank_row_sums(lzx_handle h, double *v_local, uint32_t *layout_pos, uint64_t *n_loc_pad);
// Latency of the communicator's two-double all-reduce: `reps` of them queued back to back on the handle's main stream between
// two events, microseconds each (collective: every rank calls it with the same reps; transports RCCL and peer windows).
#ifdef __cplusplus
extern "C"
#endif
int lzx_test_allreduce_latency(lzx_handle h, uint32_t reps, double *us_each);
