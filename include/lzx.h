/*
 * lzx.h -- C ABI of the MI355X-native Lanczos e^A x engine (liblzx.so).
 *
 * This is the drop-in boundary for the hot path of hdelan/MSc-HPC-Final-Project
 * (SURVEY.md section 8): everything `lanczosDecomp<T>::cu_decompose()` does on the
 * device in parallel-final, behind plain pointers and sizes.  The reference has no
 * FFI of its own (one C++ binary); each entry point below names the reference
 * code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns LZX_OK (0) or a negative lzx_status; the message of
 *     the last failure on the calling thread is lzx_last_error();
 *   - all pointers are HOST pointers owned by the caller unless a name ends in
 *     `_dev`; the library owns all device memory;
 *   - a handle is bound to one GPU and is not thread-safe; distinct handles are;
 *   - the graph is an undirected, unweighted adjacency matrix handed over as
 *     pattern-only CSR (no values array): symmetric, columns ascending within a
 *     row, no duplicates -- what adjMatrix holds (parallel-final/lib/adjMatrix.h:26-30);
 *   - vertex order at this boundary is always the caller's; the library is free
 *     to (and does) relabel internally.
 */
#ifndef LZX_H_
#define LZX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lzx_ctx *lzx_handle;

typedef enum lzx_status {
    LZX_OK = 0,
    LZX_ERR_ARG = -1,    /* bad argument (null pointer, k == 0, sizes that do not match) */
    LZX_ERR_HIP = -2,    /* a HIP runtime call or kernel launch failed                   */
    LZX_ERR_STATE = -3,  /* call order: no graph yet, no decomposition yet, ...          */
    LZX_ERR_NOMEM = -4,  /* host or device allocation failed                             */
    LZX_ERR_COMM = -5,   /* RCCL not loadable / communicator failure                     */
    LZX_ERR_LIMIT = -6   /* size outside what this build supports (nnz per rank >= 2^32) */
} lzx_status;

/* Timings of the last lzx_lanczos_f64 call (all in milliseconds).  The three per-category sums come from HIP events
 * recorded on the loop's stream.  Every event is a barrier packet (about 5 us of drained pipeline between two dependent
 * kernels), so only every 4th iteration carries them (every iteration when k < 8, or with option "timing_marks_every"
 * = 1) and the sums are scaled to all k iterations. */
typedef struct lzx_stats {
    double loop_ms;       /* host wall clock around the k-iteration loop, device-synchronised on both
                             sides; excludes upload of x0 and download of alpha/beta/Q              */
    double spmv_ms;       /* sum over iterations of the SpMV(+alpha partial) launches, HIP events on
                             the stream they run on                                                */
    double spmv_ms_min;   /* fastest single iteration's SpMV time                                   */
    double vec_ms;        /* sum of the vector kernel launches (several ranks: + the scalar all-reduce) */
    double comm_ms;       /* sum of the exposed part of the all-gather(s), 0 at one rank            */
    uint32_t iters;       /* k                                                                      */
    uint32_t spmv_kernels;/* kernel launches counted in spmv_ms per iteration                       */
    uint64_t spmv_bytes;  /* algorithmic bytes of ONE SpMV on this rank:
                             4*nnz_local + 4*(rows_local+1) + 8*n (x once) + 8*rows_local (y)       */
} lzx_stats;

typedef struct lzx_graph_info {
    uint64_t n;           /* vertices (global)                                   */
    uint64_t nnz;         /* stored entries of the whole matrix = 2 * undirected edges */
    uint64_t max_degree;
    uint64_t rows_local;  /* rows this rank owns                                 */
    uint64_t nnz_local;   /* entries in those rows                               */
    uint64_t long_rows;   /* local rows handled by the split-row path            */
    uint64_t sell_padded; /* entries of the sliced-ELL body including padding    */
    uint64_t pb_entries;  /* local entries handled by the propagation-blocked passes (0 = mode off) */
    uint64_t active_vertices; /* vertices with at least one edge (global)         */
    uint64_t exchange_slice;  /* doubles each rank contributes to the per-iteration all-gather */
    uint32_t hub_entries; /* x entries staged in LDS per workgroup               */
    uint32_t world, rank;
    uint32_t reserved_;
    uint64_t pb_values;   /* values the blocked scatter passes hand to the gather pass per SpMV (padding included) */
    uint64_t pb_reduced_entries; /* of pb_entries: entries of the reduced bands, which travel as partial row sums */
    uint64_t exchange_chunk0;    /* doubles per rank in the first of the two all-gathers that overlap the blocked SpMV
                                    (0 = one all-gather per iteration) */
    uint64_t exchange_recv;      /* doubles this rank receives from the OTHER ranks per iteration: (world - 1) *
                                    exchange_slice with the plain all-gather; with the two-chunk exchange chunk 1 is
                                    sparse -- each peer sends only the entries this rank's rows reference -- and this
                                    is (world - 1) * exchange_chunk0 + what the peers pack for this rank */
    uint32_t placement_tried;    /* option "placement_trials": allocations of the value stream timed at the hand-over (the
                                    first one included; 0 = nothing to choose), */
    uint32_t placement_kept;     /* the one kept, */
    uint32_t placement_us[8];    /* and the SpMV time measured with each, in microseconds */
} lzx_graph_info;

/* ---- lifetime -------------------------------------------------------------------------------- */

/* Create an engine on GPU `device_id`.  Replaces the cudaMalloc/cudaStreamCreate block of
 * cu_decompose (parallel-final/lib/cu_lanczos.cu:37-86); unlike it, failure leaves nothing
 * half-built.                                                                                    */
int lzx_create(lzx_handle *out, int device_id);
/* GPUs visible to this process (0 when there is no usable device: never an error).  The drop-in classes use it to
 * decide between the device ingest and the host loader and to spread a decomposition over several cards, as
 * parallel-two-cards drives its two from one process (parallel-two-cards/lib/cu_lanczos.cu:73-190). */
int lzx_device_count(int *count);
void lzx_destroy(lzx_handle h);
const char *lzx_last_error(void);

/* ---- multi-GPU wiring (optional; call before handing over the graph) ----------------------------
 * Rows are dealt to ranks by degree rank (round-robin), every rank keeps a full-length copy of the
 * current Lanczos vector, and each iteration ends with an all-gather of the owned slices plus two
 * one-double all-reduces.  Replaces parallel-two-cards' split at rows0 = n/2 and its two
 * cudaMemcpyPeer per iteration (parallel-two-cards/lib/cu_lanczos.cu:62-67,125,158).
 *   lzx_comm_unique_id / lzx_comm_init_rank : one process per GPU, RCCL over xGMI.  Rank 0 makes the
 *       128-byte id, the caller ships it to the other ranks (any side channel), everyone calls init.
 *   lzx_comm_init_local : `world` handles inside ONE process (the reference's own two-cards model,
 *       any number of cards); exchanges are device-to-device copies.  Handles may share a GPU.      */
int lzx_comm_unique_id(uint8_t id[128]);
int lzx_comm_init_rank(lzx_handle h, const uint8_t id[128], int rank, int world);
/* SURVEY.md 8(b)'s sketch of the boundary in one call: `n_devices` handles, out[i] on GPU device_ids[i] (NULL: GPUs 0 .. n_devices - 1;
 * ids may repeat: handles may share a GPU), wired as ONE in-process communicator (lzx_create + lzx_comm_init_local) -- the
 * reference's parallel-two-cards model (one process, cudaSetDevice per card: parallel-two-cards/lib/cu_lanczos.cu:73-190) for
 * any number of cards.  Use the handles with the *_local entry points; destroy each with lzx_destroy.  On failure nothing is
 * left behind and every out[i] is NULL.                                                                                       */
int lzx_create_group(lzx_handle *out, int n_devices, const int *device_ids);
int lzx_comm_init_local(lzx_handle *hs, int world);
/*   lzx_comm_ipc_export / lzx_comm_ipc_init : one process per rank WITHOUT a collective library ("peer windows"): every
 *       rank's receive buffers are mapped into its peers (HIP inter-process memory handles), each rank's own kernel
 *       pushes its slice straight into them -- over xGMI point to point between GPUs, the access pattern
 *       parallel-two-cards/lib/cu_lanczos.cu:62-67 enables with cudaDeviceEnablePeerAccess -- and the iteration's two
 *       scalars travel through mailboxes in device memory (the host round trips of cu_lanczos.cu:104-105,119-120 gone).
 *       Ranks may share a GPU.  Every rank calls export (LZX_IPC_BLOB bytes out), the caller gathers the `world` blobs
 *       in rank order over any side channel, every rank calls init with the whole list.  A peer that does not arrive
 *       within LZX_IPC_TIMEOUT_MS (environment, default 20000) makes the call in progress fail with LZX_ERR_COMM.     */
#define LZX_IPC_BLOB 128
int lzx_comm_ipc_export(lzx_handle h, uint8_t blob[LZX_IPC_BLOB]);
int lzx_comm_ipc_init(lzx_handle h, const uint8_t *blobs /* [world][LZX_IPC_BLOB] */, int rank, int world);

/* ---- graph hand-over ---------------------------------------------------------------------------
 * lzx_set_graph_csr: upload of IA/JA, parallel-final/lib/cu_lanczos.cu:88-94 (`row_offset[n+1]`,
 * `col_idx[2E]`).  row_ptr is 64-bit as in serial/ (serial/lib/adjMatrix.h:23-24);
 * lzx_set_graph_csr32 takes parallel-final's `unsigned` arrays as they are.  Every rank of a
 * communicator passes the same whole graph and keeps its share (with option "sharded_ingest" the graph
 * never sits whole on any device: the CSR is streamed from the caller's memory).                  */
int lzx_set_graph_csr(lzx_handle h, uint64_t n, uint64_t nnz, const uint64_t *row_ptr,
                      const uint32_t *col_idx);
int lzx_set_graph_csr32(lzx_handle h, uint32_t n, uint32_t nnz, const uint32_t *row_ptr,
                        const uint32_t *col_idx);

/* Device-side ingest: `m` undirected edges as 0-based endpoint pairs, in any
 * order, duplicates allowed; symmetrised, sorted and de-duplicated on the GPU into the CSR that
 * adjMatrix::populate_sparse_matrix builds with a std::set (parallel-final/lib/adjMatrix.cc:21-46):
 * a self loop becomes one diagonal entry, as it does there.  Endpoints >= n are an LZX_ERR_ARG.   */
int lzx_set_graph_edges(lzx_handle h, uint64_t n, uint64_t m, const uint32_t *src, const uint32_t *dst);

/* Seeded synthetic graphs generated on the GPU (the reference's generators are seeded from
 * std::random_device, parallel-final/lib/make_graph.cc:23-24,61-62, and cannot be reproduced).
 * Integer specification shared with oracle/lanczos_oracle.c (orc_gen_er_keys, orc_gen_rmat_keys).
 * kind 0: Erdos-Renyi G(n, draws); kind 1: R-MAT with `scale` levels, 16-bit thresholds ta/tab/tabc,
 * endpoints >= n re-drawn (up to 8 attempts).                                                      */
int lzx_gen_graph(lzx_handle h, int kind, uint32_t scale, uint64_t n, uint64_t draws, uint64_t seed,
                  uint32_t ta, uint32_t tab, uint32_t tabc);

int lzx_get_graph_info(lzx_handle h, lzx_graph_info *out);
/* Copy the whole-graph CSR (caller's vertex order) back to the host: row_ptr[n+1], col_idx[nnz]. */
int lzx_get_graph_csr(lzx_handle h, uint64_t *row_ptr, uint32_t *col_idx);

/* ---- hot path ----------------------------------------------------------------------------------
 * lzx_spmv_f64: y = A x.  Kernel-level parity hook for cu_spMV1 (parallel-final/lib/cu_SPMV.cu:31-41)
 * and CPU spMV (serial/lib/SPMV.cc:19-28).  x[n], y[n] in the caller's vertex order.              */
int lzx_spmv_f64(lzx_handle h, const double *x, double *y);
int lzx_spmv_f64_local(lzx_handle *hs, int world, const double *x, double *y);

/* lzx_lanczos_f64: the whole k-step loop of cu_decompose (parallel-final/lib/cu_lanczos.cu:97-130),
 * i.e. serial/lib/lanczos.cc:9-56 on the device:
 *     q_0 = x0/||x0||;  for j<k: v = A q_j; alpha_j = v.q_j; v -= alpha_j q_j; v -= beta_{j-1} q_{j-1};
 *                                 beta_j = ||v||; q_{j+1} = v/beta_j          (last two only for j<k-1)
 * Outputs: alpha[k], beta[k-1] (beta may be NULL when k == 1), x_norm (may be NULL), and, when Q is
 * not NULL, Q[k*n] as k contiguous vectors -- the layout cu_decompose leaves on the host
 * (&Q[k*n], cu_lanczos.cu:126) and multOut consumes with Qtrans = true
 * (parallel-final/lib/multiplyOut.cu:42-44).  The basis also stays resident on the device for
 * lzx_multout_f64.  No breakdown guard for beta_j == 0, as in the reference.
 * With a communicator every rank must call it with the same arguments.                           */
int lzx_lanczos_f64(lzx_handle h, const double *x0, uint32_t k, double *alpha, double *beta,
                    double *Q, double *x_norm, lzx_stats *stats);
/* lzx_lanczos_f64 in three steps, so that a caller can bracket exactly the loop (inputs already in HBM):
 *   prepare: upload x0, q_0 = x0/||x0||, size the resident basis      (cu_lanczos.cu:30-34,88-94)
 *   run    : the k iterations, stream-synchronised on return           (cu_lanczos.cu:97-128)
 *   fetch  : alpha[k], beta[k-1], optionally Q[k*n]                     (cu_lanczos.cu:126,129-130)   */
int lzx_lanczos_prepare_f64(lzx_handle h, const double *x0, uint32_t k, double *x_norm);
int lzx_lanczos_run(lzx_handle h, lzx_stats *stats);
int lzx_lanczos_fetch_f64(lzx_handle h, uint32_t k, double *alpha, double *beta, double *Q);
int lzx_lanczos_fetch_f64_local(lzx_handle *hs, int world, uint32_t k, double *alpha, double *beta, double *Q);
/* The same loop in chunks (the reference's open problem of choosing k: parallel-final/lib/
 * multiplyOut.cu:25-49 can only evaluate a decomposition that has already run all its iterations; writeup section 11).
 * After lzx_lanczos_prepare_f64(h, x0, k_max, ..) each call runs up to `steps` more iterations of the SAME decomposition
 * -- every piece of loop state lives in HBM, so k iterations in chunks give the bits of k iterations in one go -- and
 * after every call alpha / beta / the basis of the iterations done so far can be used (lzx_lanczos_fetch_f64,
 * lzx_multout_f64, lzx_multout_change_f64 with k <= done), so that a caller who sees its answer converge stops paying for
 * SpMVs.  lzx_lanczos_run == run_steps(all that is left).  lzx_spmv_f64 / lzx_bench_spmv between two chunks void the
 * prepared state (LZX_ERR_STATE on the next chunk); lzx_multout* and lzx_lanczos_fetch* do not.
 * lzx_lanczos_progress: iterations done / prepared (prepared == 0: nothing to resume).                            */
int lzx_lanczos_run_steps(lzx_handle h, uint32_t steps, lzx_stats *stats);
int lzx_lanczos_run_steps_local(lzx_handle *hs, int world, uint32_t steps, lzx_stats *stats);
int lzx_lanczos_prepare_f64_local(lzx_handle *hs, int world, const double *x0, uint32_t k, double *x_norm);
int lzx_lanczos_progress(lzx_handle h, uint32_t *done, uint32_t *prepared);
/* Wait for everything queued on the handle's stream. */
int lzx_sync(lzx_handle h);
/* The same over `world` handles wired with lzx_comm_init_local, driven by one host thread. */
int lzx_lanczos_f64_local(lzx_handle *hs, int world, const double *x0, uint32_t k, double *alpha,
                          double *beta, double *Q, double *x_norm, lzx_stats *stats);

/* lzx_multout_f64: ans = Q t on the device-resident basis of the last decomposition (t[k] is
 * V (e^lambda * ||x|| * V[0,:]) computed by the host as in parallel-final/lib/multiplyOut.cu:30-40;
 * this call is the second dgemv, :42-46; cf. parallel-mult-on-card/lib/cu_multiplyOut.cu:66-71).
 * ans[n] in the caller's vertex order, complete on every rank.                                   */
int lzx_multout_f64(lzx_handle h, const double *t, uint32_t k, double *ans);
int lzx_multout_f64_local(lzx_handle *hs, int world, const double *t, uint32_t k, double *ans);

/* Convergence monitor without the n-vector crossing PCIe: y_k = Q_k t is formed on the device as in lzx_multout_f64 and
 * kept there; *rel_change = ||y_k - y_prev||_2 / ||y_k||_2 against the answer of the previous call since the last
 * prepare (1.0 for the first call).  The host-side stopping rule of multOutAdaptive (host/multiplyOut.cc) on two scalars. */
int lzx_multout_change_f64(lzx_handle h, const double *t, uint32_t k, double *rel_change);
int lzx_multout_change_f64_local(lzx_handle *hs, int world, const double *t, uint32_t k, double *rel_change);

/* ---- measurement hook --------------------------------------------------------------------------
 * Runs `reps` back-to-back SpMVs of the current graph on a device-resident vector and returns the
 * average and minimum HIP-event time of one SpMV (all its kernels) in milliseconds.              */
int lzx_bench_spmv(lzx_handle h, uint32_t reps, double *avg_ms, double *min_ms);

/* Measurement hook (no reference counterpart): the device's own streaming rates, for the roofline's "fraction of a
 * measured STREAM kernel on the same box" (SURVEY.md 8(d)): a read-only sum and a copy over a scratch buffer of `bytes`
 * bytes (rounded to 16; >= 256 MiB recommended), best of `reps`.  Results in GB/s (copy: bytes read + bytes written). */
int lzx_bench_stream(lzx_handle h, uint64_t bytes, uint32_t reps, double *read_gbs, double *copy_gbs);

/* Options, to be set before the graph is handed over (the last two: any time; setting one abandons a decomposition that
 * was being advanced in chunks):
 *   "hub_entries"           x values of the highest-degree vertices staged in LDS by the SpMV (0 = none)
 *   "propagation_blocking"  1 / 0 force the two-pass blocked treatment of non-staged columns on / off
 *                           (default: on for graphs whose x does not fit the L2s); with it off the sliced-ELL
 *                           rows are summed in the reference's order and come out bit-identical to serial/
 *   "overlap_exchange"      several ranks: 1 / 0 ask for / forbid the two-chunk all-gather that overlaps the blocked
 *                           SpMV (default: on for in-process groups; over RCCL only on request -- bench.py asks in its
 *                           guarded tuning phase -- until it has run once on two or more physical GPUs)
 *   "sparse_exchange"       1 / 0: with the two-chunk exchange, send each peer only the entries of the second chunk its rows
 *                           reference (default 1)
 *   "exchange_fp32"         1: several ranks exchange the new Lanczos vector rounded to fp32 (half the bytes; one all-gather,
 *                           sums stay fp64).  Off by default and NOT within the 1e-10 criterion: 6e-8 relative rounding per
 *                           entry and iteration ends near 1e-8 in the centrality vector (6.9e-9 measured on BASELINE C1;
 *                           the reference's own float runs end at 1.2e-6, parallel-final/output/single_double.txt:319)
 *                           -- SURVEY 8(f) N4
 *   "lazy_normalisation"    1: multiply (and, with several ranks, exchange) the unnormalised vector, so that alpha and
 *                           beta come out of one reduction per iteration (one 2-double all-reduce) and one vector kernel;
 *                           0: the reference's operation order.  Default: 1 with several ranks and in blocked mode, 0 on
 *                           one GPU in plain mode.
 *                           NB with it on, alpha_j = D / B is formed from the sums over the UNNORMALISED vector -- not the
 *                           reference's operation order (serial/lib/lanczos.cc:26: alpha_j = <A q_j, q_j>); same recurrence,
 *                           operands rounded at other places, covered at 1e-10 on the fixtures (tests/test_gpu_parity.py)
 *   "timing_marks_every"    iterations between the HIP-event timing marks behind lzx_stats (default 4; 1 = every iteration)
 *   "reorthogonalise"       e > 0: the Arnoldi pass of serial/lib/lanczos.cc:58-132 (decompose_with_arnoldi): when j % e == 0
 *                           and j > 2, A q_j is orthogonalised against q_0 .. q_{j-2} (modified Gram-Schmidt in the
 *                           reference's order, one launch per basis vector) before alpha_j is taken.  The reference
 *                           hard-codes e = 2 (:71) -- which lets orthogonality go between two passes and then removes
 *                           components T does not record: "neither give good results", serial/tests/numerical_test_orthog.cc:3-4,
 *                           and on BASELINE C2 at k = 50 it is off by O(1) -- e = 1 keeps the basis orthogonal (C2, k = 50:
 *                           1e-11 from the extended-precision referee where the plain loop is 3e-8 away).  Runs the reference's
 *                           operation order (lazy_normalisation off).  Default 0 = off, as in the reference's production path.
 *                           May be changed between decompositions.
 *   "basis_fp32"            1: the resident basis is STORED as fp32 (half the HBM: 4.0 -> 2.0 GB at C3, k = 50), the three
 *                           vectors the recurrence works on stay fp64, so alpha / beta are those of the fp64 loop bit for
 *                           bit; lzx_multout_f64 and the host fetch read the rounded columns (6e-8 relative per entry:
 *                           the centrality vector ends near 1e-8, outside the 1e-10 criterion; the reference's float runs
 *                           end at 1.2e-6, parallel-final/output/single_double.txt:58-63).  Selects the lazy loop (an error with
 *                           lazy_normalisation = 0 or reorthogonalise).  Default 0.
 *                           SURVEY 8(f) N4.  May be changed between decompositions.
 *   "reference_order"       1: the loop with serial/'s own REDUCTION orders, for a maintainer who wants the device path to
 *                           reproduce the CPU path's numbers exactly: the SpMV one lane per row of the caller's CSR, entries
 *                           added in ascending column order (serial/lib/SPMV.cc:24-27 = cu_spMV1, parallel-final/lib/cu_SPMV.cu:31-41),
 *                           inner product and norm ONE left-to-right accumulator over the caller's vertex order
 *                           (serial/lib/lanczos.cc:155-171); the elementwise updates as in every mode.  alpha, beta and
 *                           the basis then equal serial/'s restatement BIT FOR BIT at any k (tests: every fixture, and the
 *                           1 M- and 10 M-vertex benchmark graphs at k = 50), also together with "reorthogonalise".  A parity
 *                           instrument, not a fast path (about 70 ms per iteration at 10 M vertices); one rank only.
 *                           Default 0.  May be changed between decompositions.
 *   "placement_trials"      t >= 0: at the end of a graph hand-over in blocked mode the value stream between the SpMV's two
 *                           passes is allocated t more times, the SpMV timed with each candidate and the fastest kept (a few
 *                           SpMVs of set-up time each; results are bit-identical whichever wins).  Where the driver places
 *                           that one buffer decides up to 15 % of the SpMV on uniform graphs and 1-2 % on R-MAT ones for the
 *                           life of the allocation -- the reference's cudaMalloc blocks (cu_lanczos.cu:37-86) have no
 *                           counterpart.  At most three candidates are alive at a time (the best so far, the one being timed,
 *                           the last loser).  Default 2 (round 5: a library that may live inside a host framework does not
 *                           multiply a buffer at its hand-over; bench.py asks for 7 and reports every candidate's time);
 *                           at most 7; 0 = take the first allocation.
 *   "sharded_ingest"        s > 0: every hand-over entry point builds the graph WITHOUT ever holding all of it on one
 *                           device -- the loader of parallel-final/lib/adjMatrix.cc:21-46 for graphs beyond one card (SURVEY.md 7.1
 *                           step 7).  The whole-graph hand-over sorts all 2 m directed entries at once and leaves the whole CSR
 *                           on every rank; with this option a rank sweeps the source (the seeded generator re-drawn, or the
 *                           resident endpoint pairs) in bounded batches -- degrees, then the staged-column counts that order
 *                           rows of equal degree, then ITS OWN rows -- and nothing is exchanged: all ranks compute the same
 *                           ranking from the same source.  The tables are those of the whole-graph hand-over entry for entry
 *                           (same bits in every result).  s = 1: batches of 2^28 entries; s >= 2: exactly s batches per sweep
 *                           (tests).  With several ranks lzx_get_graph_csr then fails with LZX_ERR_STATE (no rank holds the
 *                           graph), the matrix must be symmetric as documented above (the sparse exchange lists rely on it),
 *                           and "reference_order" stays a one-rank instrument.  lzx_set_graph_csr / _csr32: the CSR stays in the
 *                           caller's memory and is streamed past the device in chunks of whole rows (row_ptr once, col_idx
 *                           twice); the rank keeps its own rows -- what parallel-two-cards does for its two halves
 *                           (parallel-two-cards/lib/cu_lanczos.cu:94-95,108-109).  Default 0.  C5, rank 0 of 8: 19 GB at the peak and 8.7 GB
 *                           resident instead of 102 / 17+ GB, 12.7 s instead of 5.8 s (profiles/r4_sharded_ingest.txt).
 * These twelve are all liblzx.so knows.  The experiment knobs and test hooks behind DESIGN.md's tuning log ("pb_*",
 * "phase_mask", "exchange_at_world_1", ...) exist only in liblzx_dbg.so, the same sources built with -DLZX_DEBUG_KNOBS
 * (`make debug`); tools/perf_probe.py and the tests that need them load that library.                               */
int lzx_set_option(lzx_handle h, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* LZX_H_ */
