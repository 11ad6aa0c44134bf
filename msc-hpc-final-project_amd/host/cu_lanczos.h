// cu_lanczos.h -- `lanczosDecomp<T>`: the k-step Lanczos decomposition A ~ Q T Q^T, on the CPU
// (cuda == false) or on an MI355X through the lzx C ABI (cuda == true).
//
// Drop-in for parallel-final/lib/cu_lanczos.h:30-108 (file name kept so that `#include "cu_lanczos.h"`
// keeps working): same constructor -- all work happens in it --, same members alpha / beta / Q / x / ans /
// x_norm, same friends.  Layout contract kept from the reference:
//   cuda == false : Q is row-major n x k            (Q[j + row * k], parallel-final/lib/lanczos.cu:54)
//   cuda == true  : Q is k contiguous vectors of n  (&Q[k * n],      parallel-final/lib/cu_lanczos.cu:126)
// and multOut's `Qtrans` flag says which one it is given.
// The device path borrows the adjMatrix's device-resident graph (adjMatrix::device_graph(): one handle per GPU of
// lzx_host_devices(); several GPUs are driven from this one object, as parallel-two-cards/lib/cu_lanczos.cu:39-191
// drives its two cards) and leaves the basis in HBM: the k * n host copy `Q` is only made when somebody reads it
// (host multOut with Qtrans = true) -- cu_multOut / multOutAdaptive use the resident one.
// Not reproduced: the destructor / free_mem defects (cudaFree of a host pointer, leak of Q_d:
// cu_lanczos.h:75,84) and the silent half-built object after a failed allocation
// (cu_lanczos.cu:38-72) -- a failed device path throws std::runtime_error with the lzx error text.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "adjMatrix.h"
#include "device_graph.h"

struct lzx_ctx;

template <typename T, typename U>
T norm(const T *v, U n) {
  T s = 0;
  for (U i = 0; i < n; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}

template <typename T, typename U>
T inner_prod(const T *const v, const T *const w, const U n) {
  T s = 0;
  for (U i = 0; i < n; ++i) s += v[i] * w[i];
  return s;
}

template <typename T> class eigenDecomp;
template <typename T> class lanczosDecomp;
template <typename U, typename V> void check_ans(lanczosDecomp<U> &, lanczosDecomp<V> &);

// Timings of the device path of the last constructed decomposition (milliseconds).
struct lanczosTimings {
  double loop_ms = 0, spmv_ms = 0, vec_ms = 0, comm_ms = 0;
  double setup_ms = 0;   // upload + reshaping of the graph, 0 when the adjMatrix already had it on the device
  double fetch_ms = 0;   // download of the basis to the host, once somebody asked for it
  std::uint64_t spmv_bytes = 0;
  unsigned gpus = 0;
};

// What the reference's other constructors and its open problems add to the four-argument form:
//   arnoldi_every  serial/lib/lanczos.h:44-57's `bool arnoldi` (decompose_with_arnoldi, serial/lib/lanczos.cc:58-132) with its
//                  hard-coded `reorthog_every_k {2}` (:71) as a parameter: when j % e == 0 and j > 2, A q_j is orthogonalised
//                  against q_0 .. q_{j-2} before alpha_j is taken.  2 = the reference (which "does not give good results",
//                  serial/tests/numerical_test_orthog.cc:3-4: orthogonality goes between two passes); 1 keeps the basis
//                  orthogonal.  0 = off.
//   adaptive_step  > 0 (device path): the decomposition is advanced in chunks of this many iterations and stops at the first
//                  chunk whose answer y_k moved by less than adaptive_tol (relative 2-norm) -- the reference can only look at
//                  a finished decomposition (parallel-final/lib/multiplyOut.cu:25-49; writeup section 11); here convergence
//                  saves SpMVs.  get_krylov() is the dimension actually used afterwards.
//   basis_fp32     the device-resident basis stored as fp32 (half the HBM; alpha / beta unchanged bit for bit).
//   reference_order  device path: the loop with the CPU path's own reduction orders (SpMV one lane per row, inner product and
//                  norm one left-to-right accumulator: serial/lib/SPMV.cc:24-27, lanczos.cc:155-171), so that
//                  lanczosDecomp(A, k, x, true, {reference_order}) and lanczosDecomp(A, k, x, false) hold the same alpha, beta
//                  and Q BIT FOR BIT -- a parity instrument (slow), one GPU handle only.
struct lanczosOptions {
  unsigned arnoldi_every = 0;
  unsigned adaptive_step = 0;
  double adaptive_tol = 1e-10;
  bool basis_fp32 = false;
  bool reference_order = false;
};

struct convergenceReport {
  unsigned k_used = 0;
  bool converged = false;
  std::vector<unsigned> k;          // dimensions evaluated
  std::vector<double> rel_change;   // ||y_k - y_{k-step}|| / ||y_k|| (first entry: 1)
};

template <typename T>
class lanczosDecomp {
 public:
  lanczosDecomp() = delete;
  lanczosDecomp(adjMatrix &adj, const unsigned krylov, T *starting_vec, bool cuda);
  lanczosDecomp(adjMatrix &adj, const unsigned krylov, T *starting_vec, bool cuda, const lanczosOptions &opt);
  lanczosDecomp(lanczosDecomp &) = delete;
  lanczosDecomp &operator=(lanczosDecomp &) = delete;
  ~lanczosDecomp();

  // Releases the big host arrays early (main.cu:106 does this before the device run to stay out of swap).
  void free_mem();

  void get_ans() const;
  unsigned get_n() const { return A.get_n(); }
  unsigned get_krylov() const { return krylov_dim; }
  void check_ans(const T *analytic_ans) const;
  const lanczosTimings &timings() const { return times; }
  // Extensions over the reference's surface: read access to the results, and whether the basis is
  // still resident on the GPU (true after a device decomposition until free_mem()).
  const T *answer() const { return ans; }
  const T *get_alpha() const { return alpha; }
  const T *get_beta() const { return beta; }
  bool on_device() const { return graph != nullptr; }
  unsigned gpus() const { return graph ? static_cast<unsigned>(graph->ranks.size()) : 0; }
  // The host copy of the basis (layout above); a device decomposition downloads it on the first call.
  const T *basis() { ensure_host_basis(); return Q; }
  // adaptive_step > 0: what the chunked run evaluated; iterations (= SpMVs) it actually ran
  const convergenceReport &convergence() const { return adaptive; }
  unsigned iterations_run() const { return iters_run; }
  // Post-hoc orthonormalisation of the stored basis, the role of serial/lib/lanczos.cc:202-207 (LAPACKE_dgeqrf + dorgqr):
  // Q <- the orthonormal factor of Q = Q'R, here by two modified Gram-Schmidt sweeps (R's diagonal positive; LAPACK's
  // Householder form may flip the sign of a column).  A device decomposition brings its basis to the host first.
  void reorthog();

  friend class eigenDecomp<T>;
  template <typename U> friend void multOut(lanczosDecomp<U> &, eigenDecomp<U> &, adjMatrix &, bool);
  template <typename U> friend void cu_multOut(lanczosDecomp<U> &, eigenDecomp<U> &, adjMatrix &, bool);
  template <typename U> friend struct convergenceReport multOutAdaptive(lanczosDecomp<U> &, adjMatrix &, unsigned, double, bool);
  template <typename U, typename V> friend void check_ans(lanczosDecomp<U> &, lanczosDecomp<V> &);
  template <typename U> friend void write_ans(std::string filename, lanczosDecomp<U> &);

 private:
  adjMatrix &A;
  unsigned krylov_dim;
  T *alpha = nullptr;  // diagonal of T            [k]
  T *beta = nullptr;   // sub-diagonal of T        [k - 1]
  T *Q = nullptr;      // Lanczos basis            [n * k], layout above
  T *x = nullptr;      // starting vector          [n]
  T *ans = nullptr;    // e^A x once multOut ran   [n]
  T x_norm;
  lanczosOptions opts;
  convergenceReport adaptive;
  unsigned iters_run = 0;
  bool on_device_layout = false;       // Q (once on the host) holds k contiguous vectors rather than row-major n x k
  bool basis_lost = false;             // the resident basis was overwritten and could not be brought to the host
  std::shared_ptr<deviceGraph> graph;  // device path: the adjMatrix's graph on the GPU(s); the basis is resident there
  lanczosTimings times;

  void ensure_host_basis();             // device path: Q <- resident basis (once)
  void device_multout(const double *t, unsigned k, double *out);   // out = resident basis * t
  static void evict_cb(void *self);     // another decomposition is about to overwrite the resident basis

  void decompose();     // CPU:    serial/lib/lanczos.cc:9-56 == parallel-final/lib/lanczos.cu:17-60
  void decompose_with_arnoldi(unsigned every);   // CPU: serial/lib/lanczos.cc:58-132
  void cu_decompose();  // MI355X: replaces parallel-final/lib/cu_lanczos.cu:20-142
};
