// host_capi.cc -- a few extern "C" hooks over the C++ drop-in classes so that the test-suite can drive
// them through ctypes (a C++ caller uses the classes directly; `final` is the command-line face).
#include <algorithm>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "adjMatrix.h"
#include "cu_lanczos.h"
#include "eigen.h"
#include "multiplyOut.h"

static std::string g_host_err;

extern "C" {

const char *host_last_error() { return g_host_err.c_str(); }

// tridiagonal eigen-solver: d[k] in/out, e[k-1] in, z[k*k] out; returns 0 on success
int host_symtridiag(int k, double *d, const double *e, double *z) {
  std::vector<double> ec(e, e + (k > 1 ? k - 1 : 0));
  ec.push_back(0.0);
  return symtridiag_ql(k, d, ec.data(), z);
}

// The pipeline of main.cc on a graph file: adjMatrix(N, E, ifstream&) -> lanczosDecomp(cuda) ->
// eigenDecomp -> multOut / cu_multOut.  Outputs: ans[n], alpha[k], beta[k-1].  Returns n, or < 0.
long host_expm_file(const char *path, unsigned k, int cuda, int device_multout, double *ans, unsigned ans_len,
                    double *alpha, double *beta) {
  try {
    std::ifstream fs(path);
    if (fs.fail()) { g_host_err = std::string("cannot open ") + path; return -1; }
    unsigned n = 0, edges = 0;
    fs >> n >> n >> edges;
    adjMatrix A(n, edges, fs);
    if (ans_len < n) { g_host_err = "answer buffer too small"; return -2; }
    std::vector<double> x(n, 1.0);
    lanczosDecomp<double> L(A, k, x.data(), cuda != 0);
    if (alpha) std::copy(L.get_alpha(), L.get_alpha() + k, alpha);
    if (beta && k > 1) std::copy(L.get_beta(), L.get_beta() + (k - 1), beta);
    eigenDecomp<double> E(L);
    if (cuda && device_multout) cu_multOut(L, E, A, true);
    else multOut(L, E, A, cuda != 0);
    std::copy(L.answer(), L.answer() + n, ans);
    return static_cast<long>(n);
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

static long dump_csr(const adjMatrix &A, unsigned *row_offset, unsigned *col_idx, unsigned max_nnz) {
  // the arrays are private (friends only); operator<< prints them in full: "JA" cols, "IA" offsets
  std::ostringstream os;
  os << A;
  std::istringstream is(os.str());
  std::string tag;
  is >> tag;
  const unsigned nnz = A.get_nnz();
  if (nnz > max_nnz) { g_host_err = "col_idx buffer too small"; return -2; }
  for (unsigned i = 0; i < nnz; ++i) is >> col_idx[i];
  is >> tag;
  for (unsigned i = 0; i <= A.get_n(); ++i) is >> row_offset[i];
  return static_cast<long>(A.get_edges());
}

// Seeded generators of the adjMatrix drop-in: kind 'r' G(n, E), 'b' Barabasi (m = E), 'm' R-MAT (scale, draws = E).
long host_gen_csr(char kind, unsigned scale, unsigned n, unsigned long long E, unsigned long long seed,
                  unsigned *row_offset, unsigned *col_idx, unsigned max_nnz) {
  try {
    if (kind == 'm') return dump_csr(adjMatrix::rmat(scale, n, E, seed), row_offset, col_idx, max_nnz);
    if (kind == 'r') {
      adjMatrix A;  // default seed 1234 is what the (N, E) constructor uses
      (void)seed;
      A = adjMatrix(n, static_cast<unsigned>(E));
      return dump_csr(A, row_offset, col_idx, max_nnz);
    }
    if (kind == 'b') return dump_csr(adjMatrix(n, static_cast<unsigned>(E), 'b'), row_offset, col_idx, max_nnz);
    g_host_err = "unknown generator";
    return -1;
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

// Convergence monitor on a graph file: decomposition of dimension kmax, then multOutAdaptive(step, tol).
// ks[], changes[] receive up to `cap` evaluated dimensions / relative changes; returns how many, or < 0.
long host_adaptive_file(const char *path, unsigned kmax, unsigned step, double tol, int cuda, double *ans,
                        unsigned ans_len, unsigned *ks, double *changes, unsigned cap, unsigned *k_used) {
  try {
    std::ifstream fs(path);
    if (fs.fail()) { g_host_err = std::string("cannot open ") + path; return -1; }
    unsigned n = 0, edges = 0;
    fs >> n >> n >> edges;
    adjMatrix A(n, edges, fs);
    if (ans_len < n) { g_host_err = "answer buffer too small"; return -2; }
    std::vector<double> x(n, 1.0);
    lanczosDecomp<double> L(A, kmax, x.data(), cuda != 0);
    const convergenceReport rep = multOutAdaptive(L, A, step, tol, cuda != 0);
    std::copy(L.answer(), L.answer() + n, ans);
    const unsigned m = std::min<unsigned>(cap, static_cast<unsigned>(rep.k.size()));
    for (unsigned i = 0; i < m; ++i) { ks[i] = rep.k[i]; changes[i] = rep.rel_change[i]; }
    *k_used = rep.k_used;
    return static_cast<long>(rep.k.size());
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

// The pipeline with lanczosOptions: arnoldi_every (serial/lib/lanczos.h:44-57's fourth argument, generalised), adaptive_step /
// adaptive_tol (the device decomposition stops when its answer has converged), basis_fp32.  Outputs as host_expm_file plus
// info[0] = Krylov dimension used, info[1] = iterations actually run, info[2] = chunks evaluated, info[3] = converged;
// changes[] receives up to `cap` relative changes.  reorthog != 0: L.reorthog() before multOut (serial/lib/lanczos.cc:202-207).
long host_expm_options_file(const char *path, unsigned k, int cuda, int device_multout, unsigned arnoldi_every, unsigned adaptive_step,
                            double adaptive_tol, int basis_fp32, int reorthog, double *ans, unsigned ans_len, double *alpha,
                            double *beta, unsigned *info, double *changes, unsigned cap) {
  try {
    std::ifstream fs(path);
    if (fs.fail()) { g_host_err = std::string("cannot open ") + path; return -1; }
    unsigned n = 0, edges = 0;
    fs >> n >> n >> edges;
    adjMatrix A(n, edges, fs);
    if (ans_len < n) { g_host_err = "answer buffer too small"; return -2; }
    std::vector<double> x(n, 1.0);
    lanczosOptions o;
    o.arnoldi_every = arnoldi_every;
    o.adaptive_step = adaptive_step;
    o.adaptive_tol = adaptive_tol;
    o.basis_fp32 = basis_fp32 != 0;
    lanczosDecomp<double> L(A, k, x.data(), cuda != 0, o);
    const unsigned ku = L.get_krylov();
    if (alpha) std::copy(L.get_alpha(), L.get_alpha() + ku, alpha);
    if (beta && ku > 1) std::copy(L.get_beta(), L.get_beta() + (ku - 1), beta);
    if (reorthog) L.reorthog();
    eigenDecomp<double> E(L);
    if (cuda && device_multout) cu_multOut(L, E, A, true);
    else multOut(L, E, A, cuda != 0);
    std::copy(L.answer(), L.answer() + n, ans);
    const convergenceReport &rep = L.convergence();
    if (info) { info[0] = ku; info[1] = L.iterations_run(); info[2] = static_cast<unsigned>(rep.k.size()); info[3] = rep.converged; }
    for (unsigned i = 0; i < std::min<unsigned>(cap, static_cast<unsigned>(rep.rel_change.size())); ++i) changes[i] = rep.rel_change[i];
    return static_cast<long>(n);
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

// Drop-in parity, literally: the same graph file through lanczosDecomp(A, k, x, /*cuda*/false) and through
// lanczosDecomp(A, k, x, /*cuda*/true, {reference_order}); returns 1 when alpha, beta and every entry of Q are bit-identical,
// 0 when they are not (first difference in diff[0..2]: 0 alpha / 1 beta / 2 Q, index, -), < 0 on error.
long host_reference_order_check(const char *path, unsigned k, double *alpha_dev, double *alpha_cpu, unsigned *diff) {
  try {
    std::ifstream fs(path);
    if (fs.fail()) { g_host_err = std::string("cannot open ") + path; return -1; }
    unsigned n = 0, edges = 0;
    fs >> n >> n >> edges;
    adjMatrix A(n, edges, fs);
    std::vector<double> x(n, 1.0);
    lanczosDecomp<double> C(A, k, x.data(), false);
    lanczosOptions o;
    o.reference_order = true;
    lanczosDecomp<double> D(A, k, x.data(), true, o);
    if (alpha_dev) std::copy(D.get_alpha(), D.get_alpha() + k, alpha_dev);
    if (alpha_cpu) std::copy(C.get_alpha(), C.get_alpha() + k, alpha_cpu);
    for (unsigned j = 0; j < k; ++j)
      if (std::memcmp(D.get_alpha() + j, C.get_alpha() + j, sizeof(double)) != 0) { if (diff) { diff[0] = 0; diff[1] = j; } return 0; }
    for (unsigned j = 0; j + 1 < k; ++j)
      if (std::memcmp(D.get_beta() + j, C.get_beta() + j, sizeof(double)) != 0) { if (diff) { diff[0] = 1; diff[1] = j; } return 0; }
    // the bases: the CPU path stores Q row-major (serial/lib/lanczos.cc:47-48), the device path as k contiguous vectors
    const double *qd = D.basis();
    const double *qc = C.basis();
    for (unsigned j = 0; j < k; ++j)
      for (unsigned r = 0; r < n; ++r)
        if (std::memcmp(qd + static_cast<std::size_t>(j) * n + r, qc + j + static_cast<std::size_t>(r) * k, sizeof(double)) != 0) {
          if (diff) { diff[0] = 2; diff[1] = j; diff[2] = r; }
          return 0;
        }
    return 1;
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

// Loader only: CSR of a graph file as the adjMatrix file constructor builds it.
// row_offset[n+1], col_idx[2*E] (caller sizes them from the header); returns stored edges or < 0.
long host_load_csr(const char *path, unsigned *row_offset, unsigned *col_idx, unsigned max_nnz) {
  try {
    std::ifstream fs(path);
    if (fs.fail()) { g_host_err = std::string("cannot open ") + path; return -1; }
    unsigned n = 0, edges = 0;
    fs >> n >> n >> edges;
    adjMatrix A(n, edges, fs);
    return dump_csr(A, row_offset, col_idx, max_nnz);
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

// adjMatrix::load(path): parallel parser + device ingest (when a GPU is there) + binary side-car cache.
// info[0] = read from the cache, info[1] = CSR built on the GPU, info[2] = parser threads, info[3] = stored entries.
long host_load_path(const char *path, unsigned *row_offset, unsigned *col_idx, unsigned max_nnz, unsigned *info) {
  try {
    adjMatrix A = adjMatrix::load(path);
    const adjMatrix::loadReport &r = A.load_report();
    if (info) { info[0] = r.from_cache; info[1] = r.on_device; info[2] = r.threads; info[3] = A.get_nnz(); }
    return dump_csr(A, row_offset, col_idx, max_nnz);
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

// The whole pipeline on a path through adjMatrix::load, device decomposition spread over the GPUs of
// lzx_host_set_devices(devices[0..ndev)) (handles may share a GPU).  device_multout: 0 host multOut (downloads the
// basis lazily), 1 cu_multOut.  Returns n or < 0; gpus_used receives the number of handles the decomposition drove.
long host_expm_path_devices(const char *path, unsigned k, const int *devices, int ndev, int device_multout, double *ans,
                            unsigned ans_len, double *alpha, double *beta, unsigned *gpus_used) {
  try {
    lzx_host_set_devices(std::vector<int>(devices, devices + ndev));
    struct reset { ~reset() { lzx_host_set_devices({}); } } guard;
    adjMatrix A = adjMatrix::load(path);
    const unsigned n = A.get_n();
    if (ans_len < n) { g_host_err = "answer buffer too small"; return -2; }
    std::vector<double> x(n, 1.0);
    lanczosDecomp<double> L(A, k, x.data(), true);
    if (gpus_used) *gpus_used = L.gpus();
    if (alpha) std::copy(L.get_alpha(), L.get_alpha() + k, alpha);
    if (beta && k > 1) std::copy(L.get_beta(), L.get_beta() + (k - 1), beta);
    eigenDecomp<double> E(L);
    if (device_multout) cu_multOut(L, E, A, true);
    else multOut(L, E, A, true);
    std::copy(L.answer(), L.answer() + n, ans);
    // a second decomposition on the same graph re-uses the resident graph and evicts the first one's basis
    lanczosDecomp<double> L2(A, k, x.data(), true);
    if (L2.timings().setup_ms != 0.0) { g_host_err = "second decomposition uploaded the graph again"; return -4; }
    return static_cast<long>(n);
  } catch (const std::exception &e) {
    g_host_err = e.what();
    return -3;
  }
}

}  // extern "C"
