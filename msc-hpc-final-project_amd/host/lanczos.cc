// lanczos.cc -- CPU and device paths of lanczosDecomp<T> (see cu_lanczos.h).
#include <algorithm>
#include <chrono>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "SPMV.h"
#include "cu_lanczos.h"
#include "eigen.h"
#include "lzx.h"

template <typename T>
lanczosDecomp<T>::lanczosDecomp(adjMatrix &adj, const unsigned krylov, T *starting_vec, bool cuda)
    : lanczosDecomp(adj, krylov, starting_vec, cuda, lanczosOptions{}) {}

template <typename T>
lanczosDecomp<T>::lanczosDecomp(adjMatrix &adj, const unsigned krylov, T *starting_vec, bool cuda, const lanczosOptions &opt)
    : A{adj}, krylov_dim{krylov}, opts{opt} {
  if (krylov == 0) throw std::invalid_argument("lanczosDecomp: krylov dimension must be positive");
  const std::size_t n = A.get_n();
  alpha = new T[krylov];
  beta = new T[krylov > 1 ? krylov - 1 : 1];
  if (!cuda) Q = new T[n * krylov];   // the device path keeps the basis in HBM (ensure_host_basis)
  x = new T[n];
  ans = new T[n];
  x_norm = norm(starting_vec, A.get_n());
  std::copy(starting_vec, starting_vec + n, x);
  on_device_layout = cuda;
  if (cuda) cu_decompose();
  else if (opts.arnoldi_every > 0) decompose_with_arnoldi(opts.arnoldi_every);
  else decompose();
  if (!cuda) iters_run = krylov_dim;
}

template <typename T>
lanczosDecomp<T>::~lanczosDecomp() {
  free_mem();
  delete[] ans;
  ans = nullptr;
}

template <typename T>
void lanczosDecomp<T>::free_mem() {
  delete[] alpha; alpha = nullptr;
  delete[] beta; beta = nullptr;
  delete[] Q; Q = nullptr;
  delete[] x; x = nullptr;
  if (graph) {   // the graph stays with the adjMatrix; only the claim on the resident basis goes
    if (graph->owner == this) { graph->owner = nullptr; graph->evict = nullptr; }
    graph.reset();
  }
}

// The three-term recurrence with two ping-pong vectors; q_j is copied into column j of the row-major
// basis at the end of step j.
template <typename T>
void lanczosDecomp<T>::decompose() {
  const unsigned n = A.get_n(), k = krylov_dim;
  std::vector<T> v(n), cur(n), prev(n);
  const T xn = norm(x, n);
  for (unsigned r = 0; r < n; ++r) cur[r] = x[r] / xn;

  for (unsigned j = 0; j < k; ++j) {
    spMV(A, cur.data(), v.data());
    alpha[j] = inner_prod(v.data(), cur.data(), n);
    for (unsigned r = 0; r < n; ++r) v[r] -= alpha[j] * cur[r];
    if (j > 0)
      for (unsigned r = 0; r < n; ++r) v[r] -= beta[j - 1] * prev[r];
    for (unsigned r = 0; r < n; ++r) Q[j + static_cast<std::size_t>(r) * k] = cur[r];
    if (j + 1 < k) {
      beta[j] = norm(v.data(), n);
      for (unsigned r = 0; r < n; ++r) prev[r] = v[r] / beta[j];
      cur.swap(prev);  // cur = q_{j+1}, prev = q_j
    }
  }
}

// decompose() with the Arnoldi pass of serial/lib/lanczos.cc:85-90 every `every` iterations (the reference: 2): A q_j against
// q_0 .. q_{j-2}, modified Gram-Schmidt (each inner product over the updated v), before alpha_j is taken.
template <typename T>
void lanczosDecomp<T>::decompose_with_arnoldi(unsigned every) {
  const unsigned n = A.get_n(), k = krylov_dim;
  std::vector<T> v(n), cur(n), prev(n), kept(static_cast<std::size_t>(k) * n);   // q_m as contiguous vectors for the passes
  const T xn = norm(x, n);
  for (unsigned r = 0; r < n; ++r) cur[r] = x[r] / xn;

  for (unsigned j = 0; j < k; ++j) {
    spMV(A, cur.data(), v.data());
    if (j % every == 0 && j > 2) {
      for (unsigned m = 0; m + 1 < j; ++m) {
        const T *qm = kept.data() + static_cast<std::size_t>(m) * n;
        const T dot = inner_prod(v.data(), qm, n);
        for (unsigned r = 0; r < n; ++r) v[r] -= dot * qm[r];
      }
    }
    alpha[j] = inner_prod(v.data(), cur.data(), n);
    for (unsigned r = 0; r < n; ++r) v[r] -= alpha[j] * cur[r];
    if (j > 0)
      for (unsigned r = 0; r < n; ++r) v[r] -= beta[j - 1] * prev[r];
    for (unsigned r = 0; r < n; ++r) Q[j + static_cast<std::size_t>(r) * k] = cur[r];
    std::copy(cur.begin(), cur.end(), kept.begin() + static_cast<std::size_t>(j) * n);
    if (j + 1 < k) {
      beta[j] = norm(v.data(), n);
      for (unsigned r = 0; r < n; ++r) prev[r] = v[r] / beta[j];
      cur.swap(prev);
    }
  }
}

namespace {
void lzx_or_throw(int rc, const char *what) {
  if (rc != LZX_OK) throw std::runtime_error(std::string(what) + ": " + lzx_last_error());
}
}  // namespace

template <typename T>
void lanczosDecomp<T>::evict_cb(void *self) {
  auto *L = static_cast<lanczosDecomp<T> *>(self);
  // the basis is about to be overwritten: bring it to the host; if that fails the decomposition has no basis any more and
  // says so when somebody asks for it (ensure_host_basis assigns Q only after a complete fetch)
  try { L->ensure_host_basis(); } catch (...) { L->basis_lost = true; }
  L->graph.reset();
}

template <typename T>
void lanczosDecomp<T>::cu_decompose() {
  const unsigned n = A.get_n(), k = krylov_dim;
  const bool fresh = !A.dev;
  graph = A.device_graph();   // uploads + reshapes now unless the ingest or an earlier decomposition already did
  if (graph->owner && graph->owner != this && graph->evict) graph->evict(graph->owner);
  graph->owner = this;
  graph->evict = &lanczosDecomp<T>::evict_cb;
  lzx_handle *hs = graph->ranks.data();
  const int world = static_cast<int>(graph->ranks.size());
  lzx_graph_info gi;
  lzx_or_throw(lzx_get_graph_info(hs[0], &gi), "lzx_get_graph_info");
  std::cout << "\nUsing " << (gi.sell_padded + gi.n) * 4 + (static_cast<std::uint64_t>(k) + 3) * gi.n * 8
            << " bytes of HBM for the reshaped graph and " << k << " resident Lanczos vectors on " << world << " GPU handle(s)\n";

  for (int p = 0; p < world; ++p) {   // loop options: may change between decompositions on the resident graph
    lzx_or_throw(lzx_set_option(hs[p], "reorthogonalise", opts.arnoldi_every), "lzx_set_option(reorthogonalise)");
    lzx_or_throw(lzx_set_option(hs[p], "basis_fp32", opts.basis_fp32 ? 1 : 0), "lzx_set_option(basis_fp32)");
    lzx_or_throw(lzx_set_option(hs[p], "reference_order", opts.reference_order ? 1 : 0), "lzx_set_option(reference_order)");
  }
  lzx_stats st{};
  auto run = [&](const double *x0, double *a, double *b) {
    if (opts.adaptive_step == 0) {
      if (world == 1) lzx_or_throw(lzx_lanczos_f64(hs[0], x0, k, a, b, nullptr, nullptr, &st), "lzx_lanczos_f64");
      else lzx_or_throw(lzx_lanczos_f64_local(hs, world, x0, k, a, b, nullptr, nullptr, &st), "lzx_lanczos_f64_local");
      iters_run = k;
      return;
    }
    // Chunks of adaptive_step iterations; after each, y_k = ||x|| Q_k V_k e^{Lambda_k} V_k^T e_1 from the leading block of T
    // is formed on the device and only its relative change comes back (lzx_multout_change_f64).
    double xn = 0;
    if (world == 1) lzx_or_throw(lzx_lanczos_prepare_f64(hs[0], x0, k, &xn), "lzx_lanczos_prepare_f64");
    else lzx_or_throw(lzx_lanczos_prepare_f64_local(hs, world, x0, k, &xn), "lzx_lanczos_prepare_f64_local");
    std::vector<double> d, e, z, t;
    unsigned done = 0;
    while (done < k) {
      lzx_stats s1{};
      const unsigned steps = std::min(opts.adaptive_step, k - done);
      if (world == 1) lzx_or_throw(lzx_lanczos_run_steps(hs[0], steps, &s1), "lzx_lanczos_run_steps");
      else lzx_or_throw(lzx_lanczos_run_steps_local(hs, world, steps, &s1), "lzx_lanczos_run_steps_local");
      done += steps;
      st.loop_ms += s1.loop_ms; st.spmv_ms += s1.spmv_ms; st.vec_ms += s1.vec_ms; st.comm_ms += s1.comm_ms;
      st.spmv_bytes = s1.spmv_bytes;
      if (world == 1) lzx_or_throw(lzx_lanczos_fetch_f64(hs[0], done, a, b, nullptr), "lzx_lanczos_fetch_f64");
      else lzx_or_throw(lzx_lanczos_fetch_f64_local(hs, world, done, a, b, nullptr), "lzx_lanczos_fetch_f64_local");
      d.assign(a, a + done);
      e.assign(done, 0.0);
      for (unsigned i = 0; i + 1 < done; ++i) e[i] = b[i];
      z.assign(static_cast<std::size_t>(done) * done, 0.0);
      if (symtridiag_ql(static_cast<int>(done), d.data(), e.data(), z.data()) != 0)
        throw std::runtime_error("lanczosDecomp: QL iteration did not converge");
      t.assign(done, 0.0);
      for (unsigned j = 0; j < done; ++j) d[j] = std::exp(d[j]) * (xn * z[j]);
      for (unsigned i = 0; i < done; ++i) {
        double acc = 0;
        for (unsigned j = 0; j < done; ++j) acc += z[static_cast<std::size_t>(i) * done + j] * d[j];
        t[i] = acc;
      }
      double change = 1.0;
      if (world == 1) lzx_or_throw(lzx_multout_change_f64(hs[0], t.data(), done, &change), "lzx_multout_change_f64");
      else lzx_or_throw(lzx_multout_change_f64_local(hs, world, t.data(), done, &change), "lzx_multout_change_f64_local");
      adaptive.k.push_back(done);
      adaptive.rel_change.push_back(change);
      adaptive.k_used = done;
      if (!(change > opts.adaptive_tol)) { adaptive.converged = true; break; }   // also stops on NaN
    }
    iters_run = done;
    krylov_dim = done;   // what eigenDecomp / multOut / the basis fetch work with from here on
  };
  if constexpr (std::is_same<T, double>::value) {
    run(x, alpha, beta);
  } else {
    // The engine computes in fp64 (BASELINE.json north star); a float decomposition is the rounded result.
    std::vector<double> xd(x, x + n), a(k), b(k > 1 ? k - 1 : 1);
    run(xd.data(), a.data(), b.data());
    std::copy(a.begin(), a.begin() + krylov_dim, alpha);
    if (krylov_dim > 1) std::copy(b.begin(), b.begin() + (krylov_dim - 1), beta);
  }
  (void)k;
  times.setup_ms = fresh ? graph->setup_ms : 0.0;
  times.loop_ms = st.loop_ms;
  times.spmv_ms = st.spmv_ms;
  times.vec_ms = st.vec_ms;
  times.comm_ms = st.comm_ms;
  times.spmv_bytes = st.spmv_bytes;
  times.gpus = static_cast<unsigned>(world);
}

template <typename T>
void lanczosDecomp<T>::ensure_host_basis() {
  if (Q) return;
  if (basis_lost) throw std::runtime_error("lanczosDecomp: the device-resident basis was overwritten by a later decomposition before it could be brought to the host");
  if (!graph) throw std::logic_error("lanczosDecomp: no basis (free_mem() was called)");
  const auto t0 = std::chrono::steady_clock::now();
  const std::size_t n = A.get_n();
  const unsigned k = krylov_dim;
  lzx_handle *hs = graph->ranks.data();
  const int world = static_cast<int>(graph->ranks.size());
  std::vector<double> a(k), b(k > 1 ? k - 1 : 1);
  auto fetch = [&](double *Qd) {
    if (world == 1) lzx_or_throw(lzx_lanczos_fetch_f64(hs[0], k, a.data(), b.data(), Qd), "lzx_lanczos_fetch_f64");
    else lzx_or_throw(lzx_lanczos_fetch_f64_local(hs, world, k, a.data(), b.data(), Qd), "lzx_lanczos_fetch_f64_local");
  };
  std::unique_ptr<T[]> fresh(new T[n * k]);   // Q is assigned only once the fetch is complete
  if constexpr (std::is_same<T, double>::value) {
    fetch(fresh.get());
  } else {
    std::vector<double> Qd(n * k);
    fetch(Qd.data());
    for (std::size_t i = 0; i < Qd.size(); ++i) fresh[i] = static_cast<T>(Qd[i]);
  }
  Q = fresh.release();
  times.fetch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template <typename T>
void lanczosDecomp<T>::device_multout(const double *t, unsigned k, double *out) {
  if (!graph) throw std::logic_error("lanczosDecomp: the decomposition has no device-resident basis");
  lzx_handle *hs = graph->ranks.data();
  const int world = static_cast<int>(graph->ranks.size());
  if (world == 1) lzx_or_throw(lzx_multout_f64(hs[0], t, k, out), "lzx_multout_f64");
  else lzx_or_throw(lzx_multout_f64_local(hs, world, t, k, out), "lzx_multout_f64_local");
}

// Two modified Gram-Schmidt sweeps over the k stored vectors (see cu_lanczos.h).
template <typename T>
void lanczosDecomp<T>::reorthog() {
  const bool vectors = graph != nullptr || on_device_layout;   // k contiguous vectors (device layout) or row-major n x k
  ensure_host_basis();
  const std::size_t n = A.get_n();
  const unsigned k = krylov_dim;
  const std::size_t sj = vectors ? n : 1, si = vectors ? 1 : k;   // Q(i, j) = Q[j * sj + i * si]
  for (unsigned j = 0; j < k; ++j) {
    T *qj = Q + j * sj;
    for (int sweep = 0; sweep < 2; ++sweep)
      for (unsigned m = 0; m < j; ++m) {
        const T *qm = Q + m * sj;
        T dot = 0;
        for (std::size_t i = 0; i < n; ++i) dot += qj[i * si] * qm[i * si];
        for (std::size_t i = 0; i < n; ++i) qj[i * si] -= dot * qm[i * si];
      }
    T nn = 0;
    for (std::size_t i = 0; i < n; ++i) nn += qj[i * si] * qj[i * si];
    nn = std::sqrt(nn);
    for (std::size_t i = 0; i < n; ++i) qj[i * si] /= nn;
  }
}

template <typename T>
void lanczosDecomp<T>::get_ans() const {
  std::cout << "Answer vector:\n";
  for (unsigned i = 0; i < A.get_n(); ++i) std::cout << std::setprecision(20) << ans[i] << '\n';
}

// Same report as serial/lib/lanczos.cc:183-199: largest deviation, its place, absolute and relative 2-norm.
template <typename T>
void lanczosDecomp<T>::check_ans(const T *analytic_ans) const {
  const unsigned n = A.get_n();
  std::vector<T> diff(n);
  unsigned worst = 0;
  for (unsigned i = 0; i < n; ++i) {
    diff[i] = std::abs(ans[i] - analytic_ans[i]);
    if (diff[i] > diff[worst]) worst = i;
  }
  std::cout << "\nMax difference of " << diff[worst] << " found at index\n\tlanczos[" << worst << "] \t\t\t= " << ans[worst]
            << "\n\tanalytic_ans[" << worst << "] \t\t= " << analytic_ans[worst] << '\n';
  const T dn = norm(diff.data(), n);
  std::cout << "\nTotal norm of differences\t= " << dn << '\n';
  std::cout << "Relative norm of differences\t= " << dn / norm(analytic_ans, n) << '\n';
}

template class lanczosDecomp<double>;
template class lanczosDecomp<float>;
