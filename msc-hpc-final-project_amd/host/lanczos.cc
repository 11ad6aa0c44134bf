// lanczos.cc -- CPU and device paths of lanczosDecomp<T> (see cu_lanczos.h).
#include <algorithm>
#include <chrono>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "SPMV.h"
#include "cu_lanczos.h"
#include "lzx.h"

template <typename T>
lanczosDecomp<T>::lanczosDecomp(adjMatrix &adj, const unsigned krylov, T *starting_vec, bool cuda)
    : A{adj}, krylov_dim{krylov} {
  if (krylov == 0) throw std::invalid_argument("lanczosDecomp: krylov dimension must be positive");
  const std::size_t n = A.get_n();
  alpha = new T[krylov];
  beta = new T[krylov > 1 ? krylov - 1 : 1];
  if (!cuda) Q = new T[n * krylov];   // the device path keeps the basis in HBM (ensure_host_basis)
  x = new T[n];
  ans = new T[n];
  x_norm = norm(starting_vec, A.get_n());
  std::copy(starting_vec, starting_vec + n, x);
  if (cuda) cu_decompose();
  else decompose();
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

namespace {
void lzx_or_throw(int rc, const char *what) {
  if (rc != LZX_OK) throw std::runtime_error(std::string(what) + ": " + lzx_last_error());
}
}  // namespace

template <typename T>
void lanczosDecomp<T>::evict_cb(void *self) {
  auto *L = static_cast<lanczosDecomp<T> *>(self);
  try { L->ensure_host_basis(); } catch (...) {}
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

  lzx_stats st;
  auto run = [&](const double *x0, double *a, double *b) {
    if (world == 1) lzx_or_throw(lzx_lanczos_f64(hs[0], x0, k, a, b, nullptr, nullptr, &st), "lzx_lanczos_f64");
    else lzx_or_throw(lzx_lanczos_f64_local(hs, world, x0, k, a, b, nullptr, nullptr, &st), "lzx_lanczos_f64_local");
  };
  if constexpr (std::is_same<T, double>::value) {
    run(x, alpha, beta);
  } else {
    // The engine computes in fp64 (BASELINE.json north star); a float decomposition is the rounded result.
    std::vector<double> xd(x, x + n), a(k), b(k > 1 ? k - 1 : 1);
    run(xd.data(), a.data(), b.data());
    std::copy(a.begin(), a.end(), alpha);
    if (k > 1) std::copy(b.begin(), b.begin() + (k - 1), beta);
  }
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
  if (Q || !graph) return;
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
  Q = new T[n * k];
  if constexpr (std::is_same<T, double>::value) {
    fetch(Q);
  } else {
    std::vector<double> Qd(n * k);
    fetch(Qd.data());
    for (std::size_t i = 0; i < Qd.size(); ++i) Q[i] = static_cast<T>(Qd[i]);
  }
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
