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
  Q = new T[n * krylov];
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
  if (engine) { lzx_destroy(engine); engine = nullptr; }
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
void lanczosDecomp<T>::cu_decompose() {
  const unsigned n = A.get_n(), k = krylov_dim;
  const auto t0 = std::chrono::steady_clock::now();
  lzx_handle h = nullptr;
  lzx_or_throw(lzx_create(&h, 0), "lzx_create");
  engine = h;
  lzx_or_throw(lzx_set_graph_csr32(h, n, 2 * A.edge_count, A.row_offset, A.col_idx), "lzx_set_graph_csr32");
  lzx_graph_info gi;
  lzx_or_throw(lzx_get_graph_info(h, &gi), "lzx_get_graph_info");
  const auto t1 = std::chrono::steady_clock::now();
  std::cout << "\nUsing " << (gi.sell_padded + gi.n) * 4 + (static_cast<std::uint64_t>(k) + 3) * gi.n * 8
            << " bytes of HBM for the reshaped graph and " << k << " resident Lanczos vectors\n";

  lzx_stats st;
  if constexpr (std::is_same<T, double>::value) {
    lzx_or_throw(lzx_lanczos_f64(h, x, k, alpha, beta, Q, nullptr, &st), "lzx_lanczos_f64");
  } else {
    // The engine computes in fp64 (BASELINE.json north star); a float decomposition is the rounded result.
    std::vector<double> xd(x, x + n), a(k), b(k > 1 ? k - 1 : 1), Qd(static_cast<std::size_t>(n) * k);
    lzx_or_throw(lzx_lanczos_f64(h, xd.data(), k, a.data(), b.data(), Qd.data(), nullptr, &st), "lzx_lanczos_f64");
    std::copy(a.begin(), a.end(), alpha);
    if (k > 1) std::copy(b.begin(), b.begin() + (k - 1), beta);
    for (std::size_t i = 0; i < Qd.size(); ++i) Q[i] = static_cast<T>(Qd[i]);
  }
  times.setup_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  times.loop_ms = st.loop_ms;
  times.spmv_ms = st.spmv_ms;
  times.vec_ms = st.vec_ms;
  times.spmv_bytes = st.spmv_bytes;
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
