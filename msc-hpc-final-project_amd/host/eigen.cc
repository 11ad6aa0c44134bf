#include "eigen.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <vector>

// Implicit QL with Wilkinson shifts (the classical EISPACK tql2 scheme): for each l, chase the
// sub-diagonal e[l] to zero with plane rotations accumulated into z.
int symtridiag_ql(int n, double *d, double *e_in, double *z) {
  std::vector<double> e(n, 0.0);
  for (int i = 0; i + 1 < n; ++i) e[i] = e_in[i];
  std::fill(z, z + static_cast<std::size_t>(n) * n, 0.0);
  for (int i = 0; i < n; ++i) z[static_cast<std::size_t>(i) * n + i] = 1.0;
  const double eps = std::numeric_limits<double>::epsilon();

  for (int l = 0; l < n; ++l) {
    for (int sweeps = 0;; ++sweeps) {
      int m = l;
      for (; m + 1 < n; ++m) {
        const double scale = std::abs(d[m]) + std::abs(d[m + 1]);
        if (std::abs(e[m]) <= eps * scale) break;
      }
      if (m == l) break;
      if (sweeps == 60) return -1;

      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = std::hypot(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      for (; i >= l; --i) {
        double f = s * e[i];
        const double b = c * e[i];
        r = std::hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) {  // underflow: deflate and restart this l
          d[i + 1] -= p;
          e[m] = 0.0;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
        for (int row = 0; row < n; ++row) {
          double *zr = z + static_cast<std::size_t>(row) * n;
          f = zr[i + 1];
          zr[i + 1] = s * zr[i] + c * f;
          zr[i] = c * zr[i] - s * f;
        }
      }
      if (r == 0.0 && i >= l) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
  }

  // ascending order, vectors permuted along
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
  std::vector<double> ds(n), zs(static_cast<std::size_t>(n) * n);
  for (int j = 0; j < n; ++j) {
    ds[j] = d[order[j]];
    for (int i = 0; i < n; ++i) zs[static_cast<std::size_t>(i) * n + j] = z[static_cast<std::size_t>(i) * n + order[j]];
  }
  std::copy(ds.begin(), ds.end(), d);
  std::copy(zs.begin(), zs.end(), z);
  return 0;
}

template <typename T>
eigenDecomp<T>::eigenDecomp(lanczosDecomp<T> &_L)
    : eigenvalues(new T[_L.krylov_dim]), eigenvectors(new T[static_cast<std::size_t>(_L.krylov_dim) * _L.krylov_dim]), L{_L} {
  for (unsigned i = 0; i < L.get_krylov(); ++i) eigenvalues[i] = L.alpha[i];
  decompose();
}

template <typename T>
eigenDecomp<T>::~eigenDecomp() {
  delete[] eigenvalues;
  delete[] eigenvectors;
}

template <typename T>
void eigenDecomp<T>::decompose() {
  const int k = static_cast<int>(L.krylov_dim);
  std::vector<double> d(eigenvalues, eigenvalues + k), e(k > 1 ? k - 1 : 1, 0.0), z(static_cast<std::size_t>(k) * k);
  for (int i = 0; i + 1 < k; ++i) e[i] = L.beta[i];
  if (symtridiag_ql(k, d.data(), e.data(), z.data()) != 0)
    throw std::runtime_error("eigenDecomp: QL iteration did not converge");
  for (int i = 0; i < k; ++i) eigenvalues[i] = static_cast<T>(d[i]);
  for (std::size_t i = 0; i < z.size(); ++i) eigenvectors[i] = static_cast<T>(z[i]);
}

template class eigenDecomp<double>;
template class eigenDecomp<float>;
