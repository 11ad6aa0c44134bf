#include "multiplyOut.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <stdexcept>
#include <vector>

#include "lzx.h"

namespace {
// t = V (e^lambda .* ||x|| V[0,:])  -- multiplyOut.cu:30-40 of the reference: exponentiate the Ritz
// values in place, weight by the first eigenvector components, one k x k GEMV.
template <typename T>
std::vector<T> small_part(T *lambda, const T *V, unsigned k, T x_norm) {
  for (unsigned j = 0; j < k; ++j) lambda[j] = std::exp(lambda[j]);
  for (unsigned j = 0; j < k; ++j) lambda[j] *= x_norm * V[j];
  std::vector<T> t(k);
  for (unsigned i = 0; i < k; ++i) {
    T s = 0;
    for (unsigned j = 0; j < k; ++j) s += V[static_cast<std::size_t>(i) * k + j] * lambda[j];
    t[i] = s;
  }
  return t;
}
}  // namespace

template <typename T>
void multOut(lanczosDecomp<T> &L, eigenDecomp<T> &E, adjMatrix &, bool Qtrans) {
  const std::size_t n = L.get_n();
  const unsigned k = L.get_krylov();
  const std::vector<T> t = small_part(E.eigenvalues, E.eigenvectors, k, L.x_norm);
  if (Qtrans) {
    L.ensure_host_basis();   // a device decomposition downloads its basis here, on first use
    // ans = sum_j t_j q_j over contiguous vectors: stream each vector once
    for (std::size_t i = 0; i < n; ++i) L.ans[i] = 0;
    for (unsigned j = 0; j < k; ++j) {
      const T *q = L.Q + static_cast<std::size_t>(j) * n;
      const T tj = t[j];
      for (std::size_t i = 0; i < n; ++i) L.ans[i] += tj * q[i];
    }
  } else {
    if (!L.Q) throw std::logic_error("multOut: Qtrans == false asks for a row-major basis, which a device decomposition does not produce (pass Qtrans = true)");
    for (std::size_t i = 0; i < n; ++i) {
      const T *row = L.Q + i * k;
      T s = 0;
      for (unsigned j = 0; j < k; ++j) s += row[j] * t[j];
      L.ans[i] = s;
    }
  }
}

template <typename T>
void cu_multOut(lanczosDecomp<T> &L, eigenDecomp<T> &E, adjMatrix &, bool) {
  if (!L.on_device()) throw std::logic_error("cu_multOut: the decomposition has no device-resident basis");
  const unsigned k = L.get_krylov();
  const std::vector<T> t = small_part(E.eigenvalues, E.eigenvectors, k, L.x_norm);
  std::vector<double> td(t.begin(), t.end()), out(L.get_n());
  L.device_multout(td.data(), k, out.data());
  for (std::size_t i = 0; i < out.size(); ++i) L.ans[i] = static_cast<T>(out[i]);
}

template <typename T>
convergenceReport multOutAdaptive(lanczosDecomp<T> &L, adjMatrix &, unsigned step, double tol, bool Qtrans) {
  const std::size_t n = L.get_n();
  const unsigned K = L.get_krylov();
  if (step == 0) step = 5;
  convergenceReport rep;
  std::vector<double> prev, cur(n), d, e, z, t;
  for (unsigned k = std::min(step, K);; k = std::min(k + step, K)) {
    // leading k x k block of T: eigen-decomposition, then t = V (e^lambda .* ||x|| V[0,:])
    d.assign(L.alpha, L.alpha + k);
    e.assign(k, 0.0);
    for (unsigned i = 0; i + 1 < k; ++i) e[i] = L.beta[i];
    z.assign(static_cast<std::size_t>(k) * k, 0.0);
    if (symtridiag_ql(static_cast<int>(k), d.data(), e.data(), z.data()) != 0)
      throw std::runtime_error("multOutAdaptive: QL iteration did not converge");
    t.assign(k, 0.0);
    for (unsigned j = 0; j < k; ++j) d[j] = std::exp(d[j]) * (static_cast<double>(L.x_norm) * z[j]);
    for (unsigned i = 0; i < k; ++i) {
      double s = 0;
      for (unsigned j = 0; j < k; ++j) s += z[static_cast<std::size_t>(i) * k + j] * d[j];
      t[i] = s;
    }
    // y_k = Q_k t
    if (L.on_device()) {
      L.device_multout(t.data(), k, cur.data());
    } else if (Qtrans) {
      L.ensure_host_basis();
      std::fill(cur.begin(), cur.end(), 0.0);
      for (unsigned j = 0; j < k; ++j) {
        const T *q = L.Q + static_cast<std::size_t>(j) * n;
        for (std::size_t i = 0; i < n; ++i) cur[i] += t[j] * q[i];
      }
    } else {
      for (std::size_t i = 0; i < n; ++i) {
        const T *row = L.Q + i * K;
        double s = 0;
        for (unsigned j = 0; j < k; ++j) s += row[j] * t[j];
        cur[i] = s;
      }
    }
    double change = 1.0;
    if (!prev.empty()) {
      double d2 = 0, y2 = 0;
      for (std::size_t i = 0; i < n; ++i) {
        const double df = cur[i] - prev[i];
        d2 += df * df;
        y2 += cur[i] * cur[i];
      }
      change = std::sqrt(d2) / std::sqrt(y2);
    }
    rep.k.push_back(k);
    rep.rel_change.push_back(change);
    rep.k_used = k;
    prev = cur;
    if (!(change > tol)) { rep.converged = true; break; }   // also stops on NaN
    if (k == K) break;
  }
  for (std::size_t i = 0; i < n; ++i) L.ans[i] = static_cast<T>(prev[i]);
  return rep;
}

template void multOut(lanczosDecomp<double> &, eigenDecomp<double> &, adjMatrix &, bool);
template void multOut(lanczosDecomp<float> &, eigenDecomp<float> &, adjMatrix &, bool);
template void cu_multOut(lanczosDecomp<double> &, eigenDecomp<double> &, adjMatrix &, bool);
template void cu_multOut(lanczosDecomp<float> &, eigenDecomp<float> &, adjMatrix &, bool);
template convergenceReport multOutAdaptive(lanczosDecomp<double> &, adjMatrix &, unsigned, double, bool);
template convergenceReport multOutAdaptive(lanczosDecomp<float> &, adjMatrix &, unsigned, double, bool);
