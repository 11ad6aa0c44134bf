#include "check_ans.h"

#include <algorithm>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <vector>

ansDiff diff_ans(const double *a, const double *b, unsigned n) {
  double mx = 0, d2 = 0, b2 = 0, bmax = 0;
  for (unsigned i = 0; i < n; ++i) {
    const double d = std::abs(a[i] - b[i]);
    mx = std::max(mx, d);
    d2 += d * d;
    b2 += b[i] * b[i];
    bmax = std::max(bmax, std::abs(b[i]));
  }
  return {mx, std::sqrt(d2) / std::sqrt(b2), mx / bmax};
}

template <typename T, typename U>
void check_ans(lanczosDecomp<T> &L1, lanczosDecomp<U> &L2) {
  using W = decltype(T() + U());
  const unsigned n = L1.A.get_n();
  std::vector<W> diff(n);
  unsigned at = 0;
  W bmax = 0;
  for (unsigned i = 0; i < n; ++i) {
    diff[i] = std::abs(static_cast<W>(L1.ans[i]) - static_cast<W>(L2.ans[i]));
    if (diff[i] > diff[at]) at = i;
    bmax = std::max<W>(bmax, std::abs(static_cast<W>(L1.ans[i])));
  }
  std::cout << "\nMax difference of " << diff[at] << " (Relative difference: " << diff[at] / L2.ans[at] << ") found at index:\n"
            << std::setw(15) << "serial_ans[" << at << "] = " << std::setprecision(10) << std::setw(15) << L1.ans[at] << "\n"
            << std::setw(15) << "gpu_ans[" << at << "] = " << std::setprecision(10) << std::setw(15) << L2.ans[at] << '\n'
            << std::endl;
  const W dn = norm(diff.data(), n);
  std::vector<W> ref(L2.ans, L2.ans + n);
  std::cout << std::setw(30) << std::left << "Total norm of differences" << "=" << std::right << std::setprecision(20) << std::setw(30) << dn << std::endl;
  std::cout << std::setw(30) << std::left << "Relative norm of differences" << "=" << std::right << std::setprecision(20) << std::setw(30)
            << dn / norm(ref.data(), n) << std::endl;
  std::cout << std::setw(30) << std::left << "Relative inf-norm (vs serial)" << "=" << std::right << std::setprecision(20) << std::setw(30)
            << diff[at] / bmax << std::endl;
}

template void check_ans(lanczosDecomp<float> &, lanczosDecomp<float> &);
template void check_ans(lanczosDecomp<double> &, lanczosDecomp<float> &);
template void check_ans(lanczosDecomp<float> &, lanczosDecomp<double> &);
template void check_ans(lanczosDecomp<double> &, lanczosDecomp<double> &);
