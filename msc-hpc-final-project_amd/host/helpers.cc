#include "helpers.h"

#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int parseArguments(int argc, char **argv, std::string &filename, unsigned &krylov_dim, bool &verbose, unsigned &n,
                   unsigned &bar_deg, unsigned &E) {
  int c;
  optind = 1;
  while ((c = getopt(argc, argv, "k:f:b:n:e:v")) != -1) {
    switch (c) {
      case 'f': filename = optarg; break;
      case 'k': krylov_dim = static_cast<unsigned>(std::atoi(optarg)); break;
      case 'b': bar_deg = static_cast<unsigned>(std::atoi(optarg)); break;
      case 'n': n = static_cast<unsigned>(std::atoi(optarg)); break;
      case 'e': E = static_cast<unsigned>(std::atoi(optarg)); break;
      case 'v': verbose = true; break;
      default: std::fprintf(stderr, "Invalid option given\n"); return -1;
    }
  }
  return 0;
}

// relative_error = ||a - b||_2 / ||b||_2, max_entry = index of the largest |a_i - b_i|.
template <typename T>
void diff_arrays(const T *const a, const T *const b, const unsigned n, T &relative_error, unsigned &max_entry) {
  T d2 = 0, b2 = 0, worst = -1;
  max_entry = 0;
  for (unsigned i = 0; i < n; ++i) {
    const T d = std::abs(a[i] - b[i]);
    if (d > worst) { worst = d; max_entry = i; }
    d2 += d * d;
    b2 += b[i] * b[i];
  }
  relative_error = std::sqrt(d2) / std::sqrt(b2);
}

template <typename T>
void my_exp_func(T &a) { a = std::exp(a); }

template void diff_arrays<double>(const double *const, const double *const, const unsigned, double &, unsigned &);
template void diff_arrays<float>(const float *const, const float *const, const unsigned, float &, unsigned &);
template void my_exp_func<double>(double &);
template void my_exp_func<float>(float &);
