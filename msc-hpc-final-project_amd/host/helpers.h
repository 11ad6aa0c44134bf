// helpers.h -- command line and small utilities (parallel-final/lib/helpers.h, minus the cuBLAS/CBLAS
// float overloads, which exist there only so that templated tests can call D-named BLAS with floats).
#pragma once

#include <chrono>
#include <string>

// getopt string "k:f:b:n:e:v" as parallel-final/lib/helpers.cu:35; returns -1 on an unknown flag.
int parseArguments(int argc, char **argv, std::string &filename, unsigned &krylov_dim, bool &verbose, unsigned &n,
                   unsigned &bar_deg, unsigned &E);

template <typename T>
void diff_arrays(const T *const a, const T *const b, const unsigned n, T &relative_error, unsigned &max_entry);

template <typename T>
void my_exp_func(T &a);

// Wall-clock stopwatch in seconds (stands in for the cudaEvent pair of helpers.cu:14-29: the device
// path reports its own HIP-event timings through lanczosDecomp::timings()).
struct stopwatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
