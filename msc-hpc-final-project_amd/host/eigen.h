// eigen.h -- `eigenDecomp<T>`: eigen-decomposition T_k = V diag(lambda) V^T of the k x k symmetric
// tridiagonal Lanczos matrix.  Drop-in for parallel-final/lib/eigen.h:10-39, which calls
// LAPACKE_dstevd (parallel-final/lib/eigen.cu:17-21).  No LAPACK is assumed on an MI355X node, so the
// solver here is a self-contained implicit-shift QL iteration (k is 20..200: microseconds either way).
// Output convention kept from LAPACK_ROW_MAJOR/'V': eigenvalues ascending, eigenvectors[i * k + j] is
// component i of eigenvector j.  Unlike dstevd, L.beta is left untouched.
#pragma once

#include "cu_lanczos.h"

template <typename T>
class eigenDecomp {
 public:
  eigenDecomp() = delete;
  explicit eigenDecomp(lanczosDecomp<T> &_L);
  eigenDecomp(eigenDecomp<T> &) = delete;
  eigenDecomp &operator=(eigenDecomp<T> &) = delete;
  ~eigenDecomp();

  const T *values() const { return eigenvalues; }
  const T *vectors() const { return eigenvectors; }

  template <typename U> friend void multOut(lanczosDecomp<U> &, eigenDecomp<U> &, adjMatrix &, bool);
  template <typename U> friend void cu_multOut(lanczosDecomp<U> &, eigenDecomp<U> &, adjMatrix &, bool);

 private:
  T *eigenvalues;
  T *eigenvectors;
  lanczosDecomp<T> &L;
  void decompose();
};

// Symmetric tridiagonal eigen-solver: d[n] diagonal -> eigenvalues (ascending), e[n-1] sub-diagonal
// (destroyed), z[n*n] -> eigenvectors in columns (z[i*n + j]).  Returns 0, or -1 if an eigenvalue needs
// more than 60 sweeps.
int symtridiag_ql(int n, double *d, double *e, double *z);
