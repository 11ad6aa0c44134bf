#include "SPMV.h"

// out[i] = sum of in[col] over row i's columns, added one at a time in ascending column order --
// exactly the left-to-right sum the reference forms (it accumulates through out[i] itself).
template <typename T>
void spMV(const adjMatrix &A, const T *const in, T *const out) {
  for (unsigned i = 0; i < A.n; ++i) {
    T acc = 0;
    const unsigned end = A.row_offset[i + 1];
    for (unsigned j = A.row_offset[i]; j < end; ++j) acc += in[A.col_idx[j]];
    out[i] = acc;
  }
}

template void spMV<double>(const adjMatrix &, const double *const, double *const);
template void spMV<float>(const adjMatrix &, const float *const, float *const);
