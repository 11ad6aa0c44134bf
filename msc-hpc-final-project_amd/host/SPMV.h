// SPMV.h -- CPU sparse matrix-vector product for a pattern-only adjacency matrix
// (surface of parallel-final/lib/SPMV.h:6-7; arithmetic of serial/lib/SPMV.cc:19-28).
#pragma once

#include "adjMatrix.h"

template <typename T>
void spMV(const adjMatrix &A, const T *const in, T *const out);
