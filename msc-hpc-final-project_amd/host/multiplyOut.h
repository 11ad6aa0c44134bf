// multiplyOut.h -- back-projection ans = ||x|| Q V e^Lambda V^T e_1 of the Lanczos approximation to e^A x.
// Drop-in for parallel-final/lib/multiplyOut.h:11-12 (host, two GEMVs) plus the on-device variant of
// parallel-mult-on-card/lib/cu_multiplyOut.cu:16-90 (there cuBLAS; here the basis already resident in HBM).
#pragma once

#include "adjMatrix.h"
#include "cu_lanczos.h"
#include "eigen.h"

// Qtrans == true : L.Q holds k contiguous vectors (device decomposition)
// Qtrans == false: L.Q is row-major n x k          (CPU decomposition)
template <typename T>
void multOut(lanczosDecomp<T> &, eigenDecomp<T> &, adjMatrix &, bool Qtrans);

// The n x k product on the GPU against the basis the device decomposition left resident; the small
// k x k part stays on the host.  Requires L to come from cuda == true (and not free_mem'ed).
template <typename T>
void cu_multOut(lanczosDecomp<T> &, eigenDecomp<T> &, adjMatrix &, bool Qtrans = true);
