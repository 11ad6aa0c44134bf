// multiplyOut.h -- back-projection ans = ||x|| Q V e^Lambda V^T e_1 of the Lanczos approximation to e^A x.
// Drop-in for parallel-final/lib/multiplyOut.h:11-12 (host, two GEMVs) plus the on-device variant of
// parallel-mult-on-card/lib/cu_multiplyOut.cu:16-90 (there cuBLAS; here the basis already resident in HBM).
#pragma once

#include "adjMatrix.h"
#include "cu_lanczos.h"
#include "eigen.h"

#include <vector>

// Qtrans == true : L.Q holds k contiguous vectors (device decomposition)
// Qtrans == false: L.Q is row-major n x k          (CPU decomposition)
template <typename T>
void multOut(lanczosDecomp<T> &, eigenDecomp<T> &, adjMatrix &, bool Qtrans);

// The n x k product on the GPU against the basis the device decomposition left resident; the small
// k x k part stays on the host.  Requires L to come from cuda == true (and not free_mem'ed).
template <typename T>
void cu_multOut(lanczosDecomp<T> &, eigenDecomp<T> &, adjMatrix &, bool Qtrans = true);

// Convergence monitor (the reference's stated open problem, writeup section 11: "multOut for the first k < r
// columns").  From ONE decomposition of dimension K it evaluates the Krylov approximation
// y_k = ||x|| Q_k V_k e^{Lambda_k} V_k^T e_1 for k = step, 2*step, ... <= K, each from the leading k x k block of
// the tridiagonal matrix, and stops at the first k whose answer moved by less than `tol` (relative 2-norm) from
// the previous one.  L.ans holds y_{k_used} afterwards.  Uses the GPU-resident basis when L has one.
// (struct convergenceReport: cu_lanczos.h -- a decomposition constructed with lanczosOptions::adaptive_step carries one too,
//  from a run that STOPPED at k_used instead of evaluating a finished decomposition.)

template <typename T>
convergenceReport multOutAdaptive(lanczosDecomp<T> &L, adjMatrix &A, unsigned step, double tol, bool Qtrans);
