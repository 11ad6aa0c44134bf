// check_ans.h -- compare the answer vectors of two decompositions (parallel-final/lib/check_ans.h:6-7).
#pragma once

#include "cu_lanczos.h"

// Prints the reference's report (largest deviation and where, absolute and relative 2-norm of the
// difference, parallel-final/lib/check_ans.cu:12-29) plus the relative infinity-norm
// max|a - b| / max|b| that BASELINE.json's tolerance (1e-10) is stated in.
template <typename T, typename U>
void check_ans(lanczosDecomp<T> &, lanczosDecomp<U> &);

// The same numbers without printing: {max abs diff, rel 2-norm, rel inf-norm}.
struct ansDiff { double max_abs, rel_two, rel_inf; };
ansDiff diff_ans(const double *a, const double *b, unsigned n);
