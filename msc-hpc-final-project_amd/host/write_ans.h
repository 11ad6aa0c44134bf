// write_ans.h -- one answer component per line (parallel-final/lib/write_ans.h:9-16).  The reference
// asserts on open failure; this reports it on stderr instead of aborting a finished run.
#pragma once

#include <fstream>
#include <iostream>
#include <string>

#include "cu_lanczos.h"

template <typename T>
void write_ans(std::string filename, lanczosDecomp<T> &L) {
  std::ofstream fs(filename);
  if (fs.fail()) {
    std::cerr << "write_ans: cannot open " << filename << '\n';
    return;
  }
  for (unsigned i = 0; i < L.A.get_n(); ++i) fs << L.ans[i] << '\n';
}
