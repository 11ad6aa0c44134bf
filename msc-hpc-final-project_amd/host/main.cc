// final -- command-line driver with the surface of parallel-final's `./final -f NAME -k K`
// (parallel-final/main.cu:34-162): load or generate a graph, run the Lanczos e^A x once on the CPU and
// once on the MI355X, print the TIMING and ERROR CHECKING blocks, write the answer vector.
//
//   -f NAME   "../data/NAME/NAME.mtx" as in the reference; a NAME containing '/' or ending in ".mtx"
//             is taken as a path (serial/main.cc:12,34 takes a path too)
//   -k K      Krylov dimension
//   -n N -e E generate a seeded G(N, E) graph instead of reading one (the reference parses these flags
//             but hard-wires file input, main.cu:57)
//   -n N -b M generate a seeded Barabasi-Albert graph of minimum degree M
//   -v        verbose: print the answer vector
// Environment: FINAL_ADAPTIVE_STEP=S [FINAL_ADAPTIVE_TOL=t, default 1e-10]: the device run advances in chunks of S iterations and
// stops once the answer moved by less than t (K is then the upper limit; k_used is printed); FINAL_ARNOLDI=E: both runs
// re-orthogonalise every E iterations (serial/lib/lanczos.cc:58-132; the reference's constant is 2);
// FINAL_SKIP_SERIAL=1 skips the CPU run (large graphs), FINAL_DEVICE_MULTOUT=1 uses the
// on-device back-projection (parallel-mult-on-card's cu_multOut) for the GPU column; LZX_DEVICES=all | N | a,b,c
// spreads the device run over several GPUs from this one process (parallel-two-cards' model, any number of cards);
// LZX_NO_CSR_CACHE=1, LZX_HOST_INGEST=1, LZX_PARSE_THREADS=T steer the loader (host/adjMatrix.h).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "adjMatrix.h"
#include "check_ans.h"
#include "cu_lanczos.h"
#include "eigen.h"
#include "helpers.h"
#include "multiplyOut.h"
#include "write_ans.h"

namespace {
constexpr int kRule = 81;
void rule() { std::cout << std::setfill('~') << std::setw(kRule) << '\n' << std::setfill(' '); }
void row(const char *what, double cpu, double gpu, int w) {
  std::cout << std::setw(w) << std::left << what << std::right << std::setw(w) << cpu << std::setw(w) << gpu << std::setw(w)
            << (gpu > 0 ? cpu / gpu : 0.0) << "\n\n";
}
bool env_on(const char *name) {
  const char *v = std::getenv(name);
  return v && *v && *v != '0';
}
}  // namespace

int main(int argc, char **argv) {
  unsigned n = 10000, deg = 0, edges = 0, krylov_dim = 100;
  const int width = 17;
  bool verbose = false;
  std::string filename;

  if (parseArguments(argc, argv, filename, krylov_dim, verbose, n, deg, edges) != 0) return 2;

  stopwatch build;
  adjMatrix A;
  std::string ans_path;
  if (!filename.empty()) {
    const bool is_path = filename.find('/') != std::string::npos ||
                         (filename.size() > 4 && filename.compare(filename.size() - 4, 4, ".mtx") == 0);
    const std::string filepath = is_path ? filename : "../data/" + filename + "/" + filename + ".mtx";
    std::cout << "Going to open file: " << filepath << std::endl;
    {
      std::ifstream fs(filepath);
      if (fs.fail()) {
        std::cerr << "File opening failed: " << filepath << '\n';
        return 1;
      }
    }
    // the reference's `fs >> n >> n >> edges; adjMatrix(n, edges, fs)` (main.cu:62-63) with the path known: binary
    // side-car cache, several parser threads, device ingest when a GPU is present
    A = adjMatrix::load(filepath);
    ans_path = is_path ? filepath + ".ans" + std::to_string(krylov_dim) + ".txt"
                       : "../data/" + filename + "/ans" + std::to_string(krylov_dim) + ".txt";
  } else if (deg > 0) {
    A = adjMatrix(n, deg, 'b');
    ans_path = "ans" + std::to_string(krylov_dim) + ".txt";
  } else {
    if (edges == 0) edges = n * 10;  // the reference's default (main.cu:38)
    A = adjMatrix(n, edges);
    ans_path = "ans" + std::to_string(krylov_dim) + ".txt";
  }
  n = A.get_n();
  edges = A.get_edges();
  krylov_dim = std::max(1u, std::min(krylov_dim, n - 1));  // serial/main.cc:64
  std::cout << "\nTime elapsed to build adjacency matrix with n = " << n << " edges = " << edges << ":\n\t" << build.seconds()
            << " seconds\n\n";
  {
    const adjMatrix::loadReport &lr = A.load_report();
    if (lr.from_cache) std::cout << "\t(binary CSR cache read in " << lr.cache_s << " s)\n\n";
    else if (!filename.empty())
      std::cout << "\t(text parse " << lr.parse_s << " s on " << lr.threads << " thread(s); CSR build " << lr.build_s << " s "
                << (lr.on_device ? "on the GPU: graph already resident" : "on the host") << ")\n\n";
  }
  std::cout << "Running Lanczos algorithm for krylov_dim " << krylov_dim << "\n\n";

  std::vector<double> x(n, 1.0);
  const bool skip_serial = env_on("FINAL_SKIP_SERIAL");

  lanczosOptions host_opt, dev_opt;
  if (const char *v = std::getenv("FINAL_ARNOLDI")) host_opt.arnoldi_every = dev_opt.arnoldi_every = static_cast<unsigned>(std::atoi(v));
  if (const char *v = std::getenv("FINAL_ADAPTIVE_STEP")) dev_opt.adaptive_step = static_cast<unsigned>(std::atoi(v));
  if (const char *v = std::getenv("FINAL_ADAPTIVE_TOL")) dev_opt.adaptive_tol = std::atof(v);
  dev_opt.reference_order = env_on("FINAL_REFERENCE_ORDER");   // the device run with the CPU run's reduction orders: identical alpha / beta / Q

  // ---- CPU ----
  double cpu_lanczos = 0, cpu_mult = 0, cpu_whole = 0;
  std::unique_ptr<lanczosDecomp<double>> L;
  if (!skip_serial) {
    stopwatch whole, t;
    L = std::make_unique<lanczosDecomp<double>>(A, krylov_dim, x.data(), false, host_opt);
    cpu_lanczos = t.seconds();
    eigenDecomp<double> E(*L);
    stopwatch tm;
    multOut(*L, E, A, false);
    cpu_mult = tm.seconds();
    cpu_whole = whole.seconds();
    L->free_mem();  // keep only the answer (main.cu:106)
  }

  // ---- MI355X ----
  stopwatch gwhole, gt;
  lanczosDecomp<double> cu_L(A, krylov_dim, x.data(), true, dev_opt);
  const double gpu_lanczos = gt.seconds();
  if (dev_opt.adaptive_step) {
    const convergenceReport &rep = cu_L.convergence();
    std::cout << "adaptive run: k_used = " << rep.k_used << " of at most " << krylov_dim << " (" << cu_L.iterations_run()
              << " SpMVs run, " << (rep.converged ? "converged" : "NOT converged") << " at tolerance " << dev_opt.adaptive_tol
              << "); relative changes:";
    for (double c : rep.rel_change) std::cout << ' ' << c;
    std::cout << "\n\n";
  }
  eigenDecomp<double> cu_E(cu_L);
  stopwatch gm;
  if (env_on("FINAL_DEVICE_MULTOUT")) cu_multOut(cu_L, cu_E, A, true);
  else multOut(cu_L, cu_E, A, true);
  const double gpu_mult = gm.seconds();
  const double gpu_whole = gwhole.seconds();

  rule();
  std::cout << "TIMING\n";
  rule();
  std::cout << std::setw(2 * width) << "Serial" << std::setw(width) << "MI355X" << std::setw(width) << "Speedup" << '\n';
  rule();
  row("Lanczos", cpu_lanczos, gpu_lanczos, width);
  row("Multiply Out", cpu_mult, gpu_mult, width);
  row("Entire algorithm", cpu_whole, gpu_whole, width);
  const lanczosTimings &tm = cu_L.timings();
  std::cout << "device loop only: " << tm.loop_ms * 1e-3 << " s (" << cu_L.iterations_run() / (tm.loop_ms * 1e-3) << " Lanczos iterations/s); "
            << "SpMV " << tm.spmv_ms / cu_L.iterations_run() << " ms each = " << (tm.spmv_ms > 0 ? tm.spmv_bytes / (tm.spmv_ms / cu_L.iterations_run()) * 1e-6 : 0.0)
            << " GB/s of algorithmic bytes on " << tm.gpus << " GPU handle(s); graph upload + reshaping inside the constructor "
            << tm.setup_ms * 1e-3 << " s; basis download " << tm.fetch_ms * 1e-3 << " s\n";

  rule();
  std::cout << "ERROR CHECKING\n";
  rule();
  if (!skip_serial) check_ans(*L, cu_L);
  else std::cout << "(serial run skipped)\n";

  if (verbose) cu_L.get_ans();
  if (skip_serial) write_ans(ans_path, cu_L);
  else write_ans(ans_path, *L);  // the reference writes the serial answer (main.cu:158-159)
  rule();
  return 0;
}
