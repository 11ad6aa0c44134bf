// adjMatrix.h -- host container of an undirected, unweighted graph as pattern-only CSR.
//
// Drop-in for the reference's `class adjMatrix` (parallel-final/lib/adjMatrix.h:19-117): same
// constructors, accessors and friend hooks, so code written against the reference compiles against
// this.  What differs underneath:
//   * the file constructor builds the CSR by sorting 64-bit (row, col) keys instead of inserting
//     2E edges into a std::set (parallel-final/lib/adjMatrix.cc:21-46; 25-55 s per 10-35 M edges in
//     the reference's own logs) -- same resulting arrays;
//   * row_offset[0] and the offsets of trailing empty rows are always written (the reference leaves
//     them to whatever the heap held);
//   * the random generators are seeded and reproducible (the reference seeds std::random_device,
//     parallel-final/lib/make_graph.cc:23-24,61-62);
//   * ingest: the text is parsed by several threads and, when a GPU is present, symmetrised,
//     sorted and de-duplicated there (lzx_set_graph_edges) -- the graph then already sits reshaped in HBM when a
//     device decomposition asks for it; adjMatrix::load(path) adds a binary side-car cache of the CSR keyed by the
//     text file's size and modification time.
#pragma once

#include <cstdint>
#include <fstream>
#include <iosfwd>
#include <memory>
#include <string>
#include <vector>

#include "device_graph.h"

template <typename T> class eigenDecomp;
template <typename T> class lanczosDecomp;

class adjMatrix {
 public:
  adjMatrix() = default;
  // `f` is positioned just behind the "n n E" header line (parallel-final/main.cu:62-63):
  // E lines "col row", 1-indexed, one per undirected edge.
  adjMatrix(unsigned N, unsigned E, std::ifstream &f);
  // 'b': Barabasi-Albert graph, every new vertex attaches to m earlier ones.
  adjMatrix(unsigned N, unsigned m, char c);
  // G(N, E) uniform random graph.
  adjMatrix(unsigned N, unsigned E);
  // The file constructor with the path known: "n n E" header read here, binary CSR side-car `<path>.lzxcsr` used when
  // it matches the text file (size + mtime) and written after a parse (best effort; LZX_NO_CSR_CACHE=1 disables both).
  static adjMatrix load(const std::string &path);
  // R-MAT graph (not in the reference; the benchmark family of BASELINE.json).
  static adjMatrix rmat(unsigned scale, unsigned N, std::uint64_t draws, std::uint64_t seed,
                        double a = 0.57, double b = 0.19, double c = 0.19);

  adjMatrix(adjMatrix &&rhs) noexcept { steal(rhs); }
  adjMatrix &operator=(adjMatrix &&rhs) noexcept {
    if (this != &rhs) { release(); steal(rhs); }
    return *this;
  }
  adjMatrix(const adjMatrix &) = delete;  // the reference's "copy" is a shallow alias that double-frees
  adjMatrix &operator=(const adjMatrix &) = delete;
  ~adjMatrix() { release(); }

  unsigned get_n() const { return n; }
  unsigned get_edges() const { return edge_count; }
  unsigned get_nnz() const { return row_offset ? row_offset[n] : 0; }   // stored entries (2 E, less one per self loop)
  // How the last load went (seconds): text parse, CSR build (device ingest or host sort), cache read; and which path.
  struct loadReport { double parse_s = 0, build_s = 0, cache_s = 0; bool from_cache = false, on_device = false; unsigned threads = 1; };
  const loadReport &load_report() const { return report; }
  // The graph on the GPU(s) (lzx_host_devices()): uploaded and reshaped on first use, kept for later decompositions.
  // Throws std::runtime_error when there is no usable GPU or the hand-over fails.
  std::shared_ptr<deviceGraph> device_graph() const;
  void set_seed(std::uint64_t s) { seed = s; }

  // Writes "<dir>/<type>n<N>e<E>": header "n n E", then "col row" (1-indexed, col > row) per edge.
  std::string write_matrix_to_file(const std::string &dir = "../data/generated/") const;
  void print_full() const;

  template <typename T> friend void spMV(const adjMatrix &, const T *const, T *const);
  friend std::ostream &operator<<(std::ostream &, const adjMatrix &);
  template <typename T> friend void multOut(lanczosDecomp<T> &, eigenDecomp<T> &, adjMatrix &, bool);
  template <typename T> friend class lanczosDecomp;

 private:
  unsigned *row_offset = nullptr;  // [n + 1]
  unsigned *col_idx = nullptr;     // [2 * edge_count], ascending within a row
  unsigned n = 0;
  unsigned edge_count = 0;         // undirected edges actually stored
  unsigned barabasi_degree = 0;
  char matrix_type = 'f';
  std::uint64_t seed = 1234;
  mutable std::shared_ptr<deviceGraph> dev;   // device-resident form, once somebody asked for it (or the ingest built it)
  loadReport report;

  void populate_sparse_matrix(std::ifstream &f);
  void build_from_text(const std::string &text);
  bool ingest_on_device(const std::vector<unsigned> &src, const std::vector<unsigned> &dst);
  void generate_sparse_matrix(char c);
  void random_adj();
  void barabasi(unsigned m);
  // Sort + de-duplicate directed keys (row << 32 | col) and emit the CSR arrays.
  void csr_from_keys(std::vector<std::uint64_t> &keys);
  void release();
  void steal(adjMatrix &rhs);
};
