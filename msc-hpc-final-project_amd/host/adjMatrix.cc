// adjMatrix.cc -- loader and seeded generators for the adjMatrix drop-in (see adjMatrix.h).
#include "adjMatrix.h"

#include <algorithm>
#include <cassert>
#include <iostream>
#include <iterator>
#include <stdexcept>

namespace {
// SplitMix64 as a counter-based generator: word c of stream `seed`.  Same integer specification as the
// device generator (csrc/lzx_graph.hip) so host- and device-built graphs are identical.
inline std::uint64_t word(std::uint64_t seed, std::uint64_t c) {
  std::uint64_t z = seed + (c + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline std::uint64_t below(std::uint64_t w, std::uint64_t n) { return ((w >> 32) * n) >> 32; }
inline std::uint64_t key(std::uint64_t r, std::uint64_t c) { return (r << 32) | c; }
}  // namespace

void adjMatrix::release() {
  delete[] row_offset;
  delete[] col_idx;
  row_offset = col_idx = nullptr;
}

void adjMatrix::steal(adjMatrix &rhs) {
  row_offset = rhs.row_offset;
  col_idx = rhs.col_idx;
  n = rhs.n;
  edge_count = rhs.edge_count;
  barabasi_degree = rhs.barabasi_degree;
  matrix_type = rhs.matrix_type;
  seed = rhs.seed;
  rhs.row_offset = rhs.col_idx = nullptr;
  rhs.n = rhs.edge_count = 0;
}

void adjMatrix::csr_from_keys(std::vector<std::uint64_t> &keys) {
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  release();
  row_offset = new unsigned[static_cast<std::size_t>(n) + 1];
  col_idx = new unsigned[std::max<std::size_t>(keys.size(), 1)];
  std::size_t i = 0;
  for (unsigned r = 0; r < n; ++r) {
    row_offset[r] = static_cast<unsigned>(i);
    while (i < keys.size() && (keys[i] >> 32) == r) {
      col_idx[i] = static_cast<unsigned>(keys[i] & 0xffffffffu);
      ++i;
    }
  }
  row_offset[n] = static_cast<unsigned>(keys.size());
  edge_count = static_cast<unsigned>(keys.size() / 2);  // as adjMatrix.cc:44 of the reference
}

adjMatrix::adjMatrix(unsigned N, unsigned E, std::ifstream &f) : n{N}, edge_count{E}, matrix_type{'f'} {
  populate_sparse_matrix(f);
}

adjMatrix::adjMatrix(unsigned N, unsigned m, char c) : n{N}, barabasi_degree{m}, matrix_type{c} {
  generate_sparse_matrix(c);
}

adjMatrix::adjMatrix(unsigned N, unsigned E) : n{N}, matrix_type{'r'} {
  // the reference folds an over-full request back into range: E % (n(n-1)/2 + 1)
  const std::uint64_t cap = static_cast<std::uint64_t>(N) * (N - 1) / 2 + 1;
  edge_count = static_cast<unsigned>(E % cap);
  generate_sparse_matrix('r');
}

void adjMatrix::populate_sparse_matrix(std::ifstream &f) {
  // Slurp the rest of the stream and parse unsigned integers by hand: an order of magnitude faster
  // than operator>> and independent of line structure, like the reference's `f >> col >> row`.
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<std::uint64_t> keys;
  keys.reserve(2 * static_cast<std::size_t>(edge_count));
  const char *p = text.data(), *end = p + text.size();
  auto next = [&](std::uint64_t &out) -> bool {
    while (p < end && (*p < '0' || *p > '9')) ++p;
    if (p >= end) return false;
    std::uint64_t v = 0;
    while (p < end && *p >= '0' && *p <= '9') v = v * 10 + static_cast<std::uint64_t>(*p++ - '0');
    out = v;
    return true;
  };
  for (unsigned e = 0; e < edge_count; ++e) {
    std::uint64_t col, row;
    if (!next(col) || !next(row)) break;  // short file: keep what was read (the reference reads garbage)
    if (col == 0 || row == 0 || col > n || row > n) throw std::runtime_error("adjMatrix: vertex id out of range");
    keys.push_back(key(row - 1, col - 1));  // files are 1-indexed
    keys.push_back(key(col - 1, row - 1));
  }
  csr_from_keys(keys);
}

void adjMatrix::generate_sparse_matrix(char c) {
  switch (c) {
    case 'b': barabasi(barabasi_degree); break;
    case 'r': random_adj(); break;
    default: throw std::invalid_argument("adjMatrix: unknown generator (use 'b' or 'r')");
  }
}

// G(n, M): M uniform endpoint pairs; self loops and repeats are dropped (so slightly fewer than M
// edges survive, as in any multigraph-free G(n, M) sampler without rejection).
void adjMatrix::random_adj() {
  std::vector<std::uint64_t> keys;
  keys.reserve(2 * static_cast<std::size_t>(edge_count));
  for (std::uint64_t e = 0; e < edge_count; ++e) {
    const std::uint64_t u = below(word(seed, 2 * e), n), v = below(word(seed, 2 * e + 1), n);
    if (u == v) continue;
    keys.push_back(key(u, v));
    keys.push_back(key(v, u));
  }
  csr_from_keys(keys);
}

// Preferential attachment: start from a clique on m + 1 vertices; vertex t picks m targets with
// probability proportional to degree by sampling a uniform position in the running endpoint list.
void adjMatrix::barabasi(unsigned m) {
  if (m == 0 || n <= m) throw std::invalid_argument("adjMatrix: barabasi needs 0 < m < n");
  std::vector<unsigned> endpoints;
  std::vector<std::uint64_t> keys;
  endpoints.reserve(2ull * m * n);
  keys.reserve(2ull * m * n);
  auto link = [&](unsigned a, unsigned b) {
    keys.push_back(key(a, b));
    keys.push_back(key(b, a));
    endpoints.push_back(a);
    endpoints.push_back(b);
  };
  for (unsigned a = 0; a <= m; ++a)
    for (unsigned b = a + 1; b <= m; ++b) link(a, b);
  std::uint64_t ctr = 0;
  std::vector<unsigned> chosen;
  for (unsigned t = m + 1; t < n; ++t) {
    chosen.clear();
    const std::size_t pool = endpoints.size();
    while (chosen.size() < m) {
      const unsigned cand = endpoints[below(word(seed, ctr++), pool)];
      if (std::find(chosen.begin(), chosen.end(), cand) == chosen.end()) chosen.push_back(cand);
    }
    for (unsigned c : chosen) link(t, c);
  }
  csr_from_keys(keys);
}

adjMatrix adjMatrix::rmat(unsigned scale, unsigned N, std::uint64_t draws, std::uint64_t seed_, double a,
                          double b, double c) {
  if (scale == 0 || scale > 32 || (scale < 32 && N > (1ull << scale))) throw std::invalid_argument("adjMatrix::rmat: n > 2^scale");
  adjMatrix g;
  g.n = N;
  g.matrix_type = 'm';
  g.seed = seed_;
  const unsigned ta = static_cast<unsigned>(a * 65536 + 0.5), tab = static_cast<unsigned>((a + b) * 65536 + 0.5),
                 tabc = static_cast<unsigned>((a + b + c) * 65536 + 0.5);
  std::vector<std::uint64_t> keys;
  keys.reserve(2 * draws);
  for (std::uint64_t e = 0; e < draws; ++e) {
    for (unsigned t = 0; t < 8; ++t) {
      std::uint64_t u = 0, v = 0, w = 0;
      for (unsigned l = 0; l < scale; ++l) {
        if ((l & 3) == 0) w = word(seed_, (e * 8 + t) * 8 + (l >> 2));
        const unsigned r = static_cast<unsigned>(w & 0xffff);
        w >>= 16;
        u = (u << 1) | (r >= tab);
        v = (v << 1) | ((r >= ta && r < tab) || r >= tabc);
      }
      if (u < N && v < N) {
        if (u != v) { keys.push_back(key(u, v)); keys.push_back(key(v, u)); }
        break;
      }
    }
  }
  g.csr_from_keys(keys);
  return g;
}

std::string adjMatrix::write_matrix_to_file(const std::string &dir) const {
  const std::string filename = dir + std::string(1, matrix_type) + "n" + std::to_string(n) + "e" + std::to_string(edge_count);
  std::cout << "Filename: " << filename << '\n';
  std::ofstream f(filename);
  if (f.fail()) throw std::runtime_error("adjMatrix: cannot open " + filename);
  f << n << ' ' << n << ' ' << edge_count << '\n';
  for (unsigned r = 0; r < n; ++r)
    for (unsigned j = row_offset[r]; j < row_offset[r + 1]; ++j)
      if (col_idx[j] > r) f << col_idx[j] + 1 << ' ' << r + 1 << '\n';
  return filename;
}

void adjMatrix::print_full() const {
  for (unsigned r = 0; r < n; ++r) {
    unsigned j = row_offset[r];
    for (unsigned c = 0; c < n; ++c) {
      const bool one = j < row_offset[r + 1] && col_idx[j] == c;
      if (one) ++j;
      std::cout << (one ? 1 : 0) << ' ';
    }
    std::cout << '\n';
  }
}

std::ostream &operator<<(std::ostream &os, const adjMatrix &A) {
  os << "JA\n";
  for (unsigned i = 0; i < 2 * A.edge_count; ++i) os << A.col_idx[i] << ' ';
  os << "\nIA\n";
  for (unsigned i = 0; i <= A.n; ++i) os << A.row_offset[i] << ' ';
  return os;
}
