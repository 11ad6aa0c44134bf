// adjMatrix.cc -- loader and seeded generators for the adjMatrix drop-in (see adjMatrix.h).
#include "adjMatrix.h"

#include <sys/stat.h>

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <iterator>
#include <stdexcept>
#include <thread>

#include "lzx.h"

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
double seconds_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
bool env_on(const char *name) {
  const char *v = std::getenv(name);
  return v && *v && *v != '0';
}
void lzx_or_throw(int rc, const char *what) {
  if (rc != LZX_OK) throw std::runtime_error(std::string(what) + ": " + lzx_last_error());
}
}  // namespace

void adjMatrix::release() {
  delete[] row_offset;
  delete[] col_idx;
  row_offset = col_idx = nullptr;
  dev.reset();
}

void adjMatrix::steal(adjMatrix &rhs) {
  row_offset = rhs.row_offset;
  col_idx = rhs.col_idx;
  n = rhs.n;
  edge_count = rhs.edge_count;
  barabasi_degree = rhs.barabasi_degree;
  matrix_type = rhs.matrix_type;
  seed = rhs.seed;
  dev = std::move(rhs.dev);
  report = rhs.report;
  rhs.row_offset = rhs.col_idx = nullptr;
  rhs.n = rhs.edge_count = 0;
}

// One handle per configured GPU, wired as an in-process communicator when there are several
// (parallel-two-cards drives its two cards from one process the same way).
static std::shared_ptr<deviceGraph> make_handles() {
  const std::vector<int> ids = lzx_host_devices();
  if (ids.empty()) throw std::runtime_error("adjMatrix: no usable GPU (lzx_device_count() == 0)");
  auto g = std::make_shared<deviceGraph>();
  // one handle per card, wired as an in-process group: parallel-two-cards' per-card set-up (cu_lanczos.cu:73-112) in one call
  g->ranks.assign(ids.size(), nullptr);
  lzx_or_throw(lzx_create_group(g->ranks.data(), static_cast<int>(ids.size()), ids.data()), "lzx_create_group");
  return g;
}

std::shared_ptr<deviceGraph> adjMatrix::device_graph() const {
  if (dev) return dev;
  if (!row_offset) throw std::runtime_error("adjMatrix: empty graph");
  const auto t0 = std::chrono::steady_clock::now();
  auto g = make_handles();
  // stored entries: row_offset[n] (2 * edge_count is one too many per self loop of the file)
  // several cards: every card receives ITS OWN rows only, as parallel-two-cards gives each of its two cards its half of IA / JA
  // (parallel-two-cards/lib/cu_lanczos.cu:94-95,108-109) -- the CSR stays here and is streamed past each card in chunks
  if (g->ranks.size() > 1 && !env_on("LZX_WHOLE_GRAPH_PER_CARD"))
    for (lzx_ctx *h : g->ranks) lzx_or_throw(lzx_set_option(h, "sharded_ingest", 1), "lzx_set_option(sharded_ingest)");
  for (lzx_ctx *h : g->ranks) lzx_or_throw(lzx_set_graph_csr32(h, n, row_offset[n], row_offset, col_idx), "lzx_set_graph_csr32");
  g->setup_ms = seconds_since(t0) * 1e3;
  dev = g;
  return dev;
}

// Device ingest: the endpoint pairs go to the GPU(s) as they were parsed; symmetrising, sorting and de-duplicating
// happen there (lzx_set_graph_edges) and the CSR comes back for the CPU path.  false = no GPU (caller sorts on the host).
bool adjMatrix::ingest_on_device(const std::vector<unsigned> &src, const std::vector<unsigned> &dst) {
  if (env_on("LZX_HOST_INGEST") || lzx_host_devices().empty()) return false;
  const auto t0 = std::chrono::steady_clock::now();
  auto g = make_handles();
  for (lzx_ctx *h : g->ranks) lzx_or_throw(lzx_set_graph_edges(h, n, src.size(), src.data(), dst.data()), "lzx_set_graph_edges");
  lzx_graph_info gi;
  lzx_or_throw(lzx_get_graph_info(g->ranks[0], &gi), "lzx_get_graph_info");
  if (gi.nnz > 0xffffffffull) throw std::runtime_error("adjMatrix: more than 2^32 stored entries (unsigned col_idx offsets)");
  std::vector<std::uint64_t> rp(static_cast<std::size_t>(n) + 1);
  release();
  row_offset = new unsigned[static_cast<std::size_t>(n) + 1];
  col_idx = new unsigned[std::max<std::size_t>(gi.nnz, 1)];
  lzx_or_throw(lzx_get_graph_csr(g->ranks[0], rp.data(), col_idx), "lzx_get_graph_csr");
  for (std::size_t i = 0; i <= n; ++i) row_offset[i] = static_cast<unsigned>(rp[i]);
  edge_count = static_cast<unsigned>(gi.nnz / 2);
  g->ingested = true;
  g->setup_ms = seconds_since(t0) * 1e3;
  dev = g;
  return true;
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
  // Slurp the rest of the stream: parsing by hand is an order of magnitude faster than operator>> and, like the
  // reference's `f >> col >> row`, independent of line structure.
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  build_from_text(text);
}

// Parse up to edge_count "col row" pairs (1-indexed) and build the CSR.  The text is cut at line ends into one
// piece per thread; pieces are parsed independently and joined in order, so the first edge_count pairs are the
// same ones a sequential reader takes.
void adjMatrix::build_from_text(const std::string &text) {
  const auto t0 = std::chrono::steady_clock::now();
  const char *base = text.data();
  const std::size_t len = text.size();
  unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (len < (1u << 20)) threads = 1;
  if (const char *e = std::getenv("LZX_PARSE_THREADS")) threads = std::max(1, std::atoi(e));
  std::vector<std::size_t> cut(threads + 1, len);
  cut[0] = 0;
  for (unsigned t = 1; t < threads; ++t) {
    std::size_t p = std::max(cut[t - 1], len / threads * t);
    while (p < len && base[p] != '\n') ++p;   // a piece ends behind a line end
    cut[t] = p;
  }
  std::vector<std::vector<unsigned>> ps(threads), pd(threads);
  std::vector<int> bad(threads, 0), odd(threads, 0);   // odd: the piece ended between the two numbers of a pair
  auto parse = [&](unsigned t) {
    const char *p = base + cut[t], *end = base + cut[t + 1];
    std::vector<unsigned> &s = ps[t], &d = pd[t];
    s.reserve((cut[t + 1] - cut[t]) / 12 + 16);
    d.reserve((cut[t + 1] - cut[t]) / 12 + 16);
    auto next = [&](std::uint64_t &out) -> bool {
      while (p < end && (*p < '0' || *p > '9')) ++p;
      if (p >= end) return false;
      std::uint64_t v = 0;
      while (p < end && *p >= '0' && *p <= '9') v = v * 10 + static_cast<std::uint64_t>(*p++ - '0');
      out = v;
      return true;
    };
    std::uint64_t col, row;
    while (next(col)) {
      if (!next(row)) { odd[t] = 1; break; }
      if (col == 0 || row == 0 || col > n || row > n) { bad[t] = 1; return; }   // pairs behind a bad one are never wanted
      s.push_back(static_cast<unsigned>(row - 1));   // files are 1-indexed
      d.push_back(static_cast<unsigned>(col - 1));
    }
  };
  if (threads == 1) parse(0);
  else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(parse, t);
    for (auto &th : pool) th.join();
    // The reference reads `f >> col >> row` as one token stream (parallel-final/lib/adjMatrix.cc:29-31): a file whose
    // pairs are not one per line can have a pair straddle a cut.  Then the pieces pair differently from the stream:
    // read it again as ONE piece.
    bool straddles = false;
    for (unsigned t = 0; t + 1 < threads; ++t) straddles = straddles || odd[t];
    if (straddles) {
      for (unsigned t = 0; t < threads; ++t) { ps[t].clear(); pd[t].clear(); bad[t] = odd[t] = 0; }
      cut.assign(2, len);
      cut[0] = 0;
      threads = 1;
      parse(0);
    }
  }
  // join in order, up to the declared edge count (a short file keeps what was read; the reference reads garbage)
  std::vector<unsigned> src, dst;
  src.reserve(edge_count);
  dst.reserve(edge_count);
  for (unsigned t = 0; t < threads && src.size() < edge_count; ++t) {
    const std::size_t want = edge_count - src.size();
    const std::size_t take = std::min<std::size_t>(ps[t].size(), want);
    // an out-of-range id only counts when it lies among the pairs that are actually taken: the bad pair is the one behind
    // the ps[t].size() good ones of its piece
    if (bad[t] && want > ps[t].size()) throw std::runtime_error("adjMatrix: vertex id out of range");
    src.insert(src.end(), ps[t].begin(), ps[t].begin() + take);
    dst.insert(dst.end(), pd[t].begin(), pd[t].begin() + take);
    std::vector<unsigned>().swap(ps[t]);
    std::vector<unsigned>().swap(pd[t]);
  }
  report = loadReport{};
  report.threads = threads;
  report.parse_s = seconds_since(t0);
  const auto t1 = std::chrono::steady_clock::now();
  if (ingest_on_device(src, dst)) {
    report.on_device = true;
  } else {
    std::vector<std::uint64_t> keys;
    keys.reserve(2 * src.size());
    for (std::size_t e = 0; e < src.size(); ++e) {
      keys.push_back(key(src[e], dst[e]));
      keys.push_back(key(dst[e], src[e]));
    }
    csr_from_keys(keys);
  }
  report.build_s = seconds_since(t1);
}

namespace {
struct cacheHeader {
  char magic[8];
  std::uint64_t src_size;
  std::int64_t src_mtime_s, src_mtime_ns;
  std::uint32_t n, edge_count;
  std::uint64_t nnz;
};
const char kMagic[8] = {'L', 'Z', 'X', 'C', 'S', 'R', '1', 0};
}  // namespace

adjMatrix adjMatrix::load(const std::string &path) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) throw std::runtime_error("adjMatrix: cannot open " + path);
  const std::string side = path + ".lzxcsr";
  const bool use_cache = !env_on("LZX_NO_CSR_CACHE");
  adjMatrix g;
  g.matrix_type = 'f';
  if (use_cache) {
    const auto t0 = std::chrono::steady_clock::now();
    if (FILE *f = std::fopen(side.c_str(), "rb")) {
      cacheHeader h;
      bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, kMagic, 8) == 0 &&
                h.src_size == static_cast<std::uint64_t>(st.st_size) && h.src_mtime_s == static_cast<std::int64_t>(st.st_mtim.tv_sec) &&
                h.src_mtime_ns == static_cast<std::int64_t>(st.st_mtim.tv_nsec) && h.nnz <= 0xffffffffull;
      if (ok) {
        g.n = h.n;
        g.edge_count = h.edge_count;
        g.row_offset = new unsigned[static_cast<std::size_t>(h.n) + 1];
        g.col_idx = new unsigned[std::max<std::size_t>(h.nnz, 1)];
        ok = std::fread(g.row_offset, sizeof(unsigned), static_cast<std::size_t>(h.n) + 1, f) == static_cast<std::size_t>(h.n) + 1 &&
             std::fread(g.col_idx, sizeof(unsigned), h.nnz, f) == h.nnz && g.row_offset[h.n] == h.nnz;
        // a side-car is only trusted as far as it can be checked: offsets ascending from 0, every column a vertex
        if (ok) ok = g.row_offset[0] == 0;
        for (std::size_t i = 0; ok && i < h.n; ++i) ok = g.row_offset[i] <= g.row_offset[i + 1];
        for (std::size_t i = 0; ok && i < h.nnz; ++i) ok = g.col_idx[i] < h.n;
      }
      std::fclose(f);
      if (ok) {
        g.report.from_cache = true;
        g.report.cache_s = seconds_since(t0);
        return g;
      }
      g.release();
      g.n = g.edge_count = 0;
    }
  }
  std::ifstream fs(path, std::ios::binary);
  if (fs.fail()) throw std::runtime_error("adjMatrix: cannot open " + path);
  unsigned n1 = 0, n2 = 0, e = 0;
  fs >> n1 >> n2 >> e;
  g.n = n2;          // `fs >> n >> n >> edges` (parallel-final/main.cu:62)
  g.edge_count = e;
  g.populate_sparse_matrix(fs);
  if (use_cache) {   // best effort: a read-only directory just means no cache
    const std::string tmp = side + ".tmp";
    if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
      cacheHeader h;
      std::memcpy(h.magic, kMagic, 8);
      h.src_size = static_cast<std::uint64_t>(st.st_size);
      h.src_mtime_s = static_cast<std::int64_t>(st.st_mtim.tv_sec);
      h.src_mtime_ns = static_cast<std::int64_t>(st.st_mtim.tv_nsec);
      h.n = g.n;
      h.edge_count = g.edge_count;
      h.nnz = g.row_offset[g.n];
      const bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 &&
                      std::fwrite(g.row_offset, sizeof(unsigned), static_cast<std::size_t>(g.n) + 1, f) == static_cast<std::size_t>(g.n) + 1 &&
                      std::fwrite(g.col_idx, sizeof(unsigned), h.nnz, f) == h.nnz;
      const bool closed = std::fclose(f) == 0;
      if (ok && closed) std::rename(tmp.c_str(), side.c_str());
      else std::remove(tmp.c_str());
    }
  }
  return g;
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
  for (unsigned i = 0; i < A.get_nnz(); ++i) os << A.col_idx[i] << ' ';
  os << "\nIA\n";
  for (unsigned i = 0; i <= A.n; ++i) os << A.row_offset[i] << ' ';
  return os;
}
