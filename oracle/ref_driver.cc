// oracle/ref_driver.cc -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A C-ABI shim around the reference's *own* compiled sources (they stay where
// they are under /root/reference; oracle/Makefile passes them to g++ directly
// and writes only into oracle/_ref/).  It exposes exactly the two pieces of the
// hot path that build here without stand-in headers:
//   * the text loader  adjMatrix(N, E, ifstream&)  -> populate_sparse_matrix
//     (serial/lib/adjMatrix.h:39-46, serial/lib/adjMatrix.cc:21-54)
//   * the CPU SpMV     spMV<double>                 (serial/lib/SPMV.cc:19-31)
// serial/lib/{lanczos,eigen,multiplyOut}.cc include lapacke.h / cblas.h, which
// this image does not have, so they are NOT part of this build.
//
// This file contains none of the reference's code: it only includes its headers
// and calls its functions.  The CSR arrays are private in adjMatrix; they are
// reached through the class's own friend declaration
//     template <typename T> friend void spMV(const adjMatrix&, const T* const, T* const);
// by specialising that template for a tag type defined here.
#include <cstdint>
#include <cstring>
#include <fstream>

#include "adjMatrix.h"

template <typename T>
void spMV(const adjMatrix &, const T *const, T *const);   // defined + instantiated in SPMV.cc

namespace {
struct csr_peek {
    long unsigned *row_offset;
    long unsigned *col_idx;
};
}  // namespace

template <>
void spMV<csr_peek>(const adjMatrix &A, const csr_peek *const, csr_peek *const out)
{
    out->row_offset = A.row_offset;
    out->col_idx = A.col_idx;
}

struct ref_graph {
    adjMatrix A;
    long unsigned n = 0, declared_edges = 0;
    long unsigned row_offset0_as_loaded = 0;
};

extern "C" {

// What serial/main.cc:33-41 does for make_matrix == 'f'.
void *ref_load(const char *path)
{
    std::ifstream fs;
    fs.open(path);
    if (fs.fail()) return nullptr;
    ref_graph *g = new ref_graph;
    long unsigned n = 0, edges = 0;
    fs >> n >> n >> edges;
    adjMatrix B(n, edges, fs);
    fs.close();
    g->A = std::move(B);
    g->n = n;
    g->declared_edges = edges;
    // populate_sparse_matrix never writes row_offset[0] (adjMatrix.cc:34-41);
    // record what the fresh heap held, then set it to the 0 every consumer assumes.
    csr_peek p{};
    spMV<csr_peek>(g->A, nullptr, &p);
    g->row_offset0_as_loaded = p.row_offset[0];
    p.row_offset[0] = 0;
    return g;
}

void ref_info(void *h, uint64_t *n, uint64_t *edge_count, uint64_t *row_offset0_as_loaded)
{
    ref_graph *g = static_cast<ref_graph *>(h);
    *n = g->A.get_n();
    *edge_count = g->A.get_edges();
    *row_offset0_as_loaded = g->row_offset0_as_loaded;
}

void ref_csr(void *h, uint64_t *row_offset_out, uint64_t *col_idx_out)
{
    ref_graph *g = static_cast<ref_graph *>(h);
    csr_peek p{};
    spMV<csr_peek>(g->A, nullptr, &p);
    const uint64_t n = g->A.get_n();
    for (uint64_t i = 0; i <= n; ++i) row_offset_out[i] = p.row_offset[i];
    const uint64_t nnz = p.row_offset[n];
    for (uint64_t i = 0; i < nnz; ++i) col_idx_out[i] = p.col_idx[i];
}

void ref_spmv(void *h, const double *in, double *out)
{
    ref_graph *g = static_cast<ref_graph *>(h);
    spMV<double>(g->A, in, out);
}

void ref_free(void *h) { delete static_cast<ref_graph *>(h); }

}  // extern "C"
