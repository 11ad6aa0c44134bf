"""tests/golden/make_golden.py -- regenerates the committed golden vectors.  Runs ONLY where
/root/reference is mounted (this build container): it needs oracle/_ref/libref_serial.so, i.e. the
reference's own SPMV.cc + adjMatrix.cc compiled by oracle/Makefile.

Each fixture <name>.npz holds data only (no reference text):
    mtx_n, mtx_pairs      the input graph file content: header n and the E "col row" 1-indexed pairs
    ref_row_offset, ref_col_idx, ref_edge_count
                          CSR built by the REFERENCE loader, adjMatrix(N, E, ifstream&)
                          (serial/lib/adjMatrix.cc:21-54); row_offset[0] is set to 0 (the reference
                          leaves it unwritten)
    x, ref_spmv           a seeded input vector and the REFERENCE spMV<double> of it (serial/lib/SPMV.cc)
    k, alpha, beta, ans   the Lanczos loop of oracle/lanczos_oracle.c run OVER THE REFERENCE'S spMV
                          (x0 = ones, serial/main.cc:79), then eigen + multOut as oracle.py restates them
    expm_ref              scipy.sparse.linalg.expm_multiply(A, ones): an independent e^A x
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def star_plus_ring(n):
    """hub 0 joined to everyone + a ring: extreme skew, deg(0) = n - 1."""
    src = np.concatenate([np.zeros(n - 1, dtype=np.uint64), np.arange(1, n, dtype=np.uint64)])
    dst = np.concatenate([np.arange(1, n, dtype=np.uint64), np.roll(np.arange(1, n, dtype=np.uint64), -1)])
    keys = np.concatenate([(src << np.uint64(32)) | dst, (dst << np.uint64(32)) | src])
    return O.csr_from_keys(n, keys)


def tie_last_to_first(csr):
    """R-MAT leaves vertex n-1 isolated, and the reference loader never writes the offsets of trailing empty
    rows (serial/lib/adjMatrix.cc:34-41): give the last vertex one edge to vertex 0."""
    ro, ci = csr
    n = len(ro) - 1
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(ro.astype(np.int64)))
    keys = (rows << np.uint64(32)) | ci.astype(np.uint64)
    extra = np.array([((n - 1) << 32) | 0, n - 1], dtype=np.uint64)
    return O.csr_from_keys(n, np.concatenate([keys, extra]))


CASES = {
    # BASELINE.json configs[0] (C1): serial/main.cc on a 10k-node Erdos-Renyi graph, k = 20 -- bench.py's workload c1
    "er_c1_n10000": (lambda: O.gen_er(10000, 100000, 1234), 20),
    "er_n1000": (lambda: O.gen_er(1000, 5000, 1234), 20),
    "er_n4000_deg20": (lambda: O.gen_er(4000, 40000, 1234), 20),
    "rmat_n4096": (lambda: tie_last_to_first(O.gen_rmat(12, 4096, 30000, 1234)), 20),
    "rmat_n3000_skew": (lambda: tie_last_to_first(O.gen_rmat(12, 3000, 60000, 7, a=0.65, b=0.15, c=0.15)), 16),
    "star_ring_n1500": (lambda: star_plus_ring(1500), 12),
}


def main():
    O.build(ref=True)
    assert O.ref() is not None, "oracle/_ref could not be built (is /root/reference mounted?)"
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    only = set(sys.argv[1:])   # no arguments: regenerate everything
    for name, (make, k) in CASES.items():
        if only and name not in only:
            continue
        ro, ci = make()
        n = len(ro) - 1
        deg = np.diff(ro.astype(np.int64))
        assert deg[0] > 0 and deg[-1] > 0, f"{name}: the reference loader needs deg(0) > 0 and deg(n-1) > 0"
        path = f"/tmp/golden_{name}.mtx"
        E = O.write_mtx(path, n, ro, ci)
        tok = np.array(open(path).read().split(), dtype=np.int64)
        G = O.RefGraph(path)
        rro, rci = G.csr()
        x = np.random.default_rng(1234).random(n)
        ref_y = G.spmv(x)
        cb = G.spmv_callback()
        alpha, beta, Q, xn = O.lanczos(ro, ci, k, np.ones(n), ext_spmv=cb)
        lam, V = O.eigen(alpha, beta)
        ans = O.mult_out(Q, V, lam, xn)
        A = sp.csr_matrix((np.ones(len(ci)), ci.astype(np.int64), ro.astype(np.int64)), shape=(n, n))
        expm_ref = sla.expm_multiply(A, np.ones(n))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), mtx_n=n, mtx_pairs=tok[3:].reshape(-1, 2).astype(np.int32),
                            ref_row_offset=rro.astype(np.uint32), ref_col_idx=rci.astype(np.uint32),
                            ref_edge_count=G.edge_count, x=x, ref_spmv=ref_y, k=k, alpha=alpha, beta=beta, ans=ans,
                            expm_ref=expm_ref)
        print(f"{name}: n={n} E={E} ref_edges={G.edge_count} maxdeg={deg.max()} k={k} "
              f"rel-inf(ans, expm_multiply)={np.abs(ans - expm_ref).max() / np.abs(expm_ref).max():.2e} "
              f"row_offset0_as_loaded={G.row_offset0_as_loaded}")
        G.close()


if __name__ == "__main__":
    main()
