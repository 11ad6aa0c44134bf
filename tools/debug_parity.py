import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg(); O = ge.load_oracle()
graphs = {
 "er_c1": O.gen_er(10000, 100000, 1234),
 "rmat_s14": O.gen_rmat(14, 12000, 200000, 7),
 "er_tiny": O.gen_er(130, 300, 3),
 "smoke": O.gen_rmat(15, 20000, 150000, 1234),
}
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for hub in (None,):
  for name, (rp, ci) in graphs.items():
    n = len(rp) - 1
    opts = {} if hub is None else {"hub_entries": hub}
    eng = pkg.Engine(0, **opts); eng.set_graph_csr(rp, ci)
    gi = eng.info()
    x = np.random.default_rng(1).random(n)
    y = eng.spmv(x); yr = O.spmv(rp, ci, x)
    bad = np.nonzero(y != yr)[0]
    k = min(K, n - 1)
    a, b, Q, xn, st = eng.lanczos(np.ones(n), k)
    ar, br, Qr, xnr = O.lanczos(rp, ci, k, np.ones(n), q_colmajor=True)
    print(name, opts, "long", gi["long_rows"], "maxdeg", gi["max_degree"], "spmv mismatches", len(bad),
          "maxrel", (np.abs(y - yr) / np.maximum(np.abs(yr), 1e-300)).max(),
          "| alpha rel", np.abs(a - ar).max() / np.abs(ar).max(), "beta rel", np.abs(b - br).max() / np.abs(br).max(),
          "Q maxabs", np.abs(Q - Qr).max())
    lam, V = O.eigen(ar, br); ans_ref = O.mult_out(np.ascontiguousarray(Qr.T), V, lam, xnr)
    lg, Vg = O.eigen(a, b); ans = O.mult_out(np.ascontiguousarray(Q.T), Vg, lg, xn)
    ansd = eng.multout(Vg @ (np.exp(lg) * (xn * Vg[0, :])))
    print("   ANS rel-inf host", np.abs(ans - ans_ref).max() / np.abs(ans_ref).max(), "dev", np.abs(ansd - ans_ref).max() / np.abs(ans_ref).max(), "max", np.abs(ans_ref).max(),
          "first alpha rel", abs(a[0]-ar[0])/abs(ar[0]), abs(a[1]-ar[1])/abs(ar[1]), "lam_max", lam.max())
    eng.close()
