import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as ge
pkg = ge.load_pkg()
for name,(s,n,d) in {"c3":(24,10_000_000,200_000_000),"c2":(20,1<<20,20_000_000)}.items():
    e = pkg.Engine(0, propagation_blocking=0); e.gen_rmat(s,n,d,1234); rp,ci = e.get_graph_csr(); e.close()
    deg = np.diff(rp.astype(np.int64))
    print(name, "n", n, "isolated", int((deg==0).sum()), (deg==0).mean(), "deg<=2", (deg<=2).mean(), "deg<=8", (deg<=8).mean(), "median", np.median(deg), "mean", deg.mean())
    for q in (1,2,4,8,16,32,64,128,1024): print("   rows with deg >", q, int((deg>q).sum()), " nnz share", deg[deg>q].sum()/deg.sum())
