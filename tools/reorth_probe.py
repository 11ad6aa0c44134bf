"""tools/reorth_probe.py -- GPU box: the device Arnoldi pass (option reorthogonalise = e) beside the oracle's restatement
of decompose_with_arnoldi on BASELINE C2: leading coefficients, loss of orthogonality, distance to the referee."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg, O = ge.load_pkg(), ge.load_oracle()
n, k = 1 << 20, 50
x0 = np.ones(n)


def sw(a, b, xn):
    lam, V = O.eigen(a, b)
    s = min(1.0, 40.0 / lam.max())
    return V @ (np.exp(s * (lam - lam.max())) * (xn * V[0, :]))


ref = None
for e in (0, 1, 2, 3):
    eng = pkg.Engine(0, reorthogonalise=e)
    eng.gen_rmat(20, n, 20_000_000, 1234)
    if ref is None:
        rp, ci = eng.get_graph_csr()
        ref = O.referee_expm(rp, ci, k, x0, caps=(40.0,), reorth=1)["ans"][0]
    a, b, Q, xn, st = eng.lanczos(x0, k)
    loss = max(abs(Q[0] @ Q[j]) for j in range(2, k))
    err = np.abs(eng.multout(sw(a, b, xn)) - ref).max() / np.abs(ref).max()
    if e:
        ao, bo, Qo, xno = O.lanczos_arnoldi(rp, ci, k, x0, every=e)
    else:
        ao, bo, Qo, xno = O.lanczos(rp, ci, k, x0, q_colmajor=True)
    loss_o = max(abs(Qo[0] @ Qo[j]) for j in range(2, k))
    err_o = np.abs(sw(ao, bo, xno) @ Qo - ref).max() / np.abs(ref).max()
    print(f"e={e}: engine loss {loss:.2e} err {err:.2e} | oracle loss {loss_o:.2e} err {err_o:.2e} | loop {st['loop_ms']:.1f} ms vec {st['vec_ms']:.1f} ms")
    print("   engine alpha[:14]", np.array2string(a[:14], precision=10))
    print("   oracle alpha[:14]", np.array2string(ao[:14], precision=10))
    print("   engine |q0.qj| j=2..20", np.array2string(np.array([abs(Q[0] @ Q[j]) for j in range(2, 21)]), precision=1))
    print("   oracle |q0.qj| j=2..20", np.array2string(np.array([abs(Qo[0] @ Qo[j]) for j in range(2, 21)]), precision=1), flush=True)
    eng.close()
