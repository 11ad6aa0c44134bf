"""tools/referee_c2.py -- CPU only (about five minutes, 8 GB): what the extended-precision referee (oracle/referee.c) says
about BASELINE C2 (R-MAT scale 20, 20 M draws, x0 = ones).  The numbers quoted in DESIGN.md section 4 come from here.

  part 1  the fp64 oracle (serial/'s loop restated) against the referee as k grows, and how far the k-step answer itself
          is from the converged one (referee_k vs referee_50);
  part 2  WHY the oracle is 3e-8 off at k = 50: the same fp64 recurrence with pairwise-summed inner products (np.dot) is
          1e-12 off -- serial/'s left-to-right sums over 10^6 terms (serial/lib/lanczos.cc:155-171) are the error, not the
          loss of orthogonality, which both runs suffer alike (max |q_0 . q_j| = 0.17);
  part 3  the reference's Arnoldi schedule (decompose_with_arnoldi, every 2 iterations) gives garbage in EVERY precision on
          this graph, every 1 iteration works.
usage: python tools/referee_c2.py [1] [2] [3]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import oracle as O

n = 1 << 20
rp, ci = O.gen_rmat(20, n, 20_000_000, 1234)
x = np.ones(n)
parts = [int(a) for a in sys.argv[1:]] or [1, 2, 3]


def sw(a, b, xn, cap):
    lam, V = O.eigen(a, b)
    s = 1.0 if cap is None else min(1.0, cap / lam.max())
    return V @ (np.exp(s * (lam - lam.max())) * (xn * V[0, :]))


def ri(u, v):
    return np.abs(u - v).max() / np.abs(v).max()


R50 = O.referee_expm(rp, ci, 50, x, caps=(0.0, 40.0), reorth=1)
if 1 in parts:
    a, b, Q, xn = O.lanczos(rp, ci, 50, x, q_colmajor=True)
    for k in (8, 10, 12, 15, 20, 30, 40, 50):
        Rk = O.referee_expm(rp, ci, k, x, caps=(0.0, 40.0), reorth=0)
        for i, cap in enumerate((None, 40.0)):
            yo = sw(a[:k], b[:k - 1], xn, cap) @ Q[:k]
            print(f"k={k} cap={cap}: oracle_k vs referee_k {ri(yo, Rk['ans'][i]):.2e}; referee_k vs referee_50 (full re-orth) "
                  f"{ri(Rk['ans'][i], R50['ans'][i]):.2e}; orthogonality lost: referee {Rk['orth_loss']:.1e}, oracle "
                  f"{max(abs(Q[0] @ Q[j]) for j in range(2, k)):.1e}", flush=True)

if 2 in parts or 3 in parts:
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(len(ci)), ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n))

    def run(every, dot, k=50):
        xn = np.sqrt(dot(x, x))
        q, qp = x / xn, None
        a, b, Q = np.zeros(k), np.zeros(k - 1), np.zeros((k, n))
        for j in range(k):
            v = A @ q
            if every and j % every == 0 and j > 2:
                for m in range(j - 1):
                    d = dot(v, Q[m])
                    v = v - d * Q[m]
            a[j] = dot(v, q)
            v = v - a[j] * q
            if j > 0:
                v = v - b[j - 1] * qp
            Q[j] = q
            if j < k - 1:
                b[j] = np.sqrt(dot(v, v))
                qp, q = q, v / b[j]
        return a, b, Q, xn

if 2 in parts:
    a, b, Q, xn = run(0, np.dot)
    for i, cap in enumerate((None, 40.0)):
        print(f"fp64 recurrence with pairwise-summed inner products, k=50, cap={cap}: vs referee {ri(sw(a, b, xn, cap) @ Q, R50['ans'][i]):.2e}")
    print(f"   its orthogonality is lost all the same: max |q_0 . q_j| = {max(abs(Q[0] @ Q[j]) for j in range(2, 50)):.2e}")
    ao, bo, _, _ = O.lanczos(rp, ci, 2, x, want_q=False)
    print(f"   beta_0: oracle (left-to-right sums) vs pairwise: {abs(bo[0] - b[0]) / b[0]:.2e} relative")

if 3 in parts:
    for every in (1, 2, 3):
        a, b, Q, xn = run(every, np.dot)
        print(f"Arnoldi pass every {every} (fp64, pairwise sums): vs referee {ri(sw(a, b, xn, 40.0) @ Q, R50['ans'][1]):.2e}, "
              f"max |q_0 . q_j| = {max(abs(Q[0] @ Q[j]) for j in range(2, 50)):.2e}")
        ao, bo, Qo, xno = O.lanczos_arnoldi(rp, ci, 50, x, every=every)
        print(f"   the oracle's restatement (left-to-right sums): vs referee {ri(sw(ao, bo, xno, 40.0) @ Qo, R50['ans'][1]):.2e}")
        RR = O.referee_expm(rp, ci, 50, x, caps=(0.0, 40.0), reorth=100 + every)
        print(f"   the same schedule in extended precision: vs full re-orthogonalisation {ri(RR['ans'][1], R50['ans'][1]):.2e}, "
              f"max |q_0 . q_j| = {RR['orth_loss']:.2e}", flush=True)
