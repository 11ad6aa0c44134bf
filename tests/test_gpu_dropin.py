"""GPU: golden vectors through the C ABI, the C++ drop-in classes' device path, the `final` CLI and bench.py."""
import ctypes
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "msc-hpc-final-project_amd", "host")
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
_f64p = ctypes.POINTER(ctypes.c_double)


def write_pairs(path, n, pairs):
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(pairs)}\n")
        np.savetxt(f, pairs, fmt="%d")


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(q)[:-4] for q in GOLDEN])
def test_golden_through_c_abi(pkg, oracle, path):
    """Reference-derived fixtures: SpMV bit-exact (split rows 1e-13), centrality vector within 1e-10."""
    O = oracle
    g = np.load(path)
    n, k = int(g["mtx_n"]), int(g["k"])
    eng = pkg.Engine(0, propagation_blocking=0)                       # reference summation order for body rows
    eng.set_graph_csr32(g["ref_row_offset"], g["ref_col_idx"])       # parallel-final's `unsigned` arrays
    y = eng.spmv(g["x"])
    deg = np.diff(g["ref_row_offset"].astype(np.int64))
    order = np.argsort(-deg, kind="stable")
    n_split = -(-int((deg > 128).sum()) // 64) * 64
    body = np.ones(n, dtype=bool)
    body[order[:n_split]] = False
    assert np.array_equal(y[body], g["ref_spmv"][body])
    assert np.allclose(y, g["ref_spmv"], rtol=1e-13, atol=0)
    a, b, Q, xn, _ = eng.lanczos(np.ones(n), k)
    assert abs(a[0] - g["alpha"][0]) <= 1e-12 * abs(g["alpha"][0])
    lam, V = O.eigen(a, b)
    ans = eng.multout(V @ (np.exp(lam) * (xn * V[0, :])))
    assert np.abs(ans - g["ans"]).max() <= 1e-10 * np.abs(g["ans"]).max()
    assert np.abs(ans - g["expm_ref"]).max() <= 1e-10 * np.abs(g["expm_ref"]).max()
    eng.close()


def test_cpp_classes_device_path(pkg, tmp_path):
    """lanczosDecomp<double>(A, k, x, cuda=true) + eigenDecomp + multOut / cu_multOut vs the fixture."""
    pkg.lib()
    H = ctypes.CDLL(os.path.join(HOST_DIR, "libmschpc_host.so"))
    H.host_expm_file.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_uint, _f64p, _f64p]
    H.host_expm_file.restype = ctypes.c_long
    H.host_last_error.restype = ctypes.c_char_p
    for path in GOLDEN[:3]:
        g = np.load(path)
        n, k = int(g["mtx_n"]), int(g["k"])
        mtx = str(tmp_path / "g.mtx")
        write_pairs(mtx, n, g["mtx_pairs"])
        for device_multout in (0, 1):
            ans = np.zeros(n)
            rc = H.host_expm_file(mtx.encode(), k, 1, device_multout, ans.ctypes.data_as(_f64p), n, None, None)
            assert rc == n, H.host_last_error()
            assert np.abs(ans - g["ans"]).max() <= 1e-10 * np.abs(g["ans"]).max()


def test_cpp_classes_several_gpu_handles_and_device_ingest(pkg, tmp_path, monkeypatch):
    """The class path the way parallel-two-cards drives its cards: adjMatrix::load (parallel parse, CSR built by the
    device ingest and kept resident) -> lanczosDecomp(cuda) over 1 and over 3 handles (sharing this box's one GPU)
    -> host multOut (basis downloaded on first use) and cu_multOut; a second decomposition re-uses the graph."""
    pkg.lib()
    _u32p = ctypes.POINTER(ctypes.c_uint32)
    H = ctypes.CDLL(os.path.join(HOST_DIR, "libmschpc_host.so"))
    H.host_last_error.restype = ctypes.c_char_p
    H.host_load_path.argtypes = [ctypes.c_char_p, _u32p, _u32p, ctypes.c_uint, _u32p]
    H.host_load_path.restype = ctypes.c_long
    H.host_expm_path_devices.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                         _f64p, ctypes.c_uint, _f64p, _f64p, _u32p]
    H.host_expm_path_devices.restype = ctypes.c_long
    monkeypatch.setenv("LZX_NO_CSR_CACHE", "1")
    for path in (GOLDEN[0], GOLDEN[-2]):          # the C1 graph and a skewed R-MAT one
        g = np.load(path)
        n, k = int(g["mtx_n"]), int(g["k"])
        mtx = str(tmp_path / "g.mtx")
        write_pairs(mtx, n, g["mtx_pairs"])
        ro = np.zeros(n + 1, dtype=np.uint32)
        ci = np.zeros(2 * len(g["mtx_pairs"]) + 1, dtype=np.uint32)
        info = np.zeros(4, dtype=np.uint32)
        edges = H.host_load_path(mtx.encode(), ro.ctypes.data_as(_u32p), ci.ctypes.data_as(_u32p), len(ci), info.ctypes.data_as(_u32p))
        assert edges == int(g["ref_edge_count"]), H.host_last_error()
        assert info[1] == 1, "the loader did not use the device ingest on a GPU box"
        assert np.array_equal(ro, g["ref_row_offset"]) and np.array_equal(ci[:2 * edges], g["ref_col_idx"])   # bit-exact
        for devices in ([0], [0, 0, 0]):
            for device_multout in (0, 1):
                ans, used = np.zeros(n), np.zeros(1, dtype=np.uint32)
                dv = (ctypes.c_int * len(devices))(*devices)
                rc = H.host_expm_path_devices(mtx.encode(), k, dv, len(devices), device_multout, ans.ctypes.data_as(_f64p), n,
                                              None, None, used.ctypes.data_as(_u32p))
                assert rc == n, H.host_last_error()
                assert used[0] == len(devices)
                assert np.abs(ans - g["ans"]).max() <= 1e-10 * np.abs(g["ans"]).max(), (devices, device_multout)
        # the CSR built on the HOST (std::sort path of the loader), handed to three cards: each card receives its own rows only
        # (lzx_set_graph_csr32 with option sharded_ingest, as parallel-two-cards gives each card its half of IA / JA) -- the same
        # bits as with the whole CSR on every card
        got = {}
        for whole in ("0", "1"):
            with monkeypatch.context() as mp:
                mp.setenv("LZX_HOST_INGEST", "1")
                mp.setenv("LZX_WHOLE_GRAPH_PER_CARD", whole)
                ans, used = np.zeros(n), np.zeros(1, dtype=np.uint32)
                dv = (ctypes.c_int * 3)(0, 0, 0)
                rc = H.host_expm_path_devices(mtx.encode(), k, dv, 3, 1, ans.ctypes.data_as(_f64p), n, None, None, used.ctypes.data_as(_u32p))
                assert rc == n and used[0] == 3, H.host_last_error()
                assert np.abs(ans - g["ans"]).max() <= 1e-10 * np.abs(g["ans"]).max(), whole
                got[whole] = ans
        assert np.array_equal(got["0"], got["1"])


def test_final_cli(tmp_path):
    g = np.load(GOLDEN[0])
    n = int(g["mtx_n"])
    mtx = str(tmp_path / "graph.mtx")
    write_pairs(mtx, n, g["mtx_pairs"])
    out = subprocess.run([os.path.join(HOST_DIR, "final"), "-f", mtx, "-k", "20"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    assert "TIMING" in out.stdout and "ERROR CHECKING" in out.stdout and "Lanczos" in out.stdout
    rel = [l for l in out.stdout.splitlines() if l.startswith("Relative inf-norm")]
    assert rel and float(rel[0].split("=")[1]) <= 1e-10
    ans = np.loadtxt(mtx + ".ans20.txt")
    assert np.abs(ans - g["ans"]).max() <= 1e-5 * np.abs(g["ans"]).max()    # file holds 6 significant digits


def test_bench_contract_small_workload():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--steps", "20",
                          "--warmup", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["steps"] == 20 and j["n_gpus"] == 1 and j["dtype"] == "f64" and j["value"] > 0
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1


def test_convergence_monitor_device(pkg, tmp_path):
    """The same monitor on the GPU-resident basis (lzx_multout_f64 with k' <= k)."""
    pkg.lib()
    H = ctypes.CDLL(os.path.join(HOST_DIR, "libmschpc_host.so"))
    _u32p = ctypes.POINTER(ctypes.c_uint32)
    H.host_adaptive_file.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_double, ctypes.c_int, _f64p,
                                     ctypes.c_uint, _u32p, _f64p, ctypes.c_uint, _u32p]
    H.host_adaptive_file.restype = ctypes.c_long
    H.host_last_error.restype = ctypes.c_char_p
    g = np.load(GOLDEN[1])
    n = int(g["mtx_n"])
    mtx = str(tmp_path / "g.mtx")
    write_pairs(mtx, n, g["mtx_pairs"])
    ans = np.zeros(n)
    ks = np.zeros(16, dtype=np.uint32)
    ch = np.zeros(16)
    used = ctypes.c_uint()
    m = H.host_adaptive_file(mtx.encode(), 40, 5, 1e-12, 1, ans.ctypes.data_as(_f64p), n, ks.ctypes.data_as(_u32p),
                             ch.ctypes.data_as(_f64p), 16, ctypes.cast(ctypes.byref(used), _u32p))
    assert m > 2, H.host_last_error()
    assert ch[m - 1] <= 1e-12 and used.value < 40
    assert np.abs(ans - g["expm_ref"]).max() <= 1e-10 * np.abs(g["expm_ref"]).max()


def test_bench_under_torchrun_one_rank():
    """The launch line the driver uses for N > 1, at N = 1: torch.distributed (nccl) rendezvous, communicator id
    broadcast, lzx_comm_init_rank over the RCCL copy PyTorch already loaded, barrier / max plumbing."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c1",
           "--steps", "10", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["steps"] == 10 and j["value"] > 0


def test_bench_two_processes_share_the_gpu():
    """`bench.py --gpus 2` for real across processes on the one-GPU box: LZX_BENCH_ONE_GPU=1 puts both ranks on GPU 0 over
    the peer-window transport (RCCL refuses two ranks per GPU): engines wired across processes, both exchange modes timed,
    the overlapped trial under its watchdog, max / sum over ranks, one JSON line from rank 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "10", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, LZX_BENCH_ONE_GPU="1", LZX_IPC_TIMEOUT_MS="60000", HSA_ENABLE_IPC_MODE_LEGACY="0", LZX_BENCH_FORCE_ALT="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    tune = j["config"]["exchange_tuning_ms_per_iter"]
    assert j["n_gpus"] == 2 and j["steps"] == 10 and j["value"] > 0 and j["config"]["transport"] == "ipc"
    assert {"single", "overlapped"} <= set(tune) and j["config"]["lanczos_coefficients_finite"]
    # the overlapped exchange (two chunks, the second sparse, on the exchange stream) was measured in full and describes the
    # same decomposition as the single all-gather: leading coefficients equal to rounding; alpha_0 = nnz / n exactly
    assert tune["overlapped_vs_single_coefficients_rel"] < 1e-9 and j["config"]["alpha0_vs_closed_form_rel"] < 1e-14


def test_bench_rehearses_the_multi_gpu_flow():
    """The N > 1 flow of bench.py on the one-GPU box: exchange tuning over two engines (single all-gather, two-chunk
    overlapped exchange), the RCCL collectives on the 1-rank communicator (exchange_at_world_1), rank aggregation."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c2",
           "--steps", "10", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, LZX_BENCH_REHEARSE_MULTI="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    tune = j["config"]["exchange_tuning_ms_per_iter"]
    assert {"single", "overlapped"} <= set(tune) and min(tune["single"], tune["overlapped"]) > 0
    assert j["value"] > 0 and j["config"]["lanczos_coefficients_finite"]


def test_bench_watchdog_makes_a_hung_trial_visible():
    """a collective that never completes in bench.py's overlapped-exchange trial must show in the
    RETURN CODE.  Rehearsed on the one-GPU box with a trial limit no trial can meet (1 ms): the watchdog prints the line it
    holds -- the single all-gather measurement, marked "did not finish" -- and ends the rank with EXIT_TRIAL_HUNG, so the
    launcher reports failure; the held line is still a complete, valid JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c2",
           "--steps", "10", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, LZX_BENCH_REHEARSE_MULTI="1", LZX_BENCH_TRIAL_LIMIT_S="0.001"))
    assert out.returncode != 0, out.stdout[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stderr[-3000:]
    j = json.loads(lines[-1])
    assert j["value"] > 0 and j["steps"] == 10 and j["config"]["k"] == 50
    assert "did not finish" in j["config"]["exchange_tuning_ms_per_iter"]["overlapped"]
    assert "did not finish within" in out.stderr


def test_adaptive_stop_through_the_classes_and_the_cli(pkg, oracle, tmp_path):
    """N3 proper in the drop-in layer: lanczosDecomp(A, k_max, x, cuda = true, lanczosOptions{adaptive_step, adaptive_tol})
    advances the device decomposition in chunks and stops when the answer has converged -- fewer SpMVs than k_max, the
    same answer as running k_used iterations in one go (the leading block of the same recurrence), within 1e-10 of the
    oracle's e^A x; `final` prints k_used.  Also on three in-process handles, with the Arnoldi pass, and with the fp32 basis."""
    O = oracle
    pkg.lib()
    H = ctypes.CDLL(os.path.join(HOST_DIR, "libmschpc_host.so"))
    _u32p = ctypes.POINTER(ctypes.c_uint32)
    H.host_expm_options_file.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_uint, _f64p, _f64p, _u32p,
                                         _f64p, ctypes.c_uint]
    H.host_expm_options_file.restype = ctypes.c_long
    H.host_last_error.restype = ctypes.c_char_p
    rp, ci = O.gen_er(10000, 100000, 1234)               # BASELINE C1
    n, kmax, step, tol = 10000, 50, 5, 1e-12
    mtx = str(tmp_path / "c1.mtx")
    O.write_mtx(mtx, n, rp, ci)
    ans_ref = O.expm_action(rp, ci, kmax, np.ones(n))

    def run(k, arnoldi=0, step_=0, fp32=0, devices=None):
        if devices:
            os.environ["LZX_DEVICES"] = devices
        try:
            ans, alpha, beta = np.zeros(n), np.zeros(k), np.zeros(k)
            info, ch = np.zeros(4, dtype=np.uint32), np.zeros(16)
            rc = H.host_expm_options_file(mtx.encode(), k, 1, 1, arnoldi, step_, tol, fp32, 0, ans.ctypes.data_as(_f64p), n,
                                          alpha.ctypes.data_as(_f64p), beta.ctypes.data_as(_f64p), info.ctypes.data_as(_u32p),
                                          ch.ctypes.data_as(_f64p), 16)
            assert rc == n, H.host_last_error()
            return ans, alpha, beta, info, ch
        finally:
            os.environ.pop("LZX_DEVICES", None)

    ans, alpha, beta, info, ch = run(kmax, step_=step)
    k_used, iters, chunks, conv = (int(v) for v in info)
    assert conv == 1 and k_used < kmax and iters == k_used and chunks == k_used // step, info
    assert ch[0] == 1.0 and ch[chunks - 1] <= tol < ch[chunks - 2]
    assert np.abs(ans - ans_ref).max() <= 1e-10 * np.abs(ans_ref).max()
    ans_one_go, a1, b1, info1, _ = run(k_used)
    assert info1[0] == k_used and info1[1] == k_used
    assert np.array_equal(alpha[:k_used], a1) and np.array_equal(beta[:k_used - 1], b1[:k_used - 1]) and np.array_equal(ans, ans_one_go)
    print(f"adaptive device run on C1: k_used = {k_used} of {kmax}, changes {ch[:chunks]}")
    # three handles on this GPU; the Arnoldi pass every iteration; the basis stored as fp32 (1e-6 then, not 1e-10)
    ans3, _, _, info3, _ = run(kmax, step_=step, devices="0,0,0")
    assert info3[0] == k_used and np.abs(ans3 - ans_ref).max() <= 1e-10 * np.abs(ans_ref).max()
    ans_a, _, _, info_a, _ = run(kmax, arnoldi=1, step_=step)
    assert info_a[3] == 1 and np.abs(ans_a - ans_ref).max() <= 1e-10 * np.abs(ans_ref).max()
    ans_f, _, _, info_f, _ = run(kmax, step_=step, fp32=1)
    assert 1e-10 < np.abs(ans_f - ans_ref).max() / np.abs(ans_ref).max() <= 1e-6
    # the CLI
    out = subprocess.run([os.path.join(HOST_DIR, "final"), "-f", mtx, "-k", str(kmax)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, FINAL_ADAPTIVE_STEP=str(step), FINAL_ADAPTIVE_TOL=str(tol)))
    assert out.returncode == 0, out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("adaptive run: k_used")]
    assert line and f"k_used = {k_used} of at most {kmax}" in line[0] and "converged" in line[0], out.stdout[-1500:]


def test_reference_order_through_the_classes(pkg, oracle, tmp_path):
    """Drop-in parity, literally: the same graph file through `lanczosDecomp(A, k, x, /*cuda*/false)` -- the class layer's CPU
    path, serial/lib/lanczos.cc:9-56 restated -- and through `lanczosDecomp(A, k, x, true, {reference_order})`, the device
    path with serial/'s reduction orders: alpha, beta and every entry of Q bit-identical (the C shim compares them with
    memcmp).  And the default device path on the same file: the same alpha_0, not the same bits."""
    import ctypes
    import subprocess
    O = oracle
    H = ctypes.CDLL(os.path.join(ROOT, "msc-hpc-final-project_amd", "host", "libmschpc_host.so"))
    H.host_reference_order_check.restype = ctypes.c_long
    H.host_reference_order_check.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_uint)]
    H.host_last_error.restype = ctypes.c_char_p
    for name, (rp, ci) in (("er", O.gen_er(6000, 40000, 9)), ("rmat", O.gen_rmat(14, 12000, 150000, 5))):
        n = len(rp) - 1
        mtx = str(tmp_path / f"{name}.mtx")
        O.write_mtx(mtx, n, rp, ci)
        k = 24
        a_dev, a_cpu = np.zeros(k), np.zeros(k)
        diff = (ctypes.c_uint * 3)()
        rc = H.host_reference_order_check(mtx.encode(), k, a_dev.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                          a_cpu.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), diff)
        assert rc == 1, (name, rc, list(diff), H.host_last_error())
        assert np.array_equal(a_dev, a_cpu)
        a_ref, _, _, _ = O.lanczos(rp, ci, k, np.ones(n), want_q=False)
        assert np.array_equal(a_cpu, a_ref), name          # ... and both are the oracle's
    # the CLI: FINAL_REFERENCE_ORDER=1 makes `final`'s device run reproduce its serial run (relative error of the answers: 0)
    exe = os.path.join(ROOT, "msc-hpc-final-project_amd", "host", "final")
    out = subprocess.run([exe, "-n", "3000", "-e", "20000", "-k", "12"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, FINAL_REFERENCE_ORDER="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
