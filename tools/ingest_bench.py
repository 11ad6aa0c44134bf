"""tools/ingest_bench.py -- SURVEY 8(f) N1 before/after on the GPU box: load of a 10 M-edge text file through `final`
(adjMatrix::load): one parser thread + host sort (what round 1 had), 16 threads + host sort, 16 threads + device ingest,
and the binary side-car cache.  Prints the build lines of `final` and checks that the answers agree."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

O = ge.load_oracle()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FINAL = os.path.join(ROOT, "msc-hpc-final-project_amd", "host", "final")
n, draws = 1_000_000, 10_300_000
path = "/tmp/ingest_bench.mtx"
t = time.time()
rp, ci = O.gen_er(n, draws, 4321)
E = O.write_mtx(path, n, rp, ci)
print(f"wrote {path}: n={n} E={E} ({os.path.getsize(path) / 1e6:.0f} MB) in {time.time() - t:.1f} s", flush=True)


def run(tag, **env):
    e = dict(os.environ, FINAL_SKIP_SERIAL="1", FINAL_DEVICE_MULTOUT="1", **env)
    t0 = time.time()
    out = subprocess.run([FINAL, "-f", path, "-k", "20"], capture_output=True, text=True, env=e, timeout=900)
    dt = time.time() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l.strip() for l in out.stdout.splitlines()]
    build = [l for i, l in enumerate(lines) if "seconds" in l or l.startswith("(text parse") or l.startswith("(binary")]
    lan = [l for l in lines if l.startswith("Lanczos") or l.startswith("device loop only")]
    print(f"[{tag}] wall {dt:.2f} s | " + " | ".join(build) + " | " + " | ".join(lan), flush=True)
    return np.loadtxt(path + ".ans20.txt")


if os.path.exists(path + ".lzxcsr"):
    os.remove(path + ".lzxcsr")
a = run("1 thread, host sort (round 1)", LZX_NO_CSR_CACHE="1", LZX_PARSE_THREADS="1", LZX_HOST_INGEST="1")
b = run("16 threads, host sort", LZX_NO_CSR_CACHE="1", LZX_HOST_INGEST="1")
c = run("16 threads, device ingest", LZX_NO_CSR_CACHE="1")
d = run("device ingest, writes cache")
e = run("from the binary cache")
for other in (b, c, d, e):
    assert np.array_equal(a, other)
print("answers identical on all five paths")
