"""CPU: the parts of bench.py's contract that need no GPU -- the workload table, the traffic figures it quotes and the
profile files they name."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # defines main() only; nothing touches a GPU at import
    return m


def test_workloads_and_traffic_sources():
    b = _bench()
    assert {"c1", "c2", "c3", "er", "er1m"} <= set(b.WORKLOADS)
    assert b.HBM_PEAK_GBS == 8000.0                      # MI355X_MICROARCH.md: HBM3E, 8 TB/s
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key in ("c3", "c2", "er"):
        traffic, source = b.pmc_traffic(key, 1)
        assert traffic and traffic > 0 and source
        assert os.path.exists(os.path.join(ROOT, source.split(":")[0])), source    # the rocprofv3 summary it was taken from
        assert table[key]["hbm_bytes_per_launch"] == traffic
    assert b.pmc_traffic("c3", 8) == (None, None) or b.pmc_traffic("c3", 8)[0] is None   # only measured on one GPU
