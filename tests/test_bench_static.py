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
        traffic, source, build = b.pmc_traffic(key, 1)
        assert traffic and traffic > 0 and source
        assert os.path.exists(os.path.join(ROOT, source.split(":")[0])), source    # the rocprofv3 summary it was taken from
        assert table[key]["hbm_bytes_per_launch"] == traffic
        # provenance: the figure names the build it was measured on, and the profile it cites holds the bench lines of that build
        assert build and len(build) == 16, key
        prof = json.load(open(os.path.join(ROOT, source.split(":")[0])))
        assert {l["config"]["build_id"] for n_, l in prof["bench_lines"].items() if n_ != "bench_kt.json"} == {build}, key
    assert b.pmc_traffic("c3", 8)[0] is None   # only measured on one GPU
    assert len(b.build_id()) == 16 and b.build_id() == b.build_id()


def test_traffic_kernels_exist_in_the_built_library():
    """every kernel whose PMC bytes a traffic figure sums is a kernel of the library as built now (a figure from a build whose
    kernels have since been renamed or removed must be regenerated)"""
    import re
    import subprocess
    lib = os.path.join(ROOT, "msc-hpc-final-project_amd", "liblzx.so")
    if not os.path.exists(lib):
        import __graft_entry__ as ge
        ge.build()
    blob = open(lib, "rb").read()
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, entry in table.items():
        for kern in entry["kernels"]:
            base = re.search(r"(k_[a-z0-9_]+)", kern).group(1)
            assert base.encode() in blob, (key, kern)


def test_byte_cost_model_prices_the_two_directions_apart():
    """roofline.byte_cost_model: the measured bytes of one SpMV, read and written apart (their sum IS roofline.traffic), priced at the
    box's own streaming rates; a copy (as many bytes each way) must cost exactly what the copy probe measured"""
    b = _bench()
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    read, copy = 6000.0, 5000.0
    for key in ("c3", "c2", "er"):
        m = b.byte_cost_model(key, 1, (read, copy), 1.0)
        assert abs(m["read_bytes"] + m["written_bytes"] - table[key]["hbm_bytes_per_launch"]) < 1.0, key
        assert m["read_bytes"] > m["written_bytes"] > 0
        assert abs(m["ms"] - (m["read_bytes"] / read + m["written_bytes"] / m["write_GBps"]) * 1e-6) < 1e-12
        assert abs(m["measured_over_model"] * m["ms"] - 1.0) < 1e-12
    m = b.byte_cost_model("c3", 1, (read, copy), 1.0)
    both = 1e9
    assert abs((both / m["read_GBps"] + both / m["write_GBps"]) - 2 * both / copy) < 1e-6   # the copy probe's own time
    assert b.byte_cost_model("c3", 8, (read, copy), 1.0) is None      # PMC passes exist for one GPU only
    assert b.byte_cost_model("c1", 1, (read, copy), 1.0) is None      # no pass for this workload
    assert b.byte_cost_model("c3", 1, (6000.0, 13000.0), 1.0) is None  # rates that imply a negative write cost: no model
