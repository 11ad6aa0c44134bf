"""tools/summarize_prof.py <gpurun_out/prof_dir> <tag> [workload] -- condenses a rocprofv3 run (kernel-trace stats + separate
FETCH_SIZE / WRITE_SIZE passes, as MI355X_MICROARCH.md's HBM section prescribes) into profiles/<tag>_*; with a workload key (c2, c3) also refreshes that entry of
profiles/pmc_traffic.json, the per-SpMV HBM traffic bench.py reports as roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    return name if len(name) < 120 else name[:117] + "..."


stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])

pmc = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith(("void k_", "k_")) or "::k_pb" in k:
            pmc.setdefault(k, {})[counter + "_KiB_avg_per_launch"] = sum(v) / len(v)
            pmc[k][counter + "_launches"] = len(v)
# HBM traffic per launch: FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section):
# doubled for the read side; WRITE_SIZE is exact.
for k, d in pmc.items():
    f = d.get("FETCH_SIZE_KiB_avg_per_launch")
    w = d.get("WRITE_SIZE_KiB_avg_per_launch")
    if f is not None and w is not None:
        d["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
        d["hbm_bytes_per_launch_raw"] = (f + w) * 1024.0
bench = {}
for name in ("bench_kt.json", "bench_fetch.json", "bench_write.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            bench[name] = json.loads(lines[-1])
json.dump({"pmc": pmc, "bench_lines": bench}, open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1)
if len(sys.argv) > 3:
    spmv = ("k_spmv<", "k_long_finish", "k_pb_scatter", "k_pb_gather", "k_pb_finish")
    kern = {k[:60]: d["hbm_bytes_per_launch_corrected"] for k, d in pmc.items()
            if any(s in k for s in spmv) and "hbm_bytes_per_launch_corrected" in d}
    path = os.path.join(out_dir, "pmc_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    # the build the counters were taken on: config.build_id of the profiled runs' own bench lines (bench.py hashes the kernel
    # sources); a figure whose passes ran on different builds is not written
    ids = {b.get("config", {}).get("build_id") for n_, b in bench.items() if n_ in ("bench_fetch.json", "bench_write.json")}
    if len(ids) != 1 or None in ids:
        sys.exit(f"FETCH_SIZE / WRITE_SIZE passes carry no single build id: {ids}")
    table[sys.argv[3]] = {
        "build_id": ids.pop(),
        "kernels": kern, "hbm_bytes_per_launch": sum(kern.values()),
        # the two directions apart (bench.py prices them separately: roofline.byte_cost_model)
        "read_bytes_per_launch": sum(2.0 * d["FETCH_SIZE_KiB_avg_per_launch"] * 1024.0 for k, d in pmc.items()
                                     if any(s in k for s in spmv) and "hbm_bytes_per_launch_corrected" in d),
        "written_bytes_per_launch": sum(d["WRITE_SIZE_KiB_avg_per_launch"] * 1024.0 for k, d in pmc.items()
                                        if any(s in k for s in spmv) and "hbm_bytes_per_launch_corrected" in d),
        "source": f"profiles/{tag}_pmc.json: sum over the kernels of one SpMV (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                  "separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; the doubling was checked on k_scale / "
                  "k_axpy_norm, whose byte counts are known)"}
    json.dump(table, open(path, "w"), indent=1)
print("wrote", sorted(f for f in os.listdir(out_dir) if f.startswith(tag)))
