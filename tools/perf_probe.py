"""tools/perf_probe.py -- A/B probe of the SpMV kernel's knobs on synthetic R-MAT graphs (GPU box only).
usage: python tools/perf_probe.py [c2|c3] ..."""
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_pkg()
if os.environ.get("LZX_PROBE_DBG_LIB"):   # A/B against another build of the debug library (same box, same job)
    pkg.DBG_LIB_PATH = os.environ["LZX_PROBE_DBG_LIB"]
WORK = {"c2": (20, 1 << 20, 21_615_022), "c3": (24, 10_000_000, 207_184_357), "c2b": (20, 1_000_000, 20_000_000),
        "big": (25, 30_000_000, 600_000_000), "mid": (22, 4_000_000, 80_000_000), "c5": (27, 100_000_000, 2_000_000_000),
        # Erdos-Renyi (scale 0 = the ER generator): north_star's uniform family
        "er": (0, 10_000_000, 100_000_104), "er1m": (0, 1_000_000, 10_000_000), "er4m": (0, 4_000_000, 40_000_000)}


def run(name, opts_list, k=20):
    scale, n, draws = WORK[name]
    for opts in opts_list:
        eng = pkg.Engine(0, **opts)
        t = time.time()
        if scale == 0:
            eng.gen_er(n, draws, 1234)
        else:
            eng.gen_rmat(scale, n, draws, 1234)
        tg = time.time() - t
        gi = eng.info()
        avg, mn = eng.bench_spmv(20)
        bytes_ = 4 * gi["nnz"] + 4 * (n + 1) + 8 * n + 8 * n
        a, b, _, xn, st = eng.lanczos(np.ones(n), k, want_q=False)
        print(f"{name} {opts} gen={tg:.1f}s n={n} nnz={gi['nnz']} maxdeg={gi['max_degree']} long={gi['long_rows']} "
              f"padded={gi['sell_padded']} hub={gi['hub_entries']} pb={gi['pb_entries']} reduced={gi['pb_reduced_entries']} values={gi['pb_values']} | spmv avg {avg:.4f} ms min {mn:.4f} ms -> "
              f"{bytes_ / mn / 1e6:.1f} GB/s ({bytes_ / mn / 1e6 / 8000 * 100:.1f}% of 8 TB/s) | "
              f"lanczos k={k}: loop {st['loop_ms']:.2f} ms, spmv {st['spmv_ms']:.2f}, vec {st['vec_ms']:.2f} -> "
              f"{k / st['loop_ms'] * 1e3:.1f} it/s", flush=True)
        eng.close()


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("@")] or ["c2"]
    sets = [a[1:] for a in sys.argv[1:] if a.startswith("@")]
    opts = [dict(), dict(hub_entries=0), dict(hub_entries=16384, wgs_per_cu=1), dict(hub_entries=4096),
            dict(nt_index_loads=1), dict(hub_entries=0, nt_index_loads=1), dict(wgs_per_cu=1), dict(hub_entries=0, wgs_per_cu=4)]
    if "diag" in sets:
        opts = [dict(pb_stamps=1), dict(pb_stamps=1, pb_target=8192), dict(pb_stamps=1, pb_target=131072), dict(pb_stamps=1, pb_run_align=4),
                dict(pb_stamps=1, phase_mask=3 + 8), dict(pb_stamps=1, phase_mask=3 + 4), dict(pb_stamps=1, pb_unit=32768), dict(pb_stamps=1, pb_unit=262144)]
    if "diag2" in sets:
        opts = [dict(pb_stamps=1), dict(pb_stamps=1, pb_gather_waves=4), dict(pb_stamps=1, hub_entries=2), dict(pb_stamps=1, hub_entries=1024),
                dict(pb_stamps=1, hub_entries=4096), dict(pb_stamps=1, hub_entries=8192), dict(pb_stamps=1, hub_entries=2, pb_gather_waves=4)]
    if "tie" in sets:
        opts = [dict(tie_sort=1), dict(tie_sort=2), dict(tie_sort=0), dict(tie_sort=1), dict(tie_sort=2), dict(tie_sort=0)]
    if "grp" in sets:
        opts = [dict(pb_group=16384), dict(pb_group=0), dict(pb_group=8192), dict(pb_group=12288), dict(pb_group=20480), dict(pb_group=24576), dict(pb_group=16384), dict(pb_group=0)]
    if "stg" in sets:
        opts = [dict(pb_order=0), dict(pb_order=0, phase_mask=1), dict(pb_order=0, phase_mask=2), dict(pb_order=0, phase_mask=2, long_row=100000), dict(pb_order=1)]
    if "itm" in sets:
        opts = [dict(pb_order=0, phase_mask=1), dict(pb_order=0, phase_mask=2), dict(pb_order=0), dict(pb_order=0, spmv_deep=1), dict(pb_order=1)]
    if "base" in sets:
        opts = [dict(pb_order=1), dict(pb_order=0), dict(pb_order=1), dict(pb_order=0)]
    if "burst" in sets:
        opts = [dict(pb_order=0, stage_burst=1, phase_mask=0), dict(pb_order=0, stage_burst=0, phase_mask=0)] * 2
    if "nar" in sets:
        opts = [dict(pb_order=0, narrow_slices=1), dict(pb_order=0, narrow_slices=0), dict(pb_order=0, narrow_slices=1, phase_mask=2), dict(pb_order=0, narrow_slices=0, phase_mask=2),
                dict(narrow_slices=1), dict(narrow_slices=0)]
    if "c2x" in sets:
        opts = [dict(tie_sort=1), dict(tie_sort=0), dict(tie_sort=2), dict(narrow_slices=0), dict(pb_group=0), dict(tie_sort=1), dict(pb_order=0), dict(tie_sort=2, pb_order=0)]
    if "c2g" in sets:
        opts = [dict(pb_group=0), dict(pb_group_force=2), dict(pb_group_force=4), dict(pb_group_force=8), dict(pb_group=0), dict(pb_group_force=2), dict(pb_group_force=3)]
    if "bst" in sets:
        opts = [dict(pb_order=0, stage_burst=b, phase_mask=0) for b in (0, 2, 4, 8, 0, 2)]
    if "vec" in sets:
        opts = [dict(vec_blocks_per_cu=v) for v in (8, 4, 2, 1, 8, 4)]
    if "fuse" in sets:
        opts = [dict(fuse_staged=1), dict(fuse_staged=2), dict(fuse_staged=0)] * 4
    if "fuse2" in sets:
        opts = [dict(fuse_staged=2), dict(fuse_staged=2, pb_column_band=16384), dict(fuse_staged=1), dict(fuse_staged=0)] * 2
    if "unit2" in sets:
        opts = [dict(pb_unit=131072), dict(pb_unit=65536), dict(pb_unit=98304), dict(pb_unit=196608), dict(pb_unit=131072, pb_taper=0), dict(pb_unit=131072)]
    if "grp2" in sets:
        opts = [dict(pb_group=16384), dict(pb_group=0)] * 4
    if "deep" in sets:
        opts = [dict(spmv_deep=1), dict(spmv_deep=0), dict(spmv_deep=1), dict(spmv_deep=0)]
    if "st2" in sets:
        opts = [dict(pb_stamps=1), dict(pb_stamps=1, pb_persistent=2), dict(pb_stamps=1, pb_persistent=0), dict(), dict(pb_persistent=2), dict()]
    if "st" in sets:
        opts = [dict(pb_stamps=1), dict(pb_stamps=1, pb_order=0), dict(pb_stamps=1, pb_reduce=0), dict(pb_stamps=1, pb_reduce=128), dict(pb_stamps=1)]
    if "order" in sets:
        opts = [dict(), dict(pb_order=0), dict(pb_stamps=1), dict(pb_order=0, pb_stamps=1), dict(), dict(pb_order=0)]
    if "abl" in sets:
        opts = [dict(pb_persistent=0)]
    if "persist" in sets:
        opts = [dict(), dict(pb_persistent=0), dict(), dict(pb_persistent=0)]
    if "plain" in sets:
        opts = [dict(), dict(wgs_per_cu=3), dict(wgs_per_cu=4), dict(long_row=192), dict(long_row=96)]
    if "pbx" in sets:
        opts = [dict(propagation_blocking=1, hub_entries=h) for h in (4096, 8192, 12288, 16384, 17000, 19000)]
    if "pbdbg" in sets:
        opts = [dict(), dict(propagation_blocking=0), dict(propagation_blocking=1), dict(propagation_blocking=1, pb_target=16384),
                dict(propagation_blocking=1, pb_run_align=4), dict(propagation_blocking=1, pb_run_align=16)]
    if "er" in sets:   # the uniform family: blocked (every entry crosses as a value) vs plain gather, hub sizes, run formats
        opts = [dict(), dict(propagation_blocking=0), dict(propagation_blocking=0, hub_entries=0), dict(propagation_blocking=1, hub_entries=1024),
                dict(propagation_blocking=1, pb_reduce=0), dict(propagation_blocking=1, pb_column_band=8192)]
    if "pb" in sets:
        opts = [dict(propagation_blocking=0), dict(propagation_blocking=1), dict(propagation_blocking=1, hub_entries=8192),
                dict(propagation_blocking=1, hub_entries=19000)]
    if "hubk" in sets:
        opts = [dict(), dict(pb_reduce=0), dict(pb_reduce=256), dict(pb_reduce=768)]
    if "c2pb" in sets:
        opts = [dict(), dict(propagation_blocking=1), dict(propagation_blocking=1, pb_reduce=128), dict(propagation_blocking=1, pb_reduce=64), dict(propagation_blocking=1, pb_target=2048), dict(propagation_blocking=1, hub_entries=8192)]
    if "side" in sets:
        opts = [dict(), dict(side_stream=0), dict(), dict(side_stream=0)]
    if "tune2" in sets:
        opts = [dict(), dict(pb_target=4096), dict(pb_target=8192), dict(pb_target=32768), dict(pb_reduce=128), dict(pb_reduce=192), dict(pb_reduce=256), dict()]
    if "tgt" in sets:
        opts = [dict(), dict(pb_target=32768), dict(pb_target=65536), dict(pb_target=262144), dict(), dict(pb_target=65536)]
    if "iso" in sets:
        opts = [dict(), dict(phase_mask=3 + 4), dict(phase_mask=3 + 8), dict()]
    if "lr" in sets:
        opts = [dict(), dict(long_row=256), dict(long_row=512), dict(long_row=1024), dict(long_row=2048), dict(long_row=256)]
    if "unit" in sets:
        opts = [dict(), dict(pb_unit=131072), dict(pb_unit=262144), dict(), dict(pb_unit=131072), dict(pb_unit=262144), dict(pb_unit=32768)]
    if "taper" in sets:
        opts = [dict(), dict(pb_taper=0), dict(), dict(pb_taper=0)]
    if "c5x" in sets:
        opts = [dict(), dict(pb_taper=0), dict(pb_target=65536), dict(pb_target=32768), dict(long_row=128)]
    if "hubph" in sets:
        opts = [dict(), dict(phase_mask=1), dict(phase_mask=2), dict(long_row=512), dict(long_row=512, phase_mask=1), dict(long_row=512, phase_mask=2)]
    if "minrun" in sets:
        opts = [dict(), dict(pb_reduce=256), dict(pb_reduce=192), dict(pb_reduce=512), dict(), dict(pb_reduce=256)]
    if "one" in sets:
        opts = [dict(pb_reduce=0)]
    if "phase" in sets:
        opts = [dict(phase_mask=1), dict(phase_mask=2), dict(phase_mask=1, hub_entries=0), dict(phase_mask=2, hub_entries=0),
                dict(long_row=256), dict(long_row=4096), dict(long_row=65536)]
    if "fresh" in sets:
        # every configuration in its own process: engines created one after another in one process see different
        # physical memory layouts (a "fast" and a "slow" state about 5 % apart were observed), which drowns small A/B effects
        import json
        import subprocess
        for nm in names:
            for o in opts:
                subprocess.run([sys.executable, os.path.abspath(__file__), nm, "@one=" + json.dumps(o)])
        sys.exit(0)
    for a in sys.argv[1:]:
        if a.startswith("@one="):
            import json
            opts = [json.loads(a[5:])]
    for nm in names:
        run(nm, opts)
