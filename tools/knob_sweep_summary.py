"""tools/knob_sweep_summary.py <file> -- means of tools/knob_sweep.sh's SpMV averages per configuration"""
import collections
import re
import sys

d = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"(\w+) (\{.*?\}) (.*?)\| spmv avg ([\d.]+) ms min ([\d.]+)", l)
    if m:
        d.setdefault((m.group(1), m.group(2)), []).append((float(m.group(4)), m.group(3).strip()))
base = None
for (w, cfg), v in d.items():
    xs = [x for x, _ in v]
    mean = sum(xs) / len(xs)
    if cfg == "{}" and base is None:
        base = mean
    rel = f" ({(mean / base - 1) * 100:+.1f} %)" if base else ""
    print(f"{w:4s} {cfg:48s} {[round(x, 4) for x in xs]} mean {mean:.4f}{rel}   {v[0][1][:90]}")
