#!/bin/bash
# tools/knob_sweep.sh <workload> <out file> <json config> ... -- parity-controlled A/B of engine knobs (GPU box): every configuration is run in
# TWO consecutive processes (consecutive processes alternate between the two process states of the SpMV, profiles/r4_state_probe.txt), the whole list
# ROUNDS times (default 2); tools/knob_sweep_summary.py prints the means.
W=$1; O=$2; shift 2
cd ${GRAFT_REPO_ROOT:-.}
: > $O
for rep in $(seq 1 ${ROUNDS:-2}); do
for cfg in "$@"; do
  for twice in a b; do
  timeout -k 10 300 python tools/perf_probe.py $W "@one=$cfg" 2>&1 | grep "spmv avg" | sed 's/gen=.*long=/long=/' | sed 's/padded=[0-9]* hub=[0-9]* //' | cut -c1-260 >> $O || exit 1
  done
done
done
