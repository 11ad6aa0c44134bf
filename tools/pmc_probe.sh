#!/bin/bash
# tools/pmc_probe.sh <tag> <counter> [<counter> ...] -- one rocprofv3 --pmc pass (counters only, no tracing) over a short bench run
set -o pipefail
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 "$R/bench.py" --no-cpu-baseline --steps 10 --warmup 1 > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
ls "$OUT"
