#!/bin/bash
# tools/profile_bench.sh <tag> [bench args...] -- run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of
# bench.py, then FETCH_SIZE and WRITE_SIZE in two more passes (MI355X_MICROARCH.md: counters in their own runs, never
# combined with tracing).  Results land in gpurun_out/prof_<tag>/; condense with tools/summarize_prof.py.
set -o pipefail
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_kt.json" 2> "$OUT/kt.err" || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" --no-cpu-baseline --steps 10 --warmup 1 "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" --no-cpu-baseline --steps 10 --warmup 1 "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err" || exit 1
ls "$OUT"
