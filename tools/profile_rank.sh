#!/bin/bash
# tools/profile_rank.sh <tag> <world> [c3|c5] -- rocprofv3 kernel trace + stats of tools/rank_share.py (rank 0's share of a
# <world>-rank run, alone on the GPU), then FETCH_SIZE and WRITE_SIZE in two more passes (counters in their own runs, the program
# itself straight behind `--`).  Results in gpurun_out/rank_<tag>/; condense with tools/summarize_prof.py <dir> <tag>.
set -o pipefail
TAG=${1:?tag}; W=${2:?world}; WL=${3:-c3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/rank_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/tools/rank_share.py" "$W" "$WL" > "$OUT/share_kt.txt" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/tools/rank_share.py" "$W" "$WL" > "$OUT/share_fetch.txt" 2> "$OUT/fetch.err" || { tail -5 "$OUT/fetch.err"; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/tools/rank_share.py" "$W" "$WL" > "$OUT/share_write.txt" 2> "$OUT/write.err" || { tail -5 "$OUT/write.err"; exit 1; }
cat "$OUT/share_kt.txt"
