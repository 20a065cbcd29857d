#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel statistics and the two HBM-traffic counter passes for the
# default bench workload, then summarises them.  Usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>/
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- \
  python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.log" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.log" || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/pmc_hbm_traffic.json" || exit 1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
echo done
