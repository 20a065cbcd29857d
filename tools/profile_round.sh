#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel statistics, the two HBM-traffic counter passes and the SQ
# (matrix-pipe / stall) counter pass for one bench workload, then summarises them.
#   tools/profile_round.sh <tag> [workload] [git_head]   -> gpurun_out/<tag>/
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM and PMC-slot sections); the
# program itself follows `--` (no env/bash hop).
set -u
TAG=${1:-prof}
WL=${2:-synth16k_60s}
HEAD=${3:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log" || exit 1
echo "stats pass done"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.log" || exit 1
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.log" || exit 1
echo "write pass done"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o p -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_sq.log" || echo "SQ pass failed (see pmc_sq.log)"
echo "sq pass done"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_sq2" -o p -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_sq2.log" || echo "SQ2 pass failed (see pmc_sq2.log)"
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" "$WL" "$HEAD" > "$OUT/pmc_hbm_traffic.json" || exit 1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_sq2"
echo done
