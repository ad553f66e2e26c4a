#!/bin/bash
# rocprofv3 evidence for one arithmetic mode of bench.py (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag> <dtype> [extra bench args]
# writes gpurun_out/prof_<tag>/{bench.json, kernel_stats.csv, pmc_mfma.json, pmc_fetch_write.json}.
# Kernel trace and every PMC set are SEPARATE runs (counters never share a run with trace domains).
set -o pipefail
TAG=$1; DT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--dtype $DT --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-variants --no-control-loop $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
cp "$ROOT"/gpurun_out/bench_detail_*_"$DT".json "$OUT/bench_detail.json" 2>/dev/null   # the traced run's verbose objects (per-layer tile shapes)
echo "[profile] trace done" 
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_mfma" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_mfma.err" || { tail -5 "$OUT/pmc_mfma.err"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_mfma" > "$OUT/pmc_mfma.json"
echo "[profile] mfma counters done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
echo "[profile] fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/pmc_fetch_write.json"
rm -rf "$OUT/trace" "$OUT/pmc_mfma" "$OUT/pmc_fetch" "$OUT/pmc_write"
echo "[profile] done: $OUT"
