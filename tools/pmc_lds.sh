#!/bin/bash
# LDS bank conflicts of one arithmetic mode's kernels: tools/pmc_lds.sh <tag> <dtype>   (counters only, no trace domain)
set -o pipefail
TAG=$1; DT=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_lds_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -- python3 "$ROOT/bench.py" --dtype $DT --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-variants --no-control-loop > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT/p1" > "$OUT/summary.json"
rm -rf "$OUT/p1"
echo done
