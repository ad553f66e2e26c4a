#!/bin/bash
# SQ counters of one arithmetic mode's kernels (two passes of 8 counters): tools/pmc_stream.sh <tag> <dtype>
set -o pipefail
TAG=$1; DT=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/layer_times.py" $DT > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d "$OUT/p2" -- python3 "$ROOT/tools/layer_times.py" $DT > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT/p1" "$OUT/p2" > "$OUT/summary.json"
rm -rf "$OUT/p1" "$OUT/p2"
echo done
