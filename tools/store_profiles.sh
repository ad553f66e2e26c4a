#!/bin/bash
# copy the outputs of tools/profile_round.sh r06_f32 f32 / r06_bf16x3 bf16x3 / r06_f16xf8 f16xf8 (gpurun_out/prof_r06_*) to their names under profiles/
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=${1:-r06}
for m in f32 bf16x3 f16xf8; do
  d=$ROOT/gpurun_out/prof_${R}_$m
  cp "$d/kernel_stats.csv" "$ROOT/profiles/${R}_kernel_stats_$m.csv"
  cp "$d/pmc_mfma.json" "$ROOT/profiles/${R}_pmc_mfma_busy_$m.json"
  cp "$d/pmc_fetch_write.json" "$ROOT/profiles/${R}_pmc_fetch_write_$m.json"
  cp "$d/bench.json" "$ROOT/profiles/${R}_bench_${m}_under_rocprof.json"
  cp "$d/bench_detail.json" "$ROOT/profiles/${R}_bench_detail_${m}_under_rocprof.json"
done
python3 - <<PY
import json, sys
sys.path.insert(0, "$ROOT")
from riser_amd.build import csrc_sha16
print("tree", csrc_sha16(), "profiles", json.load(open("$ROOT/profiles/${R}_pmc_fetch_write_f32.json"))["_meta"])
PY
