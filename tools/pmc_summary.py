#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch).
usage: pmc_summary.py <dir> [<dir> ...]  -> prints JSON {kernel: {counter: mean}}"""
import csv, glob, json, os, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                k = k.replace("void rs::(anonymous namespace)::", "").split("(")[0]
                out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in out.items()}
# which kernels these counters belong to: bench.py reports `roofline.traffic` only while the tree's kernel sources still
# hash to this value (riser_amd/supervise.py has no torch import; neither does this)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    from riser_amd.build import csrc_sha16
    res["_meta"] = {"csrc_sha16": csrc_sha16()}
except Exception as e:                                   # noqa: BLE001 - a summary without the stamp is still a summary
    res["_meta"] = {"csrc_sha16": None, "error": str(e)}
print(json.dumps(res, indent=1, sort_keys=True))
