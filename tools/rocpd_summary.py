#!/usr/bin/env python3
"""Summaries of rocprofv3's rocpd (sqlite) outputs.

  rocpd_summary.py stats  <results.db>                 -> CSV of per-kernel duration statistics (ns)
  rocpd_summary.py pmc    <fetch.db> <write.db> STEPS  -> JSON: per kernel mean FETCH_SIZE / WRITE_SIZE (KB) per
                                                          dispatch, and conv-stack HBM bytes per step with FETCH_SIZE
                                                          doubled as MI355X_MICROARCH.md prescribes for gfx950
"""
import json, math, sqlite3, sys, collections


def short(name):
    return name.replace("void rs::(anonymous namespace)::", "").replace("rs::(anonymous namespace)::", "").split("(")[0]


def stats(db):
    c = sqlite3.connect(db)
    d = collections.defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        d[name].append(dur)
    tot = sum(sum(v) for v in d.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
    for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / n)
        print('"%s",%d,%d,%.3f,%.2f,%d,%d,%.3f' % (name, n, s, mean, 100.0 * s / tot, min(v), max(v), sd))


def pmc_means(db, counter):
    c = sqlite3.connect(db)
    d = collections.defaultdict(list)
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        d[short(name)].append(val)
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


def pmc(fdb, wdb, steps):
    f, w = pmc_means(fdb, "FETCH_SIZE"), pmc_means(wdb, "WRITE_SIZE")
    out = {"kernels": {}, "steps_profiled": steps}
    total = 0.0
    for k in sorted(set(f) | set(w)):
        fk, n = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        per_step = n / steps
        hbm = (2.0 * fk + wk) * 1024.0 * per_step
        out["kernels"][k] = {"launches_per_step": round(per_step, 2), "fetch_kb_raw": round(fk, 1),
                             "write_kb": round(wk, 1), "hbm_bytes_per_step": int(hbm)}
        if k.startswith(("conv_wino_kernel", "conv_wino4_kernel", "conv_f32_kernel", "conv_ring_h16_kernel", "conv_wres_h16_kernel", "conv_small_f32_kernel", "conv_stream_")):
            total += hbm
    out["conv_stack_hbm_bytes_per_step"] = int(total)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]))
