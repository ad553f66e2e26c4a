#!/usr/bin/env python3
"""bench.py - chunks/s of the squiggle-classification hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f32_direct|bf16|f16]

One "step" = one pass of the hot path over one batch of 512 synthetic RNA004 4 s chunks
(512 x 16000 int16 samples, BASELINE.json configs[1]) already resident in HBM:
MAD-normalise + 12-layer ConvNet forward + softmax, through the C ABI.  The default
arithmetic is fp32 end to end with the conv layers lowered to Winograd F(2,3) on the f32-input
MFMA (rs_dtype RS_F32W); --dtype f32_direct runs the direct lowering (exact fmaf chains).  With N > 1 (launched
by torch.distributed.run, one rank per GPU) every rank steps its own 512-read shard - reads
are independent, there is no data-path collective - and the printed value is the whole-job
aggregate over the max-over-ranks time ("scaling": "weak").

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      conv stack (layers 1..11, the MFMA kernel family) achieved TFLOP/s from HIP
                events recorded on the launch stream during the timed steps
  cpu_baseline  the oracle's torch-CPU port (reference structure: one read at a time) timed
                on this box's host cores over a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from riser_amd import dist as rdist          # noqa: E402
from riser_amd import synth                  # noqa: E402

SETTLE_STEPS = 25
BATCH = 512
CHUNK = 16000
SIG_SEED = 20260103
PEAK_F32_MFMA_TF = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md:42
PEAK_BF16_MFMA_TF = 2500.0    # :43
PEAK_HBM_GBS = 8000.0         # :36


def conv_flops_per_chunk(channels, L0):
    """Un-padded algorithmic FLOPs per chunk of each conv layer."""
    out = []
    c_in, L = 1, L0
    for c in channels:
        out.append(2.0 * c_in * c * 3 * L)
        c_in, L = c, L // 2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)    # ~60 ms: the shader clock needs ~20 steps to settle
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32_direct", "bf16", "f16"])
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--chunk", type=int, default=CHUNK)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-latency", action="store_true", help="skip the per-batch latency loop (profiling runs)")
    ap.add_argument("--no-variants", action="store_true", help="skip the f16 / bf16 / ensemble side measurements")
    args = ap.parse_args()

    rank, local_rank, world = rdist.env_world()
    if world != max(args.gpus, 1) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm device visible (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    device = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(device)
    rdist.init(device=device)

    from riser_amd.model import Model
    from riser_amd.preprocess import pack_reads

    B, L = args.batch, args.chunk
    lib_dtype = {"f32": "f32w", "f32_direct": "f32"}.get(args.dtype, args.dtype)
    model = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=lib_dtype, device=device)
    # each rank owns a different shard of the synthetic read population
    sigs = synth.make_signals(SIG_SEED, B, L, first_read=rank * B)
    sig, off, ln, lens = pack_reads(list(sigs), device)
    probs = torch.empty((B, 2), dtype=torch.float32, device=device)

    def step():
        model.classify_raw(sig, off, ln, lens, out=probs)

    # steady state is what the ReadUntil loop runs in: let the shader clock settle (~20 steps = 40 ms after an idle
    # period) before the W counted warm-up steps, whatever W the caller picked
    for _ in range(SETTLE_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)

    # ---- timed region: exactly K steps, barrier + sync on both sides ---------------------------
    # HIP events on the launch stream bracket the conv stack inside the timed steps (coarse level: 4 events per step;
    # one event per launch costs ~4.5 us of stream time each, 3 % of the step - that level runs in a separate pass below)
    model.profile(True, coarse=True)
    rdist.barrier(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    rdist.barrier(device)
    elapsed = time.perf_counter() - t0
    coarse_ms, calls = model.profile_read()
    model.profile(False)
    elapsed = rdist.reduce_scalar(elapsed, "max", device)
    total_chunks = rdist.reduce_scalar(B * args.steps, "sum", device)
    conv_ms_timed = float(coarse_ms[model.n_layers]) / max(calls, 1)        # layers 1..n-1, per step, from the timed region

    # per-launch detail (not timed): one event after every kernel
    model.profile(True)
    for _ in range(max(5, min(args.steps, 20))):
        step()
    torch.cuda.synchronize(device)
    stage_ms, calls = model.profile_read()
    model.profile(False)

    # ---- per-batch latency incl. H2D of the int16 batch and D2H of the probabilities -----------
    host_sig = torch.from_numpy(np.ascontiguousarray(sigs.reshape(-1))).pin_memory()
    host_probs = torch.empty((B, 2), dtype=torch.float32).pin_memory()
    lat = []
    n_lat = 0 if args.no_latency else max(30, min(200, args.steps * 5))
    for i in range(n_lat + 5 if n_lat else 0):
        t1 = time.perf_counter()
        sig.copy_(host_sig, non_blocking=True)
        step()
        host_probs.copy_(probs, non_blocking=True)
        torch.cuda.synchronize(device)
        if i >= 5:
            lat.append(time.perf_counter() - t1)
    lat_ms = np.asarray(lat if lat else [0.0]) * 1e3
    p50, p99 = float(np.percentile(lat_ms, 50)), float(np.percentile(lat_ms, 99))
    p99 = rdist.reduce_scalar(p99, "max", device)

    if rank != 0:
        rdist.finalize()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_chunks / elapsed

    # ---- roofline of the conv stack (the MFMA kernel family, layers 1..n-1) --------------------
    flops = conv_flops_per_chunk(model.channels, L)
    conv_ms_detail = float(stage_ms[2:2 + model.n_layers - 1].sum()) / max(calls, 1)   # per step, per-launch events
    conv_ms = conv_ms_timed if conv_ms_timed > 0 else conv_ms_detail             # the timed region's figure
    conv_flop = sum(flops[1:]) * B
    achieved_tf = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = PEAK_F32_MFMA_TF if args.dtype in ("f32", "f32_direct") else PEAK_BF16_MFMA_TF
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                traffic = json.load(f).get(lib_dtype, {}).get("conv_stack_hbm_bytes_per_step")
        except Exception:
            traffic = None
    info = model.layer_info()
    per_layer = []
    executed_flop = 0.0                    # MFMA FLOPs the kernels really issue: padded tiles, Winograd's 4-for-6
    for i in range(1, model.n_layers):
        ms = float(stage_ms[1 + i]) / max(calls, 1)
        P_in = model.padded_length(L) >> i
        rows = B * -(-P_in // info[i]["gemm_row_div"])                   # GEMM rows: groups of 2 / 4 conv rows for Winograd
        executed_flop += 2.0 * rows * info[i]["n_pad"] * info[i]["k_pad"]
        per_layer.append({"layer": i, "ms": round(ms, 4),
                          "tflops": round(flops[i] * B / (ms * 1e-3) / 1e12, 2) if ms > 0 else None,
                          "tile": [info[i]["bm"], info[i]["bn"], info[i]["kc"]]})
    executed_tf = executed_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    f23 = [str(i) for i in range(1, model.n_layers) if info[i]["gemm_row_div"] == 2]
    f43 = [str(i) for i in range(1, model.n_layers) if info[i]["gemm_row_div"] == 4]
    kname = {"f32w": "conv_stream_f32_kernel / conv_wino_kernel / conv_wino4_kernel (Winograd F(2,3) layers %s, F(4,3) layers %s, "
                     "f32-input MFMA)" % (",".join(f23), ",".join(f43)), "f32": "conv_f32_kernel (direct, f32-input MFMA)"}.get(
        lib_dtype, "conv_h16_kernel (%s MFMA)" % lib_dtype)
    roofline = {"bound": "mfma", "kernel": kname + ", 11 launches/step, layers 1-11",
                "achieved": round(achieved_tf, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved_tf / peak, 4), "traffic": traffic,
                "achieved_note": "algorithmic FLOPs of the direct convolution (SURVEY.md 8(d): 582.95 MFLOP per chunk in "
                                 "layers 1-11) / measured time; Winograd F(2,3) issues 2/3 of them on the matrix pipe, F(4,3) 1/2",
                "executed_mfma_tflops": round(executed_tf, 2), "executed_mfma_frac": round(executed_tf / peak, 4),
                "avg_launch_ms": round(conv_ms / (model.n_layers - 1), 4),
                "timing_note": "achieved / avg_launch_ms: HIP events around the conv stack inside the timed steps; stage_ms and "
                               "layers[]: a separate pass with one event per launch (each event adds ~4.5 us of stream time)",
                "conv_stack_ms_per_launch_events": round(conv_ms_detail, 4),
                "stage_ms": {"normalise": round(float(stage_ms[0]) / max(calls, 1), 4),
                             "conv0": round(float(stage_ms[1]) / max(calls, 1), 4),
                             "conv1_11": round(conv_ms_detail, 4),
                             "head": round(float(stage_ms[model.n_layers + 1]) / max(calls, 1), 4)},
                "layers": per_layer}

    out = {
        "metric": "signal chunks classified/sec (RNA004 4 s chunks, batch=512)",
        "value": round(value, 1), "unit": "chunks/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if lib_dtype in ("f32", "f32w") else lib_dtype,
        "data": "synthetic",
        "config": {"workload": f"mRNA RNA004 model, batch={B} x {L}-sample (4 s) int16 chunks resident in HBM, "
                               f"MAD-normalise + 12-layer ConvNet forward + softmax, {args.dtype}",
                   "conv_algorithm": {"f32w": "winograd_f23_f43_fp32", "f32": "direct_fp32"}.get(lib_dtype, "direct_" + lib_dtype),
                   "batch_per_gpu": B, "chunk_samples": L, "sharding": "reads by id across GPUs, no collectives"},
        "p50_batch_latency_ms": round(p50, 3), "p99_batch_latency_ms": round(p99, 3),
        "latency_note": "host wall time per 512-read batch incl. H2D of int16 signals from pinned memory and D2H of probabilities",
        "roofline": roofline,
    }

    # ---- side measurements on the same batch (rank 0, N = 1 only; not the headline) --------------
    if world == 1 and not args.no_variants and args.dtype == "f32":
        ref = probs.cpu().numpy().copy()
        variants = {}
        for dt in ("f32", "f16", "bf16"):
            mv = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype=dt, device=device)
            if dt in ("f16", "bf16"):
                mv.autotune(sig, off, ln, lens)      # optional per-geometry tile tuning (fp32: the planner's picks stand)
            pv = torch.empty((B, 2), dtype=torch.float32, device=device)
            for _ in range(max(2, args.warmup)):
                mv.classify_raw(sig, off, ln, lens, out=pv)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                mv.classify_raw(sig, off, ln, lens, out=pv)
            torch.cuda.synchronize(device)
            dtv = (time.perf_counter() - t1) / args.steps
            pvh = pv.cpu().numpy()
            variants["f32_direct" if dt == "f32" else dt] = {
                "chunks_per_s": round(B / dtv, 1), "ms_per_step": round(dtv * 1e3, 4),
                "max_abs_dprob_vs_f32": float(np.abs(pvh - ref).max()),
                "label_flips_at_0.9_vs_f32": int(((pvh[:, 1] > 0.9) != (ref[:, 1] > 0.9)).sum()),
                "batch": B}
            mv.close()
        # BASELINE config 3: three-model ensemble, bf16, normalise once + three forwards + decision
        from riser_amd.preprocess import Kit, SignalProcessor
        from riser_amd import _native as nv
        proc = SignalProcessor(Kit.create_from_version("RNA004"), device=device)
        ens = [Model(synth.make_state_dict(sd_), synth.Config(), None, t_, dtype="bf16", device=device)
               for sd_, t_ in ((1, "mRNA"), (2, "mtRNA"), (3, "globin"))]
        for mk in ens:
            mk.autotune(sig, off, ln, lens)
        pe = torch.empty((3, B, 2), dtype=torch.float32, device=device)
        dec = torch.empty(B, dtype=torch.uint8, device=device)
        from riser_amd.model import classify_raw_ensemble

        def ens_step():
            classify_raw_ensemble(ens, sig, off, ln, lens, out=pe, decision=dec, max_len=L, threshold=0.9,
                                  mode=nv.RS_ENRICH)
        for _ in range(max(3, args.warmup)):
            ens_step()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ens_step()
        torch.cuda.synchronize(device)
        dte = (time.perf_counter() - t1) / args.steps
        variants["ensemble3_bf16"] = {"reads_per_s": round(B / dte, 1), "model_forwards_per_s": round(3 * B / dte, 1),
                                      "ms_per_step": round(dte * 1e3, 4), "batch": B,
                                      "accepted": int((dec == 1).sum().item())}
        for mk in ens:
            mk.close()
        # BASELINE config 5: progressive 2 s / 3 s / 4 s chunks (8000 / 12000 / 16000 samples in equal
        # thirds of one batch), f16: per-read lengths are carried through every layer, no bucketing
        mix_lens = np.array([(8000, 12000, 16000)[i % 3] for i in range(B)], dtype=np.int32)
        mix_off = torch.from_numpy((np.arange(B, dtype=np.int64) * L)).to(device)
        mix_len = torch.from_numpy(mix_lens).to(device)
        mm = Model(synth.make_state_dict(1), synth.Config(), None, "mRNA", dtype="f16", device=device)
        pm = torch.empty((B, 2), dtype=torch.float32, device=device)
        mm.autotune(sig, mix_off, mix_len, mix_lens)
        for _ in range(max(3, args.warmup)):
            mm.classify_raw(sig, mix_off, mix_len, mix_lens, out=pm)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            mm.classify_raw(sig, mix_off, mix_len, mix_lens, out=pm)
        torch.cuda.synchronize(device)
        dtm = (time.perf_counter() - t1) / args.steps
        variants["mixed_2s_3s_4s_f16"] = {"chunks_per_s": round(B / dtm, 1), "ms_per_step": round(dtm * 1e3, 4),
                                          "batch": B, "samples_per_step": int(mix_lens.sum())}
        mm.close()
        out["variants"] = variants

    # ---- CPU baseline (rank 0, N = 1 only): oracle port timed on this box's host cores ---------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import torch_path
        cpu_model = torch_path.TorchCpuModel(synth.make_state_dict(1))
        torch_path.classify_per_read(cpu_model, sigs[:2])                    # warm-up
        n_done, t1 = 0, time.perf_counter()
        while n_done < B and time.perf_counter() - t1 < args.cpu_seconds:
            torch_path.classify_per_read(cpu_model, sigs[n_done:n_done + 8])
            n_done += 8
        dt = time.perf_counter() - t1
        # second figure: the same torch-CPU conv stack at batch 64 - what the reference's ops give
        # when its per-read loop is batched; normalisation stays per read (numpy), as the reference has no batched form
        from oracle import riser_oracle as ro
        nb, tb0 = 0, time.perf_counter()
        while nb < 128 and time.perf_counter() - tb0 < max(4.0, args.cpu_seconds / 2):
            xs = np.stack([ro.mad_normalise(s) for s in sigs[nb:nb + 64]]).astype(np.float32)
            torch.softmax(cpu_model.logits(torch.from_numpy(xs)), dim=1)
            nb += 64
        dtb = time.perf_counter() - tb0
        out["cpu_baseline"] = {"value": round(n_done / dt, 2), "unit": "chunks/s",
                               "batched64_value": round(nb / dtb, 2),
                               "batched64_note": f"{nb} chunks through the same torch-CPU ops at batch 64 (+ per-read numpy "
                                                 f"normalise), {dtb:.1f} s",
                               "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"first {n_done} of the step's {B} chunks ({L} samples each), one read at a "
                                         f"time: numpy MAD-normalise + torch-CPU conv stack at batch 1 "
                                         f"(structure of riser/control.py:63-69), {dt:.1f} s; host has {os.cpu_count()} logical CPUs"}
    print(json.dumps(out), flush=True)
    rdist.finalize()


if __name__ == "__main__":
    main()
