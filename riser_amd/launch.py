"""Sharded live control loop: N processes, N channel ranges, N GPUs, no collective.

    python -m riser_amd.launch --gpus N --channels C --kit RNA004 --mode enrich --out run \\
        [--client pkg.module:factory] [--model-dir model --targets mRNA,mtRNA | --model target=state.pth ...]
        [--dtype f32w] [--duration-h 48] [--threshold 0.9] [--on-rank-failure abort|restart] [--max-restarts 2]
        [--replay-script tests/golden/control.json | --replay-synthetic BATCHES] [--seeds 1,2,3] [--share-gpus] [--stub]

The reference drives ONE flow cell from one process: `ReadUntilClient.run(first_channel=1, last_channel=512)` and one
`SequencerControl.target` loop (riser/client.py:33-38, riser/control.py:25-106).  A PromethION-scale cell (BASELINE config
4: 144 k concurrent reads, 18 k per GPU) is host-bound long before it is GPU-bound - one Python loop cannot feed eight
GPUs - and reads are independent, so the scale-out is pure partitioning of the CHANNELS, which is the one thing the
ReadUntil API lets a client choose:

  * the launcher starts N fresh child processes BEFORE anything touches the GPU (RANK / LOCAL_RANK / WORLD_SIZE in the
    environment, as torchrun sets them);
  * rank r builds its client on channels [r C/N + 1, (r+1) C/N] (`rank_channel_range`: the reference's own
    first_channel / last_channel), its models on LOCAL_RANK's device, and runs the unchanged batched SequencerControl,
    writing `<out>.rank<r>.csv`;
  * nothing is exchanged between ranks.  Each child reports its once-a-minute counters (riser/control.py:116-123) and a
    final summary as JSON lines on its stdout; the parent merges them into one progress line per minute and one summary
    (`<out>.summary.json`).

Models: `--model-dir DIR --targets mRNA,mtRNA --kit RNA004` resolves `DIR/{target}_config_{kit}_{pore}.yaml` and
`DIR/{target}_model_{kit}_{pore}.pth` exactly as the reference's `get_models` does (riser/riser.py:26-42,
riser_amd/modeldir.py); `--model target=state.pth` takes a state dict with the shipped 12-layer config.

Supervision (riser_amd/supervise.py): the parent POLLS its children.  A rank that exits non-zero is logged at once with
the tail of its stderr; `--on-rank-failure abort` (default) then terminates the other ranks and exits non-zero,
`restart` starts a FRESH process for that channel range (at most `--max-restarts` times per rank; never a re-exec of a
process that has touched the GPU) so that the other ranges stay under control meanwhile.  The merged per-minute line is
printed over the ranks that are alive and names the ones that are not.  Every rank runs on its own slice of the host's
cores, its torch / hostpack thread pools sized to the slice.

`--client pkg.module:factory` names a callable `factory(logger, first_channel, last_channel)` returning an object with the
eight client methods (riser/client.py:25-69).  Without it the launcher replays scripted traffic through FakeClient -
`--replay-script` (the format of tests/golden/control.json) or `--replay-synthetic` (riser_amd.replay.scripted_batches) -
with weights from riser_amd.synth (`--seeds`): that is how the tests and bench.py exercise it.  `--stub` replaces the GPU
work by a host-only pass (CPU rehearsal of the rank plumbing).
"""
from __future__ import annotations

import argparse
import importlib
import json
import logging
import os
import re
import subprocess
import sys
import threading
import time

from . import supervise

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # the directory that holds the package
_MINUTE_RE = re.compile(r"In the last minute (\d+) signals were assessed, (\d+) were accepted and (\d+) were rejected")


def rank_channel_range(rank: int, world: int, channels: int):
    """(first_channel, last_channel), 1-based and inclusive as ReadUntilClient.run takes them (riser/client.py:33-38):
    contiguous ranges that partition 1..channels, sizes differing by at most one."""
    if not 0 <= rank < world or channels < world:
        raise ValueError(f"rank {rank} of {world} over {channels} channels")
    base, extra = divmod(channels, world)
    first = rank * base + min(rank, extra) + 1
    last = first + base + (1 if rank < extra else 0) - 1
    return first, last


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="python -m riser_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--kit", default="RNA004", choices=["RNA002", "RNA004"])
    ap.add_argument("--mode", default="enrich", choices=["enrich", "deplete"])
    ap.add_argument("--threshold", type=float, default=0.9)
    ap.add_argument("--duration-h", type=float, default=48.0)
    ap.add_argument("--unblock-duration", type=float, default=0.1)
    ap.add_argument("--out", default="riser_amd_run")
    ap.add_argument("--dtype", default="f32w")
    ap.add_argument("--client", default=None, help="pkg.module:factory(logger, first_channel, last_channel)")
    ap.add_argument("--model", action="append", default=[], help="target=state.pth (repeatable; the shipped 12-layer config)")
    ap.add_argument("--model-dir", default=None, help="directory laid out as the reference's model/: {target}_config_{kit}_{pore}.yaml "
                                                     "+ {target}_model_{kit}_{pore}.pth (riser/riser.py:35-42)")
    ap.add_argument("--targets", default="mRNA", help="comma list of targets to load from --model-dir")
    ap.add_argument("--on-rank-failure", default="abort", choices=["abort", "restart"])
    ap.add_argument("--max-restarts", type=int, default=2)
    ap.add_argument("--fail-rank", type=int, default=None, help=argparse.SUPPRESS)        # tests: this rank exits 3 ...
    ap.add_argument("--fail-after-s", type=float, default=0.0, help=argparse.SUPPRESS)    # ... this long after it started
    ap.add_argument("--fail-once", default=None, help=argparse.SUPPRESS)                  # ... only while this marker file is absent
    ap.add_argument("--seeds", default="1", help="synthetic weights for the replay modes: comma list of riser_amd.synth seeds")
    ap.add_argument("--replay-script", default=None)
    ap.add_argument("--replay-synthetic", type=int, default=0, metavar="BATCHES")
    ap.add_argument("--share-gpus", action="store_true", help="allow more ranks than visible devices: rank r runs on device r mod devices (2-3 channel ranges per GPU fill the device while each rank's host code runs; also: rehearsal on one GPU)")
    ap.add_argument("--no-signal-cache", action="store_true")
    ap.add_argument("--stub", action="store_true", help="no GPU: every received read becomes a 'try_again' CSV row")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# parent: spawn, relay, merge
# ---------------------------------------------------------------------------------------------------------------
def launch(args, argv) -> dict:
    """Start args.gpus child ranks, relay their JSON lines, merge minute counters and summaries.  The parent never
    initialises the GPU (torch.cuda.device_count() does not, on this image)."""
    n = args.gpus
    if n < 1:
        raise SystemExit("--gpus must be >= 1")
    if not args.stub and not args.share_gpus:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < n:
            raise SystemExit(f"riser_amd.launch: --gpus {n} but only {ndev} ROCm device(s) visible (--share-gpus rehearses "
                             "several ranks on one GPU)")
    log = logging.getLogger("riser_amd.launch")
    # one clock for the whole launch: every rank - a restarted one too - numbers its minutes from the parent's start and
    # stops at the parent's deadline (a rank restarted at hour 47 of 48 runs for one hour, not for another 48)
    t_start = time.time()
    deadline = t_start + args.duration_h * 3600.0

    def spawn(r):
        env = supervise.rank_env(r, n)
        env["RS_LAUNCH_T0"] = repr(t_start)
        env["RS_LAUNCH_DEADLINE"] = repr(deadline)
        env["PYTHONPATH"] = _ROOT + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        return subprocess.Popen([sys.executable, "-m", "riser_amd.launch", *argv], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, cwd=os.getcwd())

    minutes, summaries, lock = {}, {}, threading.Lock()
    down, failures = set(), []

    def merged_line(minute, final=False):
        """one line per minute over the ranks that reported it; called with the lock held"""
        slot = minutes.get(minute)
        if not slot or slot.get("printed"):
            return
        alive = [r for r in range(n) if r not in down]
        have = [r for r in alive if r in slot]
        if not final and len(have) < len(alive):
            return
        slot["printed"] = True
        tot = [sum(slot[r][k] for r in slot if isinstance(r, int)) for k in ("assessed", "accepted", "rejected")]
        missing = sorted(set(range(n)) - {r for r in slot if isinstance(r, int)})
        log.info(f"In the last minute {tot[0]} signals were assessed, {tot[1]} were accepted and {tot[2]} "
                 f"were rejected ({n - len(missing)} of {n} ranks" +
                 (f"; no report from rank(s) {missing}: their channels are not under control" if missing else "") + ")")

    def on_line(rank, line):
        line = line.strip()
        if not line.startswith("{"):
            if line:
                print(f"[rank {rank}] {line}", file=sys.stderr, flush=True)
            return
        try:
            msg = json.loads(line)
        except ValueError:
            return
        with lock:
            if msg.get("kind") == "minute":
                minutes.setdefault(msg["minute"], {})[rank] = msg
                merged_line(msg["minute"])
            elif msg.get("kind") == "summary":
                summaries[rank] = msg

    def on_event(kind, rank, rc, tail):
        first, last = rank_channel_range(rank, n, args.channels)
        with lock:
            if kind == "failed":
                down.add(rank)
                failures.append({"rank": rank, "returncode": rc, "channels": [first, last], "time": time.time(),
                                 "stderr_tail": tail[-10:]})
                log.error(f"rank {rank} (channels {first}-{last}) exited with code {rc}: its channels are not under control "
                          f"(its stderr: the lines tagged [rank {rank}] above; the last of them are kept in the summary)")
                for m in list(minutes):                        # minutes that were waiting for this rank only
                    merged_line(m)
            elif kind == "restarted":
                down.discard(rank)
                log.warning(f"rank {rank} (channels {first}-{last}) started again as a fresh process")

    restarts = args.max_restarts if args.on_rank_failure == "restart" else 0
    try:
        supervise.supervise(spawn, n, on_line, "riser_amd.launch", restarts=restarts, on_event=on_event, quiet_stderr=False)
    finally:
        with lock:
            for m in list(minutes):
                merged_line(m, final=True)
    merged = {"ranks": n, "channels": args.channels,
              "channel_ranges": [list(rank_channel_range(r, n, args.channels)) for r in range(n)],
              "per_rank": [summaries.get(r, {}) for r in range(n)],
              "rank_failures": failures,
              "minutes_merged": {str(k): {f: sum(m[f] for r, m in v.items() if isinstance(r, int))
                                          for f in ("assessed", "accepted", "rejected")}
                                 for k, v in sorted(minutes.items())}}
    for f in ("batches", "reads_received", "reads_assessed", "rejected", "finished"):
        merged[f] = sum(int(s.get(f, 0)) for s in summaries.values())
    merged["csv_files"] = [f"{args.out}.rank{r}.csv" for r in range(n)]
    with open(f"{args.out}.summary.json", "w") as f:
        json.dump(merged, f, indent=1)
    return merged


# ---------------------------------------------------------------------------------------------------------------
# child: one rank
# ---------------------------------------------------------------------------------------------------------------
class _MinuteRelay(logging.Handler):
    """turns the control loop's once-a-minute log line into a JSON line for the parent"""

    def __init__(self, rank, t0=None):
        super().__init__(level=logging.INFO)
        self.rank, self.last = rank, -1
        self.t0 = time.time() if t0 is None else float(t0)           # the LAUNCH's start (RS_LAUNCH_T0), not this process's

    def minute_index(self, now=None):
        """the launch-wide minute a report made `now` closes: minute k covers [60 k, 60 (k + 1)) s after the launch started;
        strictly increasing per process"""
        k = max(0, int(((time.time() if now is None else now) - self.t0) / 60.0 + 0.5) - 1)
        self.last = max(self.last + 1, k)
        return self.last

    def emit(self, record):
        m = _MINUTE_RE.search(record.getMessage())
        if m:
            print(json.dumps({"kind": "minute", "rank": self.rank, "minute": self.minute_index(), "assessed": int(m.group(1)),
                              "accepted": int(m.group(2)), "rejected": int(m.group(3))}), flush=True)


def _replay_batches(args):
    from .fake_client import FakeRead
    from . import synth
    if args.replay_script:
        with open(args.replay_script) as f:
            g = json.load(f)
        seed = int(g.get("raw_seed", 77))
        return [[(ch, FakeRead(rid_s, synth.make_raw_read(seed, rid, n, bool(polya)), number))
                 for ch, rid_s, rid, n, polya, number in b] for b in g["script"]]
    from .replay import scripted_batches
    return scripted_batches(args.replay_synthetic, args.channels)


class _StubControl:
    """--stub: the loop's client calls and CSV schema without a GPU - every read a client delivers becomes a row with
    decision try_again (tests of the launcher's plumbing on CPU)."""

    def __init__(self, client, logger, out_file):
        self.client, self.logger, self.out_filename = client, logger, out_file
        self.batch_latencies = []

    def start(self):
        self.client.start_streaming_reads()

    def finish(self):
        self.client.reset()

    def target(self, mode, duration_h, threshold, unblock_duration=0.1):
        from .control import _CSV_COLUMNS
        n = 0
        with open(f"{self.out_filename}.csv", "a") as sink:
            sink.write(",".join(_CSV_COLUMNS) + "\n")
            while self.client.is_running():
                t0 = time.monotonic()
                for channel, read in self.client.get_read_batch():
                    sink.write(f"{t0:.0f},{read.id},{channel},{len(read.raw_data) // 2},stub,0.0,{threshold},{mode},try_again\n")
                    n += 1
                self.client.reject_reads([], unblock_duration)
                self.client.finish_processing_reads([])
                self.batch_latencies.append(time.monotonic() - t0)
        self.logger.info(f"In the last minute {n} signals were assessed, 0 were accepted and 0 were rejected")


def remaining_duration_h(duration_h, now=None, env=None):
    """hours this rank still has to run: the launch's deadline (RS_LAUNCH_DEADLINE, set once by the parent) caps --duration-h,
    so a rank that is started again late in a run does not keep the launch alive for a second full duration"""
    dl = (os.environ if env is None else env).get("RS_LAUNCH_DEADLINE")
    if not dl:
        return duration_h
    left = (float(dl) - (time.time() if now is None else now)) / 3600.0
    return max(0.0, min(float(duration_h), left))


def _arm_test_failure(args, rank):
    """tests only (--fail-rank): this rank dies with exit code 3, at once or from a timer thread mid-run"""
    if args.fail_rank is None or args.fail_rank != rank:
        return
    if args.fail_once:
        if os.path.exists(args.fail_once):
            return                                   # the restarted process runs to completion
        open(args.fail_once, "w").close()

    def die():
        print(f"rank {rank}: simulated failure (--fail-rank)", file=sys.stderr, flush=True)
        os._exit(3)
    if args.fail_after_s <= 0:
        die()
    threading.Timer(args.fail_after_s, die).start()


def run_rank(args) -> int:
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["WORLD_SIZE"])
    _arm_test_failure(args, rank)
    first, last = rank_channel_range(rank, world, args.channels)
    logger = logging.getLogger(f"riser_amd.rank{rank}")
    logger.setLevel(logging.INFO)
    logger.addHandler(_MinuteRelay(rank, os.environ.get("RS_LAUNCH_T0")))
    out = f"{args.out}.rank{rank}"
    if args.client:
        mod, _, fn = args.client.partition(":")
        client = getattr(importlib.import_module(mod), fn)(logger, first, last)
    else:
        from .fake_client import FakeClient
        client = FakeClient(_replay_batches(args), first_channel=first, last_channel=last)
    if args.stub:
        ctl = _StubControl(client, logger, out)
    else:
        import torch
        supervise.apply_torch_threads()
        from . import synth
        from .control import SequencerControl
        from .model import Model
        from .preprocess import Kit, SignalProcessor
        ndev = torch.cuda.device_count()
        if ndev < 1 or (ndev < world and not args.share_gpus):
            raise SystemExit(f"rank {rank}: {world} ranks but {ndev} ROCm device(s) visible")
        device = torch.device("cuda", local_rank % ndev)
        torch.cuda.set_device(device)
        if args.model_dir:
            from .modeldir import get_models
            models = get_models([t for t in args.targets.split(",") if t], logger, args.kit, args.model_dir,
                                dtype=args.dtype, device=device)
        elif args.model:
            spec = [m.partition("=")[::2] for m in args.model]
            models = [Model(path, synth.Config(), logger, target, dtype=args.dtype, device=device) for target, path in spec]
        else:
            names = ("mRNA", "mtRNA", "globin")
            models = [Model(synth.make_state_dict(int(s)), synth.Config(), logger, names[k % 3], dtype=args.dtype, device=device)
                      for k, s in enumerate(args.seeds.split(","))]
        proc = SignalProcessor(Kit.create_from_version(args.kit), device=device)
        ctl = SequencerControl(client, models, proc, logger, out, signal_cache=not args.no_signal_cache)
        ctl.reserve(max(512, last - first + 1))
    t0 = time.perf_counter()
    ctl.start()
    ctl.target(args.mode, remaining_duration_h(args.duration_h), args.threshold, args.unblock_duration)
    ctl.finish()
    wall = time.perf_counter() - t0
    with open(out + ".csv") as f:
        rows = sum(1 for _ in f) - 1
    lat = sorted(ctl.batch_latencies)
    summary = {"kind": "summary", "rank": rank, "first_channel": first, "last_channel": last, "csv": out + ".csv",
               "reads_assessed": rows, "wall_s": round(wall, 3), "batches": len(getattr(client, "rejected", [])),
               "reads_received": sum(len([e for e in b if first <= e[0] <= last]) for b in getattr(client, "_batches", [])),
               "rejected": sum(len(r) for r in getattr(client, "rejected", [])),
               "finished": sum(len(r) for r in getattr(client, "finished", [])),
               "rejected_lists": getattr(client, "rejected", None) if args.replay_script else None,
               "finished_lists": getattr(client, "finished", None) if args.replay_script else None,
               "p50_ms": round(lat[len(lat) // 2] * 1e3, 3) if lat else None,
               "max_ms": round(lat[-1] * 1e3, 3) if lat else None, "latency_samples": len(lat)}
    print(json.dumps(summary), flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, stream=sys.stderr, format="%(asctime)s %(name)s %(message)s")
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:
        supervise.apply_rank_limits()                # core slice first: torch sizes its pools when it is imported
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise SystemExit(f"riser_amd.launch: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
        return run_rank(args)
    merged = launch(args, argv)
    print(json.dumps({k: v for k, v in merged.items() if k != "per_rank"}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
