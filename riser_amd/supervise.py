"""Rank supervision shared by `bench.py --gpus N` and `python -m riser_amd.launch`: fresh children, polled.

Both launchers start N child processes before anything touches the GPU (one rank per GPU, nothing re-executed).  What
this module adds is what an unattended 8-GPU run needs:

  * `rank_env`: RANK / LOCAL_RANK / WORLD_SIZE (+ MASTER_*), a DISJOINT slice of the parent's cores per rank
    (`RS_CPU_SLICE`, applied by the child with `apply_rank_limits()` before torch starts its pools) and thread caps for
    torch / OpenMP / the `_hostpack` copy threads sized to that slice - eight ranks on a 256-CPU host otherwise start
    eight default-sized pools on the same cores;
  * `supervise`: the children are POLLED.  The first rank that exits non-zero is reported at once with the tail of its
    stderr; its siblings are terminated (SIGTERM, then SIGKILL) or - for a live run that must keep its other channel
    ranges under control - the rank is started again as a fresh process.  A rank that dies before the first barrier can
    therefore not leave rank 0 waiting in `init_process_group` until some outer timeout.

No torch import here: the parent must stay off the GPU, and the child applies its limits before importing torch.
"""
from __future__ import annotations

import collections
import os
import signal
import subprocess
import sys
import threading
import time

STDERR_TAIL_LINES = 40


def _parse_cpulist(text: str) -> list:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def read_topology(sys_root: str = "/sys"):
    """(gpu_nodes, node_cpus) from sysfs, or None when the map is not readable: gpu_nodes[k] = NUMA node of the k-th AMD GPU
    (DRM cards of vendor 0x1002 in PCI-address order: the order HIP enumerates them in on a default install), node_cpus =
    {node: [cpu, ...]}.  A node of -1 (no affinity reported) makes the caller fall back."""
    import glob
    try:
        cards = []
        for dev in glob.glob(os.path.join(sys_root, "class/drm/card[0-9]*/device")):
            if not os.path.basename(os.path.dirname(dev)).replace("card", "").isdigit():
                continue
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read().strip())
            cards.append((os.path.basename(os.path.realpath(dev)), node))
        node_cpus = {}
        for nd in glob.glob(os.path.join(sys_root, "devices/system/node/node[0-9]*")):
            with open(os.path.join(nd, "cpulist")) as f:
                node_cpus[int(os.path.basename(nd)[4:])] = _parse_cpulist(f.read())
        if not cards or not node_cpus:
            return None
        return [n for _, n in sorted(cards)], node_cpus
    except (OSError, ValueError):
        return None


def numa_cpu_slices(world: int, cpus, gpu_nodes, node_cpus):
    """slices that keep rank r on the cores of GPU r's NUMA node: the ranks whose GPUs share a node split that node's share of
    `cpus` between them.  None when the map does not cover every rank (fewer GPUs than ranks, a node of -1, a node without
    usable cores): the caller falls back to contiguous slices."""
    if gpu_nodes is None or len(gpu_nodes) < world:
        return None
    allowed = set(cpus)
    by_node = {}
    for r in range(world):
        by_node.setdefault(gpu_nodes[r], []).append(r)
    out = [None] * world
    for node, ranks in by_node.items():
        mine = [c for c in node_cpus.get(node, []) if c in allowed]
        if node < 0 or len(mine) < len(ranks):
            return None
        base, extra = divmod(len(mine), len(ranks))
        at = 0
        for i, r in enumerate(ranks):
            k = base + (1 if i < extra else 0)
            out[r] = mine[at: at + k]
            at += k
    return out


def cpu_slices(world: int, cpus=None, topology="auto") -> list:
    """`world` disjoint slices of the cores this process may run on.  When the GPU -> NUMA node map is readable
    (`read_topology`; `topology` = a (gpu_nodes, node_cpus) pair overrides, None disables) rank r gets cores of GPU r's node;
    otherwise contiguous slices, sizes differing by at most one.  With fewer cores than ranks every rank gets the whole set
    (a rehearsal box)."""
    explicit = cpus is not None
    if cpus is None:
        cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cpus = list(cpus)
    if world < 1:
        raise ValueError("world must be >= 1")
    if len(cpus) < world:
        return [list(cpus) for _ in range(world)]
    if topology == "auto":
        topology = None if (explicit or os.environ.get("RS_NUMA_SLICES") == "0") else read_topology()
    if topology:
        sl = numa_cpu_slices(world, cpus, topology[0], topology[1])
        if sl:
            return sl
    base, extra = divmod(len(cpus), world)
    out, at = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append(cpus[at: at + n])
        at += n
    return out


def rank_env(rank: int, world: int, base_env=None, master_port=None, cpus=None) -> dict:
    """Environment of child rank `rank` (torchrun's variable names), with its core slice and thread caps."""
    env = dict(os.environ if base_env is None else base_env)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    if master_port is not None:
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sl = cpu_slices(world, cpus)[rank]
    env["RS_CPU_SLICE"] = ",".join(str(c) for c in sl)
    nthr = max(1, min(len(sl), 32))
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        try:                                                   # an operator's own (smaller) setting stands
            env[k] = str(max(1, min(nthr, int(env[k])))) if env.get(k) else str(nthr)
        except ValueError:
            env[k] = str(nthr)
    env["RS_TORCH_THREADS"] = str(nthr)
    env["RS_HOST_THREADS"] = str(max(1, min(8, len(sl))))          # riser_amd/csrc/hostpack.c: copy threads per gather
    return env


def apply_rank_limits() -> dict:
    """Child side, BEFORE torch is imported: pin this process to its slice (`RS_CPU_SLICE`).  Returns what was applied.
    The thread caps travel as OMP_NUM_THREADS / RS_TORCH_THREADS / RS_HOST_THREADS and are read by torch, by
    `apply_torch_threads()` and by `_hostpack` themselves."""
    done = {}
    sl = os.environ.get("RS_CPU_SLICE")
    if sl and hasattr(os, "sched_setaffinity"):
        try:
            cores = {int(c) for c in sl.split(",") if c != ""}
            if cores:
                os.sched_setaffinity(0, cores)
                done["cpus"] = len(cores)
        except (OSError, ValueError) as e:                         # a slice that is not ours to take: run unpinned
            done["cpus_error"] = str(e)
    return done


def apply_torch_threads():
    """Child side, after torch is imported: the intra-op pool follows the slice (the inter-op pool is not used here)."""
    n = os.environ.get("RS_TORCH_THREADS")
    if n:
        import torch
        torch.set_num_threads(max(1, int(n)))


class _Tail(threading.Thread):
    """relays a child's stderr to ours line by line and keeps the last lines for the failure report"""

    def __init__(self, rank, pipe, quiet=False):
        super().__init__(daemon=True)
        self.rank, self.pipe, self.quiet = rank, pipe, quiet
        self.lines = collections.deque(maxlen=STDERR_TAIL_LINES)

    def run(self):
        try:
            for line in self.pipe:
                self.lines.append(line.rstrip("\n"))
                if not self.quiet:
                    sys.stderr.write(f"[rank {self.rank}] {line}")
        except ValueError:                                         # pipe closed under us
            pass


class _Pump(threading.Thread):
    def __init__(self, rank, pipe, on_line):
        super().__init__(daemon=True)
        self.rank, self.pipe, self.on_line = rank, pipe, on_line

    def run(self):
        try:
            for line in self.pipe:
                self.on_line(self.rank, line.rstrip("\n"))
        except ValueError:
            pass


class RankFailure(SystemExit):
    """a rank exited non-zero: `.rank`, `.returncode`, `.stderr_tail`; str() is the message the parent prints"""

    def __init__(self, who, rank, returncode, stderr_tail, others, relayed=False):
        self.rank, self.returncode, self.stderr_tail = rank, returncode, list(stderr_tail)
        tail = "\n".join("    " + ln for ln in self.stderr_tail[-STDERR_TAIL_LINES:])
        msg = f"{who}: rank {rank} exited with code {returncode}; the other rank(s) {others} were terminated."
        if relayed:                                            # its stderr went to ours line by line, tagged [rank r]
            msg += f"\n  (stderr of rank {rank}: the lines tagged [rank {rank}] above)"
        else:
            msg += f"\n  last stderr lines of rank {rank}:\n{tail if tail else '    (none)'}"
        super().__init__(msg)


def _stop(procs, grace_s):
    live = [p for p in procs if p is not None and p.poll() is None]
    for p in live:
        try:
            p.send_signal(signal.SIGTERM)
        except OSError:
            pass
    t_end = time.monotonic() + grace_s
    for p in live:
        try:
            p.wait(timeout=max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def supervise(spawn, world: int, on_stdout_line, who: str, poll_s: float = 0.1, grace_s: float = 5.0,
              restarts: int = 0, on_event=None, quiet_stderr: bool = False) -> list:
    """Run `world` ranks to completion.  `spawn(rank) -> subprocess.Popen` (stdout=PIPE, stderr=PIPE, text=True) starts
    a FRESH child for a rank; `on_stdout_line(rank, line)` receives every stdout line.  Returns the exit codes (all 0).

    A rank that exits non-zero is reported through `on_event("failed", rank, returncode, stderr_tail)` the moment the
    poll sees it.  With `restarts` > 0 it is started again (a fresh process, up to `restarts` times per rank; the event
    is "restarted"); otherwise - or once its restarts are used up - every other rank is terminated and `RankFailure`
    (a SystemExit carrying the rank's stderr tail) is raised, within about `poll_s + grace_s`."""
    procs, tails, pumps, used = [None] * world, [None] * world, [None] * world, [0] * world

    def start(r):
        p = spawn(r)
        procs[r] = p
        if p.stdout is not None:
            pumps[r] = _Pump(r, p.stdout, on_stdout_line)
            pumps[r].start()
        if p.stderr is not None:
            tails[r] = _Tail(r, p.stderr, quiet_stderr)
            tails[r].start()

    try:
        for r in range(world):
            start(r)
        done = [False] * world
        while not all(done):
            for r in range(world):
                if done[r]:
                    continue
                rc = procs[r].poll()
                if rc is None:
                    continue
                for th in (pumps[r], tails[r]):                    # the pipes are at EOF: drain what is buffered
                    if th is not None:
                        th.join(timeout=5)
                if rc == 0:
                    done[r] = True
                    continue
                tail = list(tails[r].lines) if tails[r] is not None else []
                if on_event:
                    on_event("failed", r, rc, tail)
                if used[r] < restarts:
                    used[r] += 1
                    start(r)
                    if on_event:
                        on_event("restarted", r, rc, tail)
                    continue
                others = [q for q in range(world) if q != r and not done[q]]
                _stop([procs[q] for q in others], grace_s)
                raise RankFailure(who, r, rc, tail, others, relayed=not quiet_stderr)
            if not all(done):
                time.sleep(poll_s)
        return [p.returncode for p in procs]
    finally:
        _stop(procs, grace_s)                                      # nothing of ours outlives the parent, whatever happened
