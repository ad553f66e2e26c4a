"""Build libriser_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m riser_amd.build [--force]

The shared library lands in riser_amd/lib/ (git-ignored, but shipped to the GPU box with
the working tree).  hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(LIBDIR, "libriser_amd.so")
ARCH = "gfx950"

SOURCES = ["api.hip", "normalise.hip", "normalise_float.hip", "pointwise.hip", "fc_head.hip", "conv_f32.hip", "conv_wino.hip", "conv_wino_thin.hip", "conv_wino4.hip", "conv_stream_f32.hip", "conv_small_f32.hip", "conv_ring_h16.hip", "conv_ring_f8.hip", "conv_thin_h16.hip", "conv_wres_h16.hip", "conv_stream_h16.hip", "polya.hip", "seqnet.hip"]
# -ffp-contract=off: the normalise kernel must reproduce numpy's separately rounded fp64
# operations; the conv kernels use explicit fmaf / MFMA so they lose nothing.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def csrc_sha16() -> str:
    """sha256[:16] over the sources of the ConvNet path (csrc/*.hpp, api.hip, normalise.hip, pointwise.hip, conv_*.hip, sorted
    by name; not the generic conv programs, the fc head, the float normaliser or the poly(A) detector): the stamp a
    rocprofv3 PMC summary carries (tools/pmc_summary.py) so that bench.py can tell whether tracked counters still describe
    the kernels of this tree"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hpp") or f in ("api.hip", "normalise.hip", "pointwise.hip") or (f.startswith("conv_") and f.endswith(".hip")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib_name: str = None) -> str:
    """extra_flags / lib_name build a diagnostic variant next to the shipped library."""
    global OBJDIR, LIB
    if lib_name:
        OBJDIR = os.path.join(HERE, "csrc", "build_" + lib_name)
        LIB = os.path.join(LIBDIR, f"lib{lib_name}.so")
        force = True
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "riser_amd.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden",
             "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), *objs, "-o", LIB])
    if not lib_name:
        build_hostpack(force, run)
    return LIB


HOSTPACK = os.path.join(HERE, "_hostpack.so")


def build_hostpack(force: bool = False, run=None) -> str:
    """riser_amd/_hostpack.so: the per-read host loops of the control loop (csrc/hostpack.c, CPython C API, gcc; no GPU
    code).  Optional at run time: control.py falls back to the Python loops when it is missing."""
    import sysconfig
    src = os.path.join(CSRC, "hostpack.c")
    if force or _stale(HOSTPACK, [src]):
        cc = shutil.which("gcc") or shutil.which("cc")
        if not cc:
            raise RuntimeError("gcc not found (riser_amd/_hostpack.so)")
        cmd = [cc, "-O2", "-shared", "-fPIC", "-Wall", "-pthread", "-I" + sysconfig.get_paths()["include"], src, "-o", HOSTPACK]
        if run:
            run(cmd)
        else:
            subprocess.run(cmd, check=True)
    return HOSTPACK


if __name__ == "__main__":
    if "--item-stamps" in sys.argv:          # diagnostic variant, loaded with RISER_AMD_LIB=...
        print(build(extra_flags=["-DRS_ITEM_STAMPS"], lib_name="riser_amd_stamps", verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
