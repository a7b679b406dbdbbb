"""Compile the HIP engine for gfx950 with hipcc (in-tree, no JIT cache).

``python -m carl_amd.build`` -> carl_amd/lib/libcarl_amd.so.  hipcc cross-compiles
without a GPU, so this also runs in the build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcarl_amd.so")
ARCH = "gfx950"

# translation unit -> extra compile flags.  No SLP vectorizer in either unit: carl_brax.hip -- see its header comment;
# carl_amd.hip (r04, tools/ab_probe.sh, interleaved builds on one box): with it the Acrobot + MountainCar pair launch
# is 3.5 % slower at 65 536 lanes per family and 2.5 % at 8 192, CartPole at 8 192 lanes 2 % slower, and the one
# packing that paid (Pendulum's sine / cosine polynomials) is written out by hand in fast_math.hip.h: sincos_fast_pk.
SOURCES = {"carl_amd.hip": ["-fno-slp-vectorize"], "carl_brax.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _deps() -> list[str]:
    out = []
    for root in (CSRC, os.path.join(os.path.dirname(_HERE), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".h", ".hpp", ".inc")):
                out.append(os.path.join(root, f))
    return out


HASH_PATH = os.path.join(LIB_DIR, "build.sha256")


def _source_hash(extra_flags: list[str] | None = None) -> str:
    """Content hash of every source / header and of the flags.  File times do not survive a copy
    of the tree (the GPU boxes receive a snapshot), so staleness is decided on content."""
    import hashlib

    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    for d in _deps():
        h.update(os.path.relpath(d, root).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(repr(sorted(SOURCES.items())).encode())
    h.update(repr(extra_flags or []).encode())
    return h.hexdigest()


def needs_build(extra_flags: list[str] | None = None) -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as f:
        return f.read().strip() != _source_hash(extra_flags)


def _refuse_ablation(extra_flags: list[str] | None) -> None:
    """The profiling-only switches of rounds 1-3 (CARL_EXP_*: kernels that skip stores / the done path) no longer
    exist in the sources (removed in round 4; commit 3a3c7b2 still builds them): a flag that names one is a mistake."""
    bad = [f for f in (extra_flags or []) if "CARL_EXP_" in f or "CARL_ABLATION" in f or "CARL_STORERS" in f]
    if bad:
        raise ValueError(f"{bad}: the CARL_EXP_* / CARL_STORERS ablation switches were removed from the sources in round 4")


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    _refuse_ablation(extra_flags)
    if not force and not needs_build(extra_flags):
        return LIB_PATH
    import fcntl

    os.makedirs(LIB_DIR, exist_ok=True)
    # one builder at a time: the ranks of a multi-GPU launch import the package simultaneously
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build(extra_flags):  # another process built it while we waited
            return LIB_PATH
        return _build_locked(verbose, extra_flags)


def _build_locked(verbose: bool, extra_flags: list[str] | None) -> str:
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
              *(extra_flags or [])]
    procs, objs = [], []
    for src, flags in SOURCES.items():  # the translation units compile side by side
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [_hipcc(), *common, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", *objs, "-o", tmp]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(tmp, LIB_PATH)  # atomic: a concurrent loader never maps a half-written file
    with open(HASH_PATH, "w") as f:
        f.write(_source_hash(extra_flags) + "\n")
    return LIB_PATH


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-") and a not in ("--force",)]
    print(build(force=True, verbose=True, extra_flags=flags))
