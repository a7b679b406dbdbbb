#!/usr/bin/env python3
"""Static instruction statistics of the loops of one kernel in a hipcc -S listing (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S carl_brax.hip -o brax.s
    python tools/isa_loop_stats.py brax.s 'brax_kernel<1, false, 9, false, false>' [--depth 3] [--min 100]

Uses the compiler's own loop annotations ("This Loop Header: Depth=d", "in Loop: Header=BBx", "Parent Loop BBy"): for
every loop it prints the instructions of all its blocks (nested loops included), split by class -- float64 VALU,
float64 conversions, transcendentals, other VALU, SALU, LDS, VMEM, branches -- and a weighted issue estimate from the
per-class rates of profiles/r03_fp64_rate.txt.  A STATIC count: divergent regions are counted once whether or not a
wavefront enters them, inner loops once whatever their trip count.  The substep loop of the Brax step kernel is the
depth-3 loop (fragments > env steps > substeps) with the largest float64 count.
"""
from __future__ import annotations

import re
import subprocess
import sys


def demangle(name: str) -> str:
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op: str) -> str:
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "trans64" if op.endswith("f64") else "trans32"
        if op.startswith("v_cvt_") and ("f64" in op):
            return "cvt64"
        if "f64" in op:
            return "f64"
        if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64", "v_mad_i64", "v_mul_hi_i32", "v_mul_lo_i32")):
            return "int_mul"
        if op.startswith("v_pk_"):
            return "pk"
        if op.startswith("v_cmp"):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane")):
            return "mov"
        return "valu32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


VALU = ("f64", "cvt64", "trans64", "trans32", "int_mul", "pk", "cmp", "cndmask", "mov", "valu32")
# issue cost per class, cycles per wavefront-instruction at >= 2 wavefronts per SIMD (profiles/r03_fp64_rate.txt)
COST = {"f64": 4.7, "cvt64": 4.56, "trans64": 16.5, "trans32": 8.6, "int_mul": 4.55, "pk": 4.95, "cmp": 3.0,
        "cndmask": 3.0, "mov": 3.0, "valu32": 3.0}


def kernel_body(path: str, want: str) -> list[str]:
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z[^ :]+):", l)
        if m and want.replace(" ", "") in demangle(m.group(1)).replace(" ", ""):
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel {want!r} not found")
    end = start
    while end < len(lines) and not lines[end].strip().startswith("s_endpgm"):
        end += 1
    return lines[start:end + 1]


def main() -> None:
    path, want = sys.argv[1], sys.argv[2]
    min_len, depth_only = 100, None
    a = sys.argv[3:]
    while a:
        if a[0] == "--min":
            min_len = int(a[1]); a = a[2:]
        elif a[0] == "--depth":
            depth_only = int(a[1]); a = a[2:]
        else:
            raise SystemExit(f"unknown argument {a[0]}")
    body = kernel_body(path, want)
    # blocks: (name, annotation lines, instructions)
    blocks: list[dict] = []
    cur = {"name": "entry", "ann": [], "ops": []}
    blocks.append(cur)
    for l in body[1:]:
        m = re.match(r"^(?:\.L(BB[0-9_]+):|; %bb\.([0-9]+):)\s*(;.*)?$", l)
        if m:
            cur = {"name": m.group(1) or ("bb." + m.group(2)), "ann": [m.group(3) or ""], "ops": []}
            blocks.append(cur)
            continue
        s = l.strip()
        if s.startswith(";") and not cur["ops"]:
            cur["ann"].append(s)
            continue
        s = l.split(";")[0].strip()
        if not s or s.startswith(".") or re.match(r"^[A-Za-z_0-9$.]+:$", s):
            continue
        cur["ops"].append(s.split()[0])
    parents: dict[str, list[str]] = {}  # loop header -> chain of enclosing loops (outermost first), itself last
    depth: dict[str, int] = {}
    for b in blocks:
        ann = " ".join(b["ann"])
        m = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", ann)
        if m:
            chain = re.findall(r"Parent Loop (BB[0-9_]+)", ann)
            parents[b["name"]] = chain + [b["name"]]
            depth[b["name"]] = int(m.group(1))
    total = sum(len(b["ops"]) for b in blocks)
    print(f"kernel: {demangle(body[0].split(':')[0])[:100]}  instructions: {total}")
    counts: dict[str, dict[str, int]] = {h: {} for h in parents}
    for b in blocks:
        ann = " ".join(b["ann"])
        if b["name"] in parents:
            chain = parents[b["name"]]
        else:
            m = re.search(r"in Loop: Header=(BB[0-9_]+)", ann)
            if not m or m.group(1) not in parents:
                continue
            chain = parents[m.group(1)]
        for op in b["ops"]:
            c = classify(op)
            for h in chain:
                counts[h][c] = counts[h].get(c, 0) + 1
    for h in sorted(parents, key=lambda h: (depth[h], -sum(counts[h].values()))):
        cnt = counts[h]
        n = sum(cnt.values())
        if n < min_len or (depth_only is not None and depth[h] != depth_only):
            continue
        valu = sum(cnt.get(c, 0) for c in VALU)
        cyc = sum(cnt.get(c, 0) * COST[c] for c in VALU)
        parts = " ".join(f"{c}={cnt[c]}" for c in sorted(cnt, key=lambda c: -cnt[c]))
        print(f"depth {depth[h]} {h:10s} len {n:5d}  VALU {valu:5d} (~{cyc:6.0f} cyc)  {parts}")


if __name__ == "__main__":
    main()
