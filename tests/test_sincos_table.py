"""Acrobot's float64 sin/cos comes from a generated table (carl_amd/csrc/sincos_table.inc, tools/gen_sincos_table.py)
plus a short correction (classic_control.hip.h: SinCosTab).  CPU checks: every table entry is the correctly rounded
double of sin / cos (i pi / 256), the reduction constants are the split of pi / 256, and a NumPy emulation of the
kernel's formula (without fma: a slightly pessimistic bound) stays below 6e-11 over +-40 rad (round 4: the formula was
shortened -- one-term reduction, sin r without its r^5 / 120 term, cos r without its r^4 / 24 term -- from 3e-16; the
parity bar is 1e-5 and the float32 state is rounded at 6e-8)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table():
    src = open(os.path.join(ROOT, "carl_amd", "csrc", "sincos_table.inc")).read()
    consts = {m.group(1): float.fromhex(m.group(2)) for m in re.finditer(r"#define (CARL_SINCOS_TAB_\w+) (-?0x[0-9a-f.]+p[+-]\d+)", src)}
    vals = [float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", src.split("CARL_SINCOS_TAB_VALUES")[1])]
    return consts, np.array(vals).reshape(-1, 2)


def test_table_entries_are_correctly_rounded():
    import mpmath as mp

    mp.mp.dps = 60
    consts, tab = _table()
    assert tab.shape == (512, 2)
    step = mp.pi / 256
    for i in range(512):
        s, c = float(mp.sin(step * i)), float(mp.cos(step * i))
        s, c = (0.0 if abs(s) < 1e-30 else s), (0.0 if abs(c) < 1e-30 else c)
        assert tab[i, 0] == s and tab[i, 1] == c, i
    hi = consts["CARL_SINCOS_TAB_STEP_HI"]
    assert hi == float(step) and consts["CARL_SINCOS_TAB_STEP_LO"] == float(step - mp.mpf(hi))
    assert consts["CARL_SINCOS_TAB_INV_STEP"] == float(1 / step)


def test_table_formula_error_bound():
    consts, tab = _table()
    inv, hi, lo = (consts[k] for k in ("CARL_SINCOS_TAB_INV_STEP", "CARL_SINCOS_TAB_STEP_HI", "CARL_SINCOS_TAB_STEP_LO"))
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 1_000_000), rng.uniform(-np.pi, np.pi, 1_000_000)])
    magic = float.fromhex("0x1.8p52")
    t = (np.longdouble(x) * np.longdouble(inv) + np.longdouble(magic)).astype(np.float64)  # the kernel's fma(x, inv, magic)
    k = t - magic
    assert np.abs(k - np.rint(x * inv)).max() <= 1.0  # (differs from rint() only at exact ties of the unrounded product)
    i = (t.view(np.uint64) & np.uint64(511)).astype(np.int64)  # the low mantissa bits ARE the integer
    assert (i == (k.astype(np.int64) & 511)).all()
    r = (np.longdouble(x) - np.longdouble(k) * np.longdouble(hi)).astype(np.float64)  # fma(k, -hi, x): exact product
    # (the 80-bit emulation of the fma leaves 11 fractional bits beside 1.5 * 2^52: k can sit 2^-11 of a step past the tie)
    assert np.abs(r).max() <= np.pi / 512 * (1 + 1e-3)
    assert np.abs(k).max() * abs(lo) < 2e-15  # what dropping the low part of the step costs
    S, C = tab[i, 0], tab[i, 1]
    z = r * r
    sr = r + (r * z) * (-1 / 6)
    cr = 1 + z * -0.5
    sn, cs = S * cr + C * sr, C * cr - S * sr
    assert np.abs(sn - np.sin(np.longdouble(x))).max() < 6e-11
    assert np.abs(cs - np.cos(np.longdouble(x))).max() < 6e-11
