"""Acrobot's float64 sin/cos comes from a generated table (carl_amd/csrc/sincos_table.inc, tools/gen_sincos_table.py)
plus a short correction (classic_control.hip.h: SinCosTab).  CPU checks: every table entry is the correctly rounded
double of sin / cos (i pi / 256), the reduction constants are the split of pi / 256, and a NumPy emulation of the
kernel's formula (without fma: a slightly pessimistic bound) stays below 1e-13 over +-40 rad (round 4 shortened the
formula -- one-term reduction, sin r without its r^5 / 120 term, cos r without its r^4 / 24 term: 6e-11; round 6 took the
cosine's term back, because stiff contexts inside the reference's declared bounds amplify a stage's trig error by up to
1e8: tools/fuzz_wide_contexts.py)."""
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
    cr = 1 + z * (-0.5 + z * (1 / 24))
    sn, cs = S * cr + C * sr, C * cr - S * sr
    assert np.abs(sn - np.sin(np.longdouble(x))).max() < 1e-13
    assert np.abs(cs - np.cos(np.longdouble(x))).max() < 1e-13


def test_mountaincar_double_angle_cosine_error_bound():
    """fast_math.hip.h: cos_twice_fast (MountainCar / MountainCarContinuous: cos(3 position) as cos(2 h), h = 1.5
    position).  The constants are read out of the header; the fp32 Horner steps are emulated with exact products
    rounded once (what v_fma_f32 does).  Bound over the whole fitted range |2 h| <= 3.7: 2e-7."""
    src = open(os.path.join(ROOT, "carl_amd", "csrc", "fast_math.hip.h")).read()
    body = src.split("float cos_twice_fast(float h)")[1].split("return r;")[0]
    c = [np.float32(float.fromhex(v)) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+(?=f)", body)]
    assert len(c) == 5, c
    guard = float(re.search(r"fabsf\(h\) <= ([0-9.]+)f", body).group(1))
    f32, f64 = np.float32, np.float64

    def fma(a, b, d):
        return (a.astype(f64) * b.astype(f64) + f64(d)).astype(f32)  # 24 x 24-bit product is exact in double

    x = np.linspace(-2 * guard, 2 * guard, 2_000_001).astype(f32)
    h = (x * f32(0.5)).astype(f32)
    z = (h * h).astype(f32)
    p = fma(z, np.full_like(z, c[0]), c[1])
    for ck in c[2:]:
        p = fma(z, p, ck)
    cs = fma(z, p, 1.0)
    r = fma((cs + cs).astype(f32), cs, -1.0)
    assert np.abs(r.astype(f64) - np.cos(x.astype(f64))).max() < 2e-7
    # 1.5 p and 3 p / 2 are the same float (what lets the kernel form h straight from the position)
    pos = np.random.default_rng(0).uniform(-1.3, 0.7, 100_000).astype(f32)
    assert np.array_equal((f32(1.5) * pos).astype(f32), ((f32(3.0) * pos).astype(f32) * f32(0.5)).astype(f32))
    # default position range [-1.2, 0.6] sits inside the fitted range
    assert 1.5 * 1.2 <= guard


def test_pendulum_packed_sincos_error_bound():
    """fast_math.hip.h: sincos_fast_pk (Pendulum).  NumPy emulation of its formula -- nearest integer by the
    1.5 * 2^23 trick with the integer read from the low mantissa bits, three-term reduction, the two polynomials,
    quadrant signs by XOR -- with every fp32 fma as an exact double product rounded once.  Constants read out of the
    header.  Bound 1.5e-7 over |x| <= 1e5 (where the kernel leaves the fast path) and 1.2e-7 over +-100 rad."""
    src = open(os.path.join(ROOT, "carl_amd", "csrc", "fast_math.hip.h")).read()
    body = src.split("void sincos_fast_pk(float x, float& sn, float& cs)")[1].split("const bool big")[0]
    hexes = [float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+(?=f)", body)]
    two_over_pi, hi, mid, lo, magic = hexes[:5]
    assert magic == 1.5 * 2 ** 23
    dec = [float(v) for v in re.findall(r"(-?\d\.\d+e-\d+)f", body)]
    (s3, c3), (s2_, c2_), (s1, c1), (s0, c0) = [(dec[i], dec[i + 1]) for i in range(0, 8, 2)]
    f32, f64 = np.float32, np.float64

    def fma(a, b, d):
        return (np.asarray(a, f32).astype(f64) * np.asarray(b, f32).astype(f64) + np.asarray(d, f32).astype(f64)).astype(f32)

    rng = np.random.default_rng(5)
    for span, bound in ((1.0e5, 1.5e-7), (100.0, 1.2e-7)):
        x = rng.uniform(-span, span, 2_000_000).astype(f32)
        t = fma(x, f32(two_over_pi), f32(magic))
        k = (t - f32(magic)).astype(f32)
        q = t.view(np.uint32)
        assert np.array_equal((q & 3).astype(np.int64), k.astype(np.int64) & 3)  # the low mantissa bits are k mod 4
        assert np.abs(k.astype(f64) - x.astype(f64) * (2 / np.pi)).max() <= 0.5 + 1e-2
        r = fma(k, f32(-hi), x)
        r = fma(k, f32(-mid), r)
        r = fma(k, f32(-lo), r)
        z = (r * r).astype(f32)
        ps = fma(z, f32(s3), f32(s2_)); ps = fma(z, ps, f32(s1)); ps = fma(z, ps, f32(s0))
        pc = fma(z, f32(c3), f32(c2_)); pc = fma(z, pc, f32(c1)); pc = fma(z, pc, f32(c0))
        S = fma((r * z).astype(f32), ps, r)
        C = fma((z * z).astype(f32), pc, fma(z, f32(-0.5), f32(1.0)))
        odd = (q & 1).astype(bool)
        sn = np.where(odd, C, S).view(np.uint32) ^ ((q << np.uint32(30)) & np.uint32(0x80000000))
        cs = np.where(odd, S, C).view(np.uint32) ^ (((q + np.uint32(1)) << np.uint32(30)) & np.uint32(0x80000000))
        sn, cs = sn.view(f32), cs.view(f32)
        assert np.abs(sn.astype(f64) - np.sin(x.astype(f64))).max() < bound
        assert np.abs(cs.astype(f64) - np.cos(x.astype(f64))).max() < bound
