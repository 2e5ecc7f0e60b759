"""Host restatement of the 64-bit fixed-point scatter planes (csrc/lds_plane.h: fix_scale / lds_add_fix_biased): the
arithmetic the resample2d d/d input1 kernel relies on, checked with numpy -- no GPU.

  * the scale is the power of two that puts the largest |gradient| at 2^40 (exponent clamped for tiny maxima);
  * float -> integer is `(double)v + 1.5 * 2^52`, whose bit pattern minus 0x4338000000000000 is rint(v) in two's complement;
  * every float contribution down to 2^-16 of the maximum is represented exactly, so the integer sum equals the exact sum
    of those floats, in any order."""
import numpy as np

MAGIC = np.float64(6755399441055744.0)          # 1.5 * 2^52
MAGIC_BITS = np.uint64(0x4338000000000000)


def fix_scale(amax):
    bits = np.float32(amax).view(np.uint32)
    e = int(bits >> 23)
    se = min(253, 127 + 40 - (e - 127))
    up = np.uint32(se << 23).view(np.float32)
    down = np.uint64((1023 - (se - 127)) << 52).view(np.float64)
    return up, down


def to_fixed(v_scaled):
    d = np.float64(v_scaled) + MAGIC
    return (d.view(np.uint64) - MAGIC_BITS).view(np.int64)


def test_magic_number_conversion_is_round_to_nearest_for_both_signs():
    assert MAGIC.view(np.uint64) == MAGIC_BITS
    xs = np.array([0.0, 1.0, -1.0, 2.5, -2.5, 3.5, -3.5, 2.0 ** 40, -(2.0 ** 40), 2.0 ** 50 - 1, 123456789.4, -987654321.6])
    got = np.array([to_fixed(x) for x in xs])
    assert np.array_equal(got, np.rint(xs).astype(np.int64))


def test_scale_puts_the_maximum_at_two_to_the_forty():
    for amax in (1.0, 3.7e-5, 6.1e4, 2.0 ** -60, 1.9999999):
        up, down = fix_scale(amax)
        assert 2.0 ** 40 <= np.float64(amax) * np.float64(up) < 2.0 ** 41
        assert np.float64(up) * down == 1.0
    up, down = fix_scale(2.0 ** -120)             # tiny maxima: the scale's exponent is clamped, not overflowed
    assert np.isfinite(up) and np.float64(up) * down == 1.0


def test_sum_is_exact_and_order_independent():
    rng = np.random.default_rng(0)
    g = (rng.standard_normal(4096) * 3).astype(np.float32)
    w = rng.random(4096).astype(np.float32)        # weights <= 1, as the tap weights are
    v = (g * w).astype(np.float32)                 # the float contributions the kernel forms
    up, down = fix_scale(np.abs(g).max())
    big = np.abs(v) >= np.abs(g).max() * 2.0 ** -16
    fixed = np.array([to_fixed(np.float32(x) * up) for x in v])
    # contributions within 2^-16 of the maximum are exact; the rest round at 2^-40 of it
    assert np.array_equal(fixed[big].astype(np.float64) * down, v[big].astype(np.float64))
    assert np.max(np.abs(fixed.astype(np.float64) * down - v.astype(np.float64))) <= np.abs(g).max() * 2.0 ** -40
    total = np.int64(fixed.sum())
    for _ in range(5):
        assert np.int64(fixed[rng.permutation(fixed.size)].sum()) == total
    assert abs(np.float64(total) * down - v.astype(np.float64).sum()) <= 4096 * np.abs(g).max() * 2.0 ** -41
    # headroom: the scaled maximum is below 2^41, so 2^22 - 1 (4 M) contributions of that magnitude still fit an int64
    assert (2 ** 41) * (2 ** 22 - 1) < 2 ** 63
