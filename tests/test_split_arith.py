"""The arithmetic of the split kernels (csrc/conv_split.h, dense_split.h, conv_split_train.h), restated in NumPy on the CPU: an f32
operand as two f16 pieces a = ah + al (ah = f16(a), al = f16(a - ah)), a product block as ah bh + ah bl + al bh accumulated in f32.
What the GPU tests then hold the kernels to - f32-grade error - is a property of this scheme, shown here without a GPU:
with operands scaled into the f16 normal range the error against float64 is that of a plain f32 product sum; without the scaling the
low pieces go subnormal and the error grows by orders of magnitude (why the kernels scale activations by 2^6 and weights to [2^12, 2^13))."""
import numpy as np


def split(a):
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def split_matmul(a, b, sa=1.0, sb=1.0):
    ah, al = split((a * sa).astype(np.float32))
    bh, bl = split((b * sb).astype(np.float32))
    # every f16 x f16 product is exact in f32; the sums run in f32 (np.float32 matmul), the dropped al * bl is 2^-22 of a product
    c = (al @ bh).astype(np.float32) + (ah @ bl).astype(np.float32) + (ah @ bh).astype(np.float32)
    return c / np.float32(sa * sb)


def pow2_scale(w, top=13):
    e = np.frexp(np.abs(w).max())[1]
    return float(np.ldexp(1.0, top - e))


def test_split_products_are_f32_grade_once_scaled():
    rng = np.random.default_rng(0)
    a = np.maximum(rng.standard_normal((512, 180)), 0).astype(np.float32)          # activations after a ReLU
    w = (0.08 * rng.standard_normal((180, 40))).astype(np.float32)                   # conv-sized weights
    ref = a.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(ref).max()
    e32 = np.abs((a @ w).astype(np.float64) - ref).max() / scale
    es = np.abs(split_matmul(a, w, 64.0, pow2_scale(w)).astype(np.float64) - ref).max() / scale
    assert e32 < 1e-6 and es < 2 * e32 + 2e-7, (e32, es)
    # pieces are exact: a == ah + al to 2^-22 of |a| wherever the low piece is a normal f16 number
    ah, al = split(a * 64.0)
    big = np.abs(a * 64.0) >= 1.0
    assert np.abs((ah + al) - a * 64.0)[big].max() <= 2.0 ** -21 * np.abs(a * 64.0)[big].max()


def test_unscaled_small_operands_lose_the_low_piece():
    rng = np.random.default_rng(1)
    a = (2.0 ** -8 * rng.standard_normal((256, 180))).astype(np.float32)
    w = (2.0 ** -8 * rng.standard_normal((180, 40))).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(ref).max()
    e_plain = np.abs(split_matmul(a, w).astype(np.float64) - ref).max() / scale
    e_scaled = np.abs(split_matmul(a, w, pow2_scale(a, 10), pow2_scale(w)).astype(np.float64) - ref).max() / scale
    assert e_plain > 5 * e_scaled and e_scaled < 1e-6, (e_plain, e_scaled)       # measured on the GPU too: tools/f16split_probe.hip (7e-6 against 3e-8 .. 5e-7)


def test_overflow_bound_of_the_scaled_planes():
    # activations are stored times 2^6 in f16: finite up to 65504 / 64 = 1023.5; the kernels flag anything above 60000 / 64 = 937.5
    with np.errstate(over='ignore'):
        assert np.isfinite(np.float16(1023.0 * 64.0)) and not np.isfinite(np.float16(1024.0 * 64.0))
