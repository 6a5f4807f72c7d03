"""CPU checks of the LK oracle's building blocks against hand-computable cases (the reference
holds no fixture for this path; these pin the restatement's own conventions)."""
import numpy as np

import lk_oracle as LK


def test_pyr_down_constant_and_impulse():
    img = np.full((40, 50), 77, np.uint8)
    assert np.all(LK.pyr_down(img) == 77) and LK.pyr_down(img).shape == (20, 25)
    imp = np.zeros((41, 41), np.uint8)
    imp[20, 20] = 255
    d = LK.pyr_down(imp)
    assert d.shape == (21, 21)
    assert d[10, 10] == (255 * 36 + 128) >> 8 and d[10, 9] == (255 * 6 + 128) >> 8 and d[9, 9] == (255 * 1 + 128) >> 8


def test_scharr_on_ramps():
    x = np.tile(np.arange(30, dtype=np.uint8) * 3, (20, 1))
    d = LK.scharr_deriv(x)
    assert np.all(d[:, 1:-1, 0] == 16 * 2 * 3) and np.all(d[:, :, 1] == 0)      # un-normalised gain 32 per unit slope
    assert np.all(d[:, 0, 0] == 0)                                              # reflect-101 makes the edge symmetric
    y = x.T.copy()
    d = LK.scharr_deriv(y)
    assert np.all(d[1:-1, :, 1] == 96) and np.all(d[:, :, 0] == 0)


def test_padding_and_rounding_conventions():
    img = (np.arange(30 * 40).reshape(30, 40) % 251).astype(np.uint8)
    L = LK.Level(img, 21)
    assert L.I[21, 21 - 1] == img[0, 1] and L.I[21 - 2, 21] == img[2, 0]          # reflect-101, not replicate
    assert np.all(L.D[:21] == 0) and np.all(L.D[:, :21] == 0)
    assert LK._weights(0.5, 0.5) == (4096, 4096, 4096, 4096)
    assert LK._weights(0.0, 0.0) == (16384, 0, 0, 0)
    assert LK._descale(255 * 16384, 9) == 255 * 32
    # sequential float32 accumulation differs from pairwise summation on purpose
    v = np.full(441, 16777.0, np.float32) ** 2
    assert LK._seq_sum(v) == np.add.accumulate(v, dtype=np.float32)[-1]


def test_identity_tracking():
    rng = np.random.default_rng(0)
    base = rng.integers(0, 255, (12, 16)).astype(np.float64)
    img = np.kron(base, np.ones((10, 10)))
    k = np.ones(7) / 7
    img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, img)
    img = np.apply_along_axis(lambda c: np.convolve(c, k, "same"), 0, img).astype(np.uint8)
    pts = np.array([[60.3, 50.7], [100.0, 70.0], [31.0, 90.5]], np.float32)
    lk = LK.LucasKanadeOracle(max_level=2)
    lk.set_reference(img, pts)
    out, st, good, ssim = lk.track(img, pts.copy(), np.zeros(3, np.int32))
    assert good == 3 and np.all(st == 0)
    assert np.max(np.abs(out - pts)) < 0.02 and np.all(ssim > 0.999)
