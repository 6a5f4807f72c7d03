"""Oracle of the Shi-Tomasi extractor (oracle/shi_oracle.py, reference modules/features/shi_tomasi.cc):
the literal single-pass restatement and the per-cell closed form agree bit for bit over successive
calls of one (stateful) extractor, and reproduce the committed golden.  Parity unpinned: the reference
holds no vector for this path (the golden is the restatement's own output)."""
import os

import numpy as np
import pytest

import shi_oracle as SH

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shi_160x120.npz")


def _image(rows, cols, seed):
    r = np.random.default_rng(seed)
    a = r.normal(size=(rows + 8, cols + 8))
    for _ in range(2):
        a = (a[:-2] + a[1:-1] + a[2:]) / 3
        a = (a[:, :-2] + a[:, 1:-1] + a[:, 2:]) / 3
    a = a[2:2 + rows, 2:2 + cols]
    a = (a - a.min()) / (a.max() - a.min())
    return (255 * a).astype(np.uint8)


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("rows,cols,nms", [(24, 40, 3), (33, 33, 2), (5, 9, 1), (6, 6, 0), (40, 64, 5)])
def test_literal_pass_equals_closed_form(rows, cols, nms):
    A, B = SH.ShiTomasi(nms), SH.ShiTomasi(nms)
    prev = None
    for call in range(4):
        im = _image(rows, cols, 100 * rows + call)
        xa, ia = A.extract(im, prev, literal=True)
        xb, ib = B.extract(im, prev, literal=False)
        assert np.array_equal(A.Xg, B.Xg) and np.array_equal(A.Yg, B.Yg)
        assert _same(A.scores, B.scores)
        assert np.array_equal(xa, xb) and np.array_equal(ia, ib)
        prev = xa[:1] if len(xa) else prev               # keep one: later calls still find new points


def test_state_carries_between_calls():
    """The last-row pass reads the previous call's last-row X gradients and the never-written score cells
    keep their -1 marks: the same image gives different buffers to a fresh and to a used extractor."""
    im0, im1 = _image(30, 48, 1), _image(30, 48, 2)
    used, fresh = SH.ShiTomasi(3), SH.ShiTomasi(3)
    used.extract(im0, np.array([[0.0, 29.0]], np.float32))      # marks a cell of the (never rewritten) last row
    used.extract(im1)
    fresh.extract(im1)
    assert used.scores[29, 0] == -1.0 and fresh.scores[29, 0] == 0.0
    assert not _same(used.scores[26, 2:28], fresh.scores[26, 2:28])
    assert _same(used.scores[:26], fresh.scores[:26])


def test_requires_landscape():
    with pytest.raises(ValueError):
        SH.ShiTomasi().extract(np.zeros((40, 30), np.uint8))


def test_golden_reproduced_by_closed_form():
    g = np.load(GOLD)
    ex = SH.ShiTomasi(int(g["nms"]))
    xy0, id0 = ex.extract(g["im0"], None, g["mask"])
    assert np.array_equal(xy0, g["out_xy0"]) and np.array_equal(id0, g["out_id0"])
    xy1, id1 = ex.extract(g["im1"], g["prev1"], g["mask"])
    assert np.array_equal(xy1, g["out_xy1"]) and np.array_equal(id1, g["out_id1"])
    assert _same(ex.scores, g["out_scores1"]) and np.array_equal(ex.Xg, g["out_xg1"]) and np.array_equal(ex.Yg, g["out_yg1"])
    assert len(xy0) > 50 and len(xy1) > 3
