"""CPU oracle for the Shi-Tomasi corner extractor (TEST INFRASTRUCTURE ONLY) -- SURVEY.md 8(f) row f3.

Restatement of reference modules/features/shi_tomasi.cc (Extract :38-54, ResizeBuffers :56-66,
GetKeyPoints :75-99, IsLocalMaximum :123-160, FastSobelXYandScore :163-345, DetectCorner :347-400,
ComputeMinEigenValue :402-409) and of the caller's mask filter (modules/tracking/tracking.cc:118-134).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file.

Parity pinning: **parity unpinned** -- the reference holds no test or golden vector for the extractor;
the goldens are this restatement's own output.  Two restatements are kept and held to each other:
  * `scores_literal`: the reference's single pass over the image, statement by statement, with its
    rolling three-row pointers, its short-term column sums c1..c3 / row sums r1..r3 and its two-column
    tensor memory -- slow (Python loops), used on small images;
  * `scores_closed`: what that pass computes, written per output cell (NumPy), used at 640x480.
What the pass does, as executed (it is not a textbook Sobel / structure tensor at the borders):
  * gradients are int16 3x3 Sobel sums WITHOUT normalisation; gradient row i (i >= 4) is taken from
    image rows i-2, i-1, i (one row above where it is stored), rows 2 and 3 from rows (1,2,2) and
    (2,2,3) (:256-270 re-reads row 2); row 0 and the last row use their own two-row formulas and run
    their column loops to `rows` (not `cols`) (:187,:336) -- so images must be at least as wide as high;
    X gradient columns 0 and cols-1, Y gradient rows 0 and rows-1 are never written (stay 0).
  * score row r (0 <= r <= rows-4) holds min-eigenvalue of the 3x3 box tensor (times 1/9, float32)
    of gradient rows r..r+2, columns c-1..c+1, for c in [1, cols-2] (:293-313 writes scores.ptr(i-2)).
  * the last-row pass (:318-344) then overwrites score row rows-4, columns 1..rows-2, with the tensor
    of gradient rows rows-3..rows-1, reading X-gradient entries of the last row BEFORE this pass has
    rewritten them: all but column 1 are the previous Extract call's values (zeros on the first call).
  * buffers persist between calls (:39-41 reallocates only on a size change): score cells the pass
    never writes (columns 0 and cols-1, rows >= rows-3) keep the -1 marks of earlier calls.
  * all tensor sums are integers below 2^24, hence exact in float32 whatever the order; the eigenvalue
    is float32 arithmetic with separate multiplies and adds (the reference builds with -O3 only,
    CMakeLists.txt:19: baseline x86-64, no FMA) and std::sqrt(float); a negative radicand gives NaN,
    which passes the `< 80` test (:142) exactly as in the reference.
"""
import numpy as np

F32 = np.float32
INV9 = F32(1.0) / F32(9.0)
MIN_SCORE = F32(80.0)
N_PREV = 15          # IsLocalMaximum :125


def _eig(t0, t1, t2):
    """ComputeMinEigenValue :402-409 on float32 arrays / scalars."""
    with np.errstate(invalid="ignore"):
        t0, t1, t2 = F32(t0), F32(t1), F32(t2)
        tr = t0 + t2
        det = t0 * t2 - t1 * t1
        root = tr * tr - F32(4) * det
        return ((tr - np.sqrt(root)) * F32(0.5)).astype(F32) if isinstance(root, np.ndarray) else F32((tr - np.sqrt(root)) * F32(0.5))


def _round_half_away(v):
    v = np.asarray(v, np.float64)
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int64)


class ShiTomasi:
    def __init__(self, nms_window=5):
        self.w = int(nms_window)
        self.shape = None
        self.next_id = 0

    def _resize(self, shape):
        self.shape = shape
        self.Xg = np.zeros(shape, np.int16)
        self.Yg = np.zeros(shape, np.int16)
        self.scores = np.zeros(shape, np.float32)

    # ------------------------------------------------------------------ literal single pass
    def scores_literal(self, im):
        I = im.astype(np.int64)
        rows, cols = I.shape
        Xg, Yg, sc = self.Xg, self.Yg, self.scores
        st = {}                                              # the two-column tensor memory of DetectCorner

        def detect(col, gx, gy, score_row):
            def colsum(c):
                a = sum(int(Xg[r, c]) * int(Xg[r, c]) for r in gx)
                b = sum(int(Xg[r, c]) * int(Yg[q, c]) for r, q in zip(gx, gy))
                d = sum(int(Yg[q, c]) * int(Yg[q, c]) for q in gy)
                return a, b, d
            if col == 1:
                st["c1"], st["c2"] = colsum(col), colsum(col + 1)
                c0 = colsum(col - 1)
                t = [st["c1"][k] + st["c2"][k] + c0[k] for k in range(3)]
            else:
                part = [st["c1"][k] + st["c2"][k] for k in range(3)]
                st["c1"] = st["c2"]
                st["c2"] = colsum(col + 1)
                t = [part[k] + st["c2"][k] for k in range(3)]
            tf = [F32(F32(v) * INV9) for v in t]
            sc[score_row, col] = _eig(tf[0], tf[1], tf[2])

        # first row (:170-192): two-row sums, column loop bounded by `rows`
        p1, p2 = 0, 1
        c1 = 3 * I[p1, 0] + I[p2, 0]; c2 = 3 * I[p1, 1] + I[p2, 1]; c3 = 3 * I[p1, 2] + I[p2, 2]
        Xg[0, 1] = c3 - c1
        for j in range(2, rows - 1):
            c1, c2 = c2, c3
            c3 = 2 * I[p1, j + 1] + 2 * I[p2, j + 1]
            Xg[0, j] = c3 - c1
        # second row (:197-253)
        p0, p1, p2 = 0, 1, 2
        r1 = np.zeros(cols, np.int64); r2 = np.zeros(cols, np.int64); r3 = np.zeros(cols, np.int64)
        r1[0] = 2 * I[p0, 0] + 2 * I[p0, 1]; r2[0] = 2 * I[p1, 0] + 2 * I[p1, 1]; r3[0] = 2 * I[p2, 0] + 2 * I[p2, 1]
        Yg[1, 0] = r3[0] - r1[0]
        c1 = I[p0, 0] + 2 * I[p1, 0] + I[p2, 0]; c2 = I[p0, 1] + 2 * I[p1, 1] + I[p2, 1]; c3 = I[p0, 2] + 2 * I[p1, 2] + I[p2, 2]
        Xg[1, 1] = c3 - c1
        r1[1] = I[p0, 0] + 2 * I[p0, 1] + I[p2, 2]              # (:221 reads pIm[2][2])
        r2[1] = I[p1, 0] + 2 * I[p1, 1] + I[p1, 2]
        r3[1] = I[p2, 0] + 2 * I[p2, 1] + I[p2, 2]
        Yg[1, 1] = r3[1] - r1[1]
        for j in range(2, cols - 1):
            c1, c2 = c2, c3
            c3 = I[p0, j + 1] + 2 * I[p1, j + 1] + I[p2, j + 1]
            Xg[1, j] = c3 - c1
            r1[j] = I[p0, j - 1] + 2 * I[p0, j] + I[p0, j + 1]
            r2[j] = I[p1, j - 1] + 2 * I[p1, j] + I[p1, j + 1]
            r3[j] = I[p2, j - 1] + 2 * I[p2, j] + I[p2, j + 1]
            Yg[1, j] = r3[j] - r1[j]
        for rr, pp in ((r1, p0), (r2, p1), (r3, p2)):
            rr[cols - 1] = 2 * I[pp, cols - 1] + 2 * I[pp, cols - 2]
        Yg[1, cols - 1] = r3[cols - 1] - r1[cols - 1]
        # inner rows (:256-316)
        for i in range(2, rows - 1):
            p0, p1, p2 = p1, p2, i
            r1 = r2.copy(); r2 = r3.copy()
            r3[0] = 2 * I[p2, 0] + 2 * I[p2, 1]
            Yg[i, 0] = r3[0] - r1[0]
            c1 = I[p0, 0] + 2 * I[p1, 0] + I[p2, 0]; c2 = I[p0, 1] + 2 * I[p1, 1] + I[p2, 1]; c3 = I[p0, 2] + 2 * I[p1, 2] + I[p2, 2]
            Xg[i, 1] = c3 - c1
            r3[1] = I[p2, 0] + 2 * I[p2, 1] + I[p2, 2]
            Yg[i, 1] = r3[1] - r1[1]
            g = (i - 2, i - 1, i)
            for j in range(2, cols - 1):
                c1, c2 = c2, c3
                c3 = I[p0, j + 1] + 2 * I[p1, j + 1] + I[p2, j + 1]
                Xg[i, j] = c3 - c1
                r3[j] = I[p2, j - 1] + 2 * I[p2, j] + I[p2, j + 1]
                Yg[i, j] = r3[j] - r1[j]
                detect(j - 1, g, g, i - 2)
            r3[cols - 1] = 2 * I[p2, cols - 1] + 2 * I[p2, cols - 2]
            Yg[i, cols - 1] = r3[cols - 1] - r1[cols - 1]
            detect(cols - 2, g, g, i - 2)
        # last row (:321-344): p0, p1, p2 are still those of the last inner row; scores still go to row rows-4
        g = (rows - 3, rows - 2, rows - 1)
        c1 = 3 * I[p1, 0] + I[p2, 0]; c2 = 3 * I[p1, 1] + I[p2, 1]; c3 = 3 * I[p1, 2] + I[p2, 2]
        Xg[rows - 1, 1] = c3 - c1
        for j in range(1, rows - 1):
            c1, c2 = c2, c3
            c3 = I[p0, j + 1] + 2 * I[p1, j + 1] + I[p2, j + 1]
            Xg[rows - 1, j] = c3 - c1
            detect(j, g, g, rows - 4)

    # ------------------------------------------------------------------ per-cell closed form
    def scores_closed(self, im):
        I = im.astype(np.int64)
        rows, cols = I.shape
        Xg = np.zeros((rows, cols), np.int64)
        Yg = np.zeros((rows, cols), np.int64)

        def trip(i):                                           # image rows behind gradient row i >= 1
            return (0, 1, 2) if i == 1 else (1, 2, 2) if i == 2 else (2, 2, 3) if i == 3 else (i - 2, i - 1, i)

        def rs(r):                                             # row sums: 1-2-1 inside, 2-2 at the ends
            v = np.empty(cols, np.int64)
            v[1:-1] = I[r, :-2] + 2 * I[r, 1:-1] + I[r, 2:]
            v[0] = 2 * I[r, 0] + 2 * I[r, 1]
            v[-1] = 2 * I[r, -1] + 2 * I[r, -2]
            return v
        # row 0: columns 1 .. rows-2
        C0 = np.where(np.arange(cols) <= 2, 3 * I[0] + I[1], 2 * I[0] + 2 * I[1])
        j = np.arange(1, rows - 1)
        Xg[0, j] = C0[j + 1] - C0[j - 1]
        for i in range(1, rows - 1):
            a, b, c = trip(i)
            Cs = I[a] + 2 * I[b] + I[c]
            Xg[i, 1:-1] = Cs[2:] - Cs[:-2]
        top = rs(0)
        top[1] = I[0, 0] + 2 * I[0, 1] + I[2, 2]
        Yg[1] = rs(2) - top
        if rows > 3:
            Yg[2] = rs(2) - rs(1)
        if rows > 4:
            Yg[3] = rs(3) - rs(2)
        for i in range(4, rows - 1):
            Yg[i] = rs(i) - rs(i - 2)
        sc = self.scores

        def tensor_rows(r0, X3, Y3, c_lo, c_hi):
            xx = (X3 * X3).sum(0); xy = (X3 * Y3).sum(0); yy = (Y3 * Y3).sum(0)
            box = lambda v: v[c_lo - 1:c_hi] + v[c_lo:c_hi + 1] + v[c_lo + 1:c_hi + 2]
            t = [(box(v).astype(F32) * INV9).astype(F32) for v in (xx, xy, yy)]
            sc[r0, c_lo:c_hi + 1] = _eig(t[0], t[1], t[2])
        for r in range(0, rows - 3):
            tensor_rows(r, Xg[r:r + 3], Yg[r:r + 3], 1, cols - 2)
        # last-row pass: X gradient of the last row as it is READ there = the previous call's, except column 1
        a, b, c = trip(rows - 2)
        Cs = I[a] + 2 * I[b] + I[c]
        Cp = 3 * I[b] + I[c]
        seq = np.concatenate([Cp[:3], Cs[2:]])                 # c-values in the order the pass forms them
        new_last = np.zeros(cols, np.int64)
        new_last[j] = seq[j + 2] - seq[j]
        read_last = self.Xg[rows - 1].astype(np.int64).copy()
        read_last[1] = new_last[1]
        X3 = np.stack([Xg[rows - 3], Xg[rows - 2], read_last])
        Y3 = np.stack([Yg[rows - 3], Yg[rows - 2], np.zeros(cols, np.int64)])
        tensor_rows(rows - 4, X3, Y3, 1, rows - 2)
        Xg[rows - 1] = new_last
        self.Xg[:] = Xg.astype(np.int16)
        self.Yg[:] = Yg.astype(np.int16)

    # ------------------------------------------------------------------ IsLocalMaximum / GetKeyPoints
    def local_maxima(self):
        """Boolean map of IsLocalMaximum (:123-160) over the whole score buffer."""
        sc = self.scores
        rows, cols = sc.shape
        with np.errstate(invalid="ignore"):
            cand = (sc != F32(-1.0)) & ~(sc < MIN_SCORE)
        marked = (sc == F32(-1.0))

        def window_any(m, n):
            out = m.copy()
            for d in range(1, n + 1):
                out[d:, :] |= m[:-d, :]; out[:-d, :] |= m[d:, :]
            m2 = out.copy()
            for d in range(1, n + 1):
                out[:, d:] |= m2[:, :-d]; out[:, :-d] |= m2[:, d:]
            return out

        def window_fmax(v, n):
            out = v.copy()
            for d in range(1, n + 1):
                out[d:, :] = np.fmax(out[d:, :], v[:-d, :]); out[:-d, :] = np.fmax(out[:-d, :], v[d:, :])
            v2 = out.copy()
            for d in range(1, n + 1):
                out[:, d:] = np.fmax(out[:, d:], v2[:, :-d]); out[:, :-d] = np.fmax(out[:, :-d], v2[:, d:])
            return out
        near_mark = window_any(marked, N_PREV)
        with np.errstate(invalid="ignore"):
            beaten = window_fmax(sc, self.w) > sc               # a NaN never beats and is never beaten
        return cand & ~near_mark & ~beaten

    def local_maxima_literal(self):
        sc = self.scores
        rows, cols = sc.shape
        out = np.zeros(sc.shape, bool)
        for r in range(rows):
            for c in range(cols):
                cur = sc[r, c]
                if cur == F32(-1.0) or cur < MIN_SCORE:
                    continue
                ok = True
                for i in range(max(0, r - N_PREV), min(rows - 1, r + N_PREV) + 1):
                    for j in range(max(0, c - N_PREV), min(cols - 1, c + N_PREV) + 1):
                        if sc[i, j] == F32(-1.0):
                            ok = False
                        elif abs(i - r) <= self.w and abs(j - c) <= self.w and sc[i, j] > cur:
                            ok = False
                        if not ok:
                            break
                    if not ok:
                        break
                out[r, c] = ok
        return out

    def extract(self, im, prev_xy=None, mask=None, literal=False):
        """Extract (:38-54) followed by the caller's mask filter (tracking.cc:118-134).  prev_xy: the
        keypoints already held (their cells are marked -1 and they are not returned again).  Returns
        (xy float32 n x 2, class ids) of the NEW keypoints in row-major order."""
        im = np.ascontiguousarray(im, np.uint8)
        if im.shape[1] < im.shape[0] or im.shape[0] < 5:
            raise ValueError("the reference's row passes index columns up to rows-1: width >= height >= 5 required")
        if self.shape != im.shape:
            self._resize(im.shape)
        (self.scores_literal if literal else self.scores_closed)(im)
        if prev_xy is not None and len(prev_xy):
            p = np.asarray(prev_xy, np.float32).reshape(-1, 2)
            self.scores[_round_half_away(p[:, 1]), _round_half_away(p[:, 0])] = F32(-1.0)
        lm = self.local_maxima_literal() if literal else self.local_maxima()
        rr, cc = np.nonzero(lm)                                # row-major, like the r / c loops of :77-88
        ids = self.next_id + np.arange(len(rr))
        self.next_id += len(rr)
        xy = np.stack([cc, rr], 1).astype(np.float32)
        if mask is not None:
            keep = np.asarray(mask)[rr, cc] != 0
            xy, ids = xy[keep], ids[keep]
        return xy, ids.astype(np.int32)
