"""CPU oracle for the pyramidal Lucas-Kanade tracker (TEST INFRASTRUCTURE ONLY).

NumPy restatement of reference modules/matching/lucas_kanade_tracker.cc (a21 SetReferenceImage
:47-168, a22 Track :170-596) and of what cv::buildOpticalFlowPyramid hands to it (a23).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file.

Parity pinning: **parity unpinned**.  The reference holds no test or golden vector for this path
and OpenCV's source is not under /root/reference (find_package(OpenCV 4), modules/CMakeLists.txt:3,
unpinned; README.md:47 "tested with 3.2.0 and 4.4.0").  The pyramid below restates OpenCV's
documented behaviour: 5-tap [1 4 6 4 1]/16 pyrDown with (sum+128)>>8 and
reflect-101 borders, un-normalised Scharr 3-10-3 derivative into int16x2 with reflect-101 at the
image edge, image border reflect-101 / derivative border zero, winSize of padding.  LK parity is
defined "given identical pyramids"; the goldens are this restatement's own output.

Arithmetic conventions restated from the reference source:
  * fixed-point bilinear sampling, W_BITS = 14, cvRound = round-half-to-even, CV_DESCALE (LK:31,102-107)
  * all window sums are *sequential* float32 accumulations in row-major order (LK:147-148,343-344,
    396-401) with separate multiply and add (no FMA contraction)
  * `int diff = J*alpha - I - beta` truncates toward zero (LK:392)
  * cv::norm / Point2f::ddot are evaluated in double (LK:444,452)
  * cv::Mat::dot accumulates double products of the float entries (LK:577-579); `short / 32` and
    convertTo(CV_8U) round half to even and saturate (LK:546-551)
"""
import numpy as np

F32 = np.float32
W_BITS = 14
FLT_SCALE = F32(1.0 / (1 << 20))

# LandmarkStatus (modules/utilities/landmark_status.h:23-30)
TRACKED_WITH_3D, TRACKED, JUST_TRIANGULATED, BAD, OUT_IMAGE_BOUNDARIES, BAD_FEATURE = range(6)


def is_usable(s):
    return s in (TRACKED_WITH_3D, TRACKED, JUST_TRIANGULATED)


# ----------------------------------------------------------------------------
# a23: pyramid = what cv::buildOpticalFlowPyramid(img, pyr, winSize, maxLevel) returns
# ----------------------------------------------------------------------------
def _reflect101(i, n):
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def pyr_down(img):
    h, w = img.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int32)
    src = img.astype(np.int32)
    cols = _reflect101(2 * np.arange(ow)[:, None] + np.arange(-2, 3)[None, :], w)        # ow x 5
    tmp = (src[:, cols] * k[None, None, :]).sum(axis=2)                                     # h x ow
    rows = _reflect101(2 * np.arange(oh)[:, None] + np.arange(-2, 3)[None, :], h)        # oh x 5
    out = (tmp[rows, :] * k[None, :, None]).sum(axis=1)                                     # oh x ow
    return ((out + 128) >> 8).astype(np.uint8)


def scharr_deriv(img):
    """calcSharrDeriv: int16 (dx, dy), reflect-101 at the image edge."""
    h, w = img.shape
    s = img.astype(np.int32)
    r0 = s[_reflect101(np.arange(h) - 1, h), :]
    r2 = s[_reflect101(np.arange(h) + 1, h), :]
    t0 = (r0 + r2) * 3 + s * 10
    t1 = r2 - r0
    xm = _reflect101(np.arange(w) - 1, w)
    xp = _reflect101(np.arange(w) + 1, w)
    dx = t0[:, xp] - t0[:, xm]
    dy = (t1[:, xp] + t1[:, xm]) * 3 + t1 * 10
    return np.stack([dx, dy], axis=2).astype(np.int16)


class Level:
    """One pyramid level with winSize of padding: img(y, x) / deriv(y, x) valid for
    -pad <= x < w + pad."""

    def __init__(self, img, pad):
        self.h, self.w = img.shape
        self.pad = pad
        ys = _reflect101(np.arange(-pad, self.h + pad), self.h)
        xs = _reflect101(np.arange(-pad, self.w + pad), self.w)
        self.img = img
        self.I = img[np.ix_(ys, xs)].astype(np.int32)                 # reflect-101 border
        d = scharr_deriv(img).astype(np.int32)
        self.D = np.zeros((self.h + 2 * pad, self.w + 2 * pad, 2), np.int32)   # constant (zero) border
        self.D[pad:pad + self.h, pad:pad + self.w] = d


def build_pyramid(img, max_level=4, win=21):
    img = np.ascontiguousarray(img, np.uint8)
    levels = [Level(img, win)]
    cur = img
    for _ in range(max_level):
        nh, nw = (cur.shape[0] + 1) // 2, (cur.shape[1] + 1) // 2
        if nw <= win or nh <= win:
            break
        cur = pyr_down(cur)
        levels.append(Level(cur, win))
    return levels


# ----------------------------------------------------------------------------
# fixed-point window sampling (LK:100-144, 288-339)
# ----------------------------------------------------------------------------
def _weights(a, b):
    a, b = F32(a), F32(b)
    one = F32(1.0)
    s = F32(1 << W_BITS)
    iw00 = int(np.rint((one - a) * (one - b) * s))
    iw01 = int(np.rint(a * (one - b) * s))
    iw10 = int(np.rint((one - a) * b * s))
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10
    return iw00, iw01, iw10, iw11


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _sample(L, ix, iy, wts, win, with_deriv=True):
    iw00, iw01, iw10, iw11 = wts
    p = L.pad
    y0, x0 = iy + p, ix + p
    A = L.I[y0:y0 + win + 1, x0:x0 + win + 1]
    val = _descale(A[:-1, :-1] * iw00 + A[:-1, 1:] * iw01 + A[1:, :-1] * iw10 + A[1:, 1:] * iw11, W_BITS - 5)
    if not with_deriv:
        return val.astype(np.int32), None
    B = L.D[y0:y0 + win + 1, x0:x0 + win + 1, :]
    d = _descale(B[:-1, :-1] * iw00 + B[:-1, 1:] * iw01 + B[1:, :-1] * iw10 + B[1:, 1:] * iw11, W_BITS)
    return val.astype(np.int32), d.astype(np.int32)


def _seq_sum(v):
    """sequential float32 accumulation in row-major order (what the reference's loops do)."""
    v = np.asarray(v, F32).ravel()
    return F32(np.add.accumulate(v, dtype=F32)[-1]) if len(v) else F32(0)


class LucasKanadeOracle:
    """Mirror of LucasKanadeTracker with the constants System sets (SLAM/system.cc:77-84)."""

    def __init__(self, win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4):
        self.win, self.max_level, self.max_iters = win, max_level, max_iters
        self.eps, self.min_eig = F32(epsilon), F32(min_eig)
        self.clear()

    def clear(self):
        self.prev = np.zeros((0, 2), F32)
        self.meanI, self.meanI2, self.Iref, self.Idref = [], [], [], []

    # ---- a21
    def set_reference(self, img, pts, mask=None):
        pyr = build_pyramid(img, self.max_level, self.win)
        self.n_levels = len(pyr)
        pts = np.asarray(pts, F32).reshape(-1, 2)
        n = len(pts)
        self.prev = pts.copy()
        nl = self.max_level + 1
        self.meanI = [np.full(n, -1, F32) for _ in range(nl)]
        self.meanI2 = [np.full(n, -1, F32) for _ in range(nl)]
        self.Iref = [[None] * n for _ in range(nl)]
        self.Idref = [[None] * n for _ in range(nl)]
        half = F32((self.win - 1) * 0.5)
        gap = self.win // 2                                           # round(winSize.width/2), LK:58
        area = F32(self.win * self.win)
        for level in range(len(pyr) - 1, -1, -1):
            L = pyr[level]
            sf = 1 << level
            for i in range(n):
                px = F32(pts[i, 0] / F32(sf)) - half
                py = F32(pts[i, 1] / F32(sf)) - half
                ix, iy = int(np.floor(px)), int(np.floor(py))
                if ix < -gap or ix >= L.w - gap or iy < -gap or iy >= L.h - gap:
                    continue
                if mask is not None:
                    # LK:125-131 reads mask.at(my, mx) without a bounds check; out-of-image reads are
                    # treated as "not masked" here (documented choice)
                    mx = (ix + np.arange(self.win)) * sf
                    my = (iy + np.arange(self.win)) * sf
                    okx = (mx >= 0) & (mx < mask.shape[1])
                    oky = (my >= 0) & (my < mask.shape[0])
                    sub = mask[np.ix_(np.clip(my, 0, mask.shape[0] - 1), np.clip(mx, 0, mask.shape[1] - 1))]
                    if np.any((sub == 0) & oky[:, None] & okx[None, :]):
                        continue
                val, d = _sample(L, ix, iy, _weights(px - F32(ix), py - F32(iy)), self.win)
                self.meanI[level][i] = (_seq_sum(val.astype(F32)) * FLT_SCALE) / area
                self.meanI2[level][i] = (_seq_sum((val * val).astype(F32)) * FLT_SCALE) / area
                self.Iref[level][i] = val.astype(np.int16)
                self.Idref[level][i] = d.astype(np.int16)

    # ---- a22
    def track(self, img, pts, status, initial_flow=True, min_ssim=0.7):
        pyr = build_pyramid(img, self.max_level, self.win)
        pts = np.array(pts, F32).reshape(-1, 2)
        status = np.array(status, np.int32)
        n = len(self.prev)
        win = self.win
        half = F32((win - 1) * 0.5)
        gap = win // 2 + 1                                             # LK:186
        area = F32(win * win)
        top = self.max_level
        for level in range(top, -1, -1):
            if level >= len(pyr):
                continue
            L = pyr[level]
            inv = F32(1.0 / (1 << level))
            for i in range(n):
                if not is_usable(status[i]):
                    continue
                prev = self.prev[i] * inv
                if level == top:
                    nxt = pts[i] * inv if initial_flow else prev.copy()
                else:
                    nxt = pts[i] * F32(2.0)
                pts[i] = nxt
                pvx, pvy = prev[0] - half, prev[1] - half
                ipx, ipy = int(np.floor(pvx)), int(np.floor(pvy))
                if ipx < -gap or ipx >= L.w - gap or ipy < -gap or ipy >= L.h - gap:
                    if level == 0:
                        status[i] = OUT_IMAGE_BOUNDARIES
                    continue
                if self.Iref[level][i] is None:
                    if level == 0:
                        status[i] = OUT_IMAGE_BOUNDARIES
                    continue
                meanI, meanI2 = self.meanI[level][i], self.meanI2[level][i]
                Iw = self.Iref[level][i].astype(F32)
                dI = self.Idref[level][i].astype(F32)
                start = nxt.copy()
                nx, ny = F32(nxt[0] - half), F32(nxt[1] - half)
                pdx = pdy = F32(0)
                for j in range(self.max_iters):
                    ix, iy = int(np.floor(nx)), int(np.floor(ny))
                    if ix < -gap or ix >= L.w - gap or iy < -gap or iy >= L.h - gap:
                        if level == 0:
                            status[i] = OUT_IMAGE_BOUNDARIES
                        break
                    val, d = _sample(L, ix, iy, _weights(nx - F32(ix), ny - F32(iy)), win)
                    meanJ = (_seq_sum(val.astype(F32)) * FLT_SCALE) / area
                    meanJ2 = (_seq_sum((val * val).astype(F32)) * FLT_SCALE) / area
                    with np.errstate(divide="ignore", invalid="ignore"):
                        alpha = F32(np.sqrt(F32(meanI2 / meanJ2)))
                    beta = F32(meanI - F32(alpha * meanJ))
                    Jf = val.astype(F32)
                    with np.errstate(invalid="ignore", over="ignore"):
                        diff = np.trunc(F32(F32(Jf * alpha) - Iw) - beta)
                        diff = np.where(np.isfinite(diff), diff, 0).astype(np.int64).astype(F32)   # int diff, then int*float
                        dx = (dI[:, :, 0] + (d[:, :, 0].astype(F32) * alpha).astype(F32)).astype(F32)
                        dy = (dI[:, :, 1] + (d[:, :, 1].astype(F32) * alpha).astype(F32)).astype(F32)
                        b1 = _seq_sum((diff * dx).astype(F32)) * FLT_SCALE
                        b2 = _seq_sum((diff * dy).astype(F32)) * FLT_SCALE
                        A11 = _seq_sum((dx * dx).astype(F32)) * FLT_SCALE
                        A22 = _seq_sum((dy * dy).astype(F32)) * FLT_SCALE
                        A12 = _seq_sum((dx * dy).astype(F32)) * FLT_SCALE
                        D = F32(F32(A11 * A22) - F32(A12 * A12))
                        disc = F32(F32(F32(A11 - A22) * F32(A11 - A22)) + F32(F32(F32(4.0) * A12) * A12))
                        min_eig = F32(F32(F32(A22 + A11) - F32(np.sqrt(disc))) / F32(2 * win * win))
                    if bool(min_eig < self.min_eig) or bool(D < np.finfo(F32).eps):       # NaN compares false (LK:422)
                        # `continue`: the state is unchanged, every remaining iteration repeats this outcome
                        if level == 0:
                            status[i] = BAD_FEATURE
                        break
                    D = F32(F32(1.0) / D)
                    dlx = F32(F32(F32(A12 * b2) - F32(A22 * b1)) * D)
                    dly = F32(F32(F32(A12 * b1) - F32(A11 * b2)) * D)
                    nx, ny = F32(nx + dlx), F32(ny + dly)
                    pts[i] = (F32(nx + half), F32(ny + half))
                    if (pts[i, 0] < gap + 1 or pts[i, 0] >= L.w - 1 - gap or
                            pts[i, 1] < gap + 1 or pts[i, 1] >= L.h - 1 - gap):
                        if level == 0:
                            status[i] = OUT_IMAGE_BOUNDARIES
                        break
                    ddx, ddy = float(pts[i, 0] - start[0]), float(pts[i, 1] - start[1])
                    if np.sqrt(ddx * ddx + ddy * ddy) > 10:
                        pts[i] = start
                        if level == 0:
                            status[i] = BAD
                        break
                    if float(dlx) * float(dlx) + float(dly) * float(dly) <= float(self.eps):
                        break
                    if j > 0 and abs(float(F32(dlx + pdx))) < 0.01 and abs(float(F32(dly + pdy))) < 0.01:
                        pts[i] = (F32(pts[i, 0] - F32(dlx * F32(0.5))), F32(pts[i, 1] - F32(dly * F32(0.5))))
                        break
                    pdx, pdy = dlx, dly
        # ---- SSIM gate at level 0 (LK:465-592)
        L = pyr[0]
        C1 = F32((0.01 * 255) * (0.01 * 255))
        C2 = F32((0.03 * 255) * (0.03 * 255))
        N_inv = F32(1.0) / F32(win * win)
        N_inv_1 = F32(1.0) / F32(win * win - 1)
        good = 0
        ssims = np.full(n, np.nan, F32)
        for i in range(n):
            if not is_usable(status[i]):
                continue
            if np.isnan(pts[i, 0]) or np.isnan(pts[i, 1]):
                status[i] = OUT_IMAGE_BOUNDARIES
                continue
            nx, ny = F32(pts[i, 0] - half), F32(pts[i, 1] - half)
            ix, iy = int(np.floor(nx)), int(np.floor(ny))
            if ix < -gap or ix >= L.w - gap * 2 or iy < -gap or iy >= L.h - gap * 2:
                status[i] = OUT_IMAGE_BOUNDARIES
                continue
            val, _ = _sample(L, ix, iy, _weights(nx - F32(ix), ny - F32(iy)), win, with_deriv=False)
            cur = np.clip(np.rint(val / 32.0), 0, 255).astype(F32)                 # short/32 then CV_8U
            ref = np.rint(self.Iref[0][i].astype(np.float64) / 32.0).astype(F32)  # short/32
            mu_x = F32(ref.sum(dtype=np.float64)) * N_inv                          # exact integer sums
            mu_y = F32(cur.sum(dtype=np.float64)) * N_inv
            xn = (ref - mu_x).astype(F32).astype(np.float64)
            yn = (cur - mu_y).astype(F32).astype(np.float64)
            sx = F32(np.sqrt(F32(np.sum(xn * xn) * float(N_inv_1))))
            sy = F32(np.sqrt(F32(np.sum(yn * yn) * float(N_inv_1))))
            sxy = F32(np.sum(xn * yn) * float(N_inv_1))
            two = F32(2.0)
            ssim = F32(F32(F32(F32(two * mu_x) * mu_y + C1) * F32(F32(two * sxy) + C2)) /
                       F32(F32(F32(mu_x * mu_x) + F32(mu_y * mu_y) + C1) * F32(F32(sx * sx) + F32(sy * sy) + C2)))
            ssims[i] = ssim
            if ssim < F32(min_ssim):
                status[i] = BAD_FEATURE
            else:
                good += 1
        return pts, status, good, ssims
