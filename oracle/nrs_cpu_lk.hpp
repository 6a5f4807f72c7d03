// CPU restatement of the pyramidal Lucas-Kanade tracker in C++ -- TEST INFRASTRUCTURE ONLY (part of oracle/nrs_cpu.cpp).
// The compiled CPU side of the tracked-fps half of the metric next to oracle/nrs_cpu_track.hpp, and a second checker next to
// oracle/lk_oracle.py, whose conventions it shares (see that file's header): reference
//   LucasKanadeTracker::SetReferenceImage   modules/matching/lucas_kanade_tracker.cc:47-168
//   LucasKanadeTracker::Track               modules/matching/lucas_kanade_tracker.cc:170-596
//   cv::buildOpticalFlowPyramid             call sites LK:50,184; OpenCV's source is not in the tree: the pyramid restates the
//                                           documented pyrDown / Scharr / border behaviour: PARITY UNPINNED
// Arithmetic: fixed-point bilinear sampling (W_BITS 14, cvRound = round half to even), sequential float32 window sums in row-major
// order with separate multiply and add (this file is compiled with contraction off for these functions), `int diff` truncation,
// double precision norms.  Held bit for bit to oracle/lk_oracle.py on the committed golden (tests/test_oracle_cpp_track_cpu.py).
#pragma once

#define LK_NOFMA __attribute__((optimize("fp-contract=off")))

inline int lk_reflect101(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

// numpy's pairwise summation of a contiguous float64 array (numpy/core/src/umath/loops_utils.h.src: blocks of 128, eight
// partial sums): the SSIM gate of the NumPy oracle sums its 441 products with np.sum, and this restatement is held to it bit for bit
inline double lk_np_sum(const double* a, int n) {
    if (n < 8) { double r = 0.; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return lk_np_sum(a, n2) + lk_np_sum(a + n2, n - n2);
}

struct LkLevel {
    int w = 0, h = 0, pad = 0, stride = 0;
    vector<uint8_t> img;                                             // h x w
    vector<uint8_t> I;                                               // (h + 2 pad) x (w + 2 pad), reflect-101 border
    vector<int16_t> D;                                               // same extent x 2 (dx, dy), zero border
    void build(const vector<uint8_t>& src, int w_, int h_, int pad_) {
        w = w_; h = h_; pad = pad_; stride = w + 2 * pad;
        img = src;
        I.resize((size_t)(h + 2 * pad) * stride);
        for (int y = -pad; y < h + pad; ++y)
            for (int x = -pad; x < w + pad; ++x) I[(size_t)(y + pad) * stride + (x + pad)] = img[(size_t)lk_reflect101(y, h) * w + lk_reflect101(x, w)];
        D.assign((size_t)(h + 2 * pad) * stride * 2, 0);
        // calcSharrDeriv: t0 = (r0 + r2) 3 + r1 10, t1 = r2 - r0 per column; dx = t0[x+1] - t0[x-1], dy = (t1[x+1] + t1[x-1]) 3 + t1[x] 10
        vector<int> t0(w), t1(w);
        for (int y = 0; y < h; ++y) {
            const uint8_t* r0 = &img[(size_t)lk_reflect101(y - 1, h) * w];
            const uint8_t* r1 = &img[(size_t)y * w];
            const uint8_t* r2 = &img[(size_t)lk_reflect101(y + 1, h) * w];
            for (int x = 0; x < w; ++x) { t0[x] = ((int)r0[x] + (int)r2[x]) * 3 + (int)r1[x] * 10; t1[x] = (int)r2[x] - (int)r0[x]; }
            for (int x = 0; x < w; ++x) {
                const int xm = lk_reflect101(x - 1, w), xp = lk_reflect101(x + 1, w);
                int16_t* d = &D[((size_t)(y + pad) * stride + (x + pad)) * 2];
                d[0] = (int16_t)(t0[xp] - t0[xm]);
                d[1] = (int16_t)((t1[xp] + t1[xm]) * 3 + t1[x] * 10);
            }
        }
    }
};

inline void lk_pyr_down(const vector<uint8_t>& src, int w, int h, vector<uint8_t>& out, int& ow, int& oh) {
    ow = (w + 1) / 2; oh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    vector<int> tmp((size_t)h * ow);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < ow; ++x) {
            int s = 0;
            for (int t = 0; t < 5; ++t) s += (int)src[(size_t)y * w + lk_reflect101(2 * x + t - 2, w)] * k[t];
            tmp[(size_t)y * ow + x] = s;
        }
    out.resize((size_t)oh * ow);
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            int s = 0;
            for (int t = 0; t < 5; ++t) s += tmp[(size_t)lk_reflect101(2 * y + t - 2, h) * ow + x] * k[t];
            out[(size_t)y * ow + x] = (uint8_t)((s + 128) >> 8);
        }
}

struct LkTracker {
    int win = 21, max_level = 4, max_iters = 10;
    float eps = 1e-4f, min_eig = 1e-4f;
    int n = 0;
    vector<float> prev;                                              // n x 2
    vector<vector<float>> meanI, meanI2;                             // per level
    vector<vector<uint8_t>> filled;
    vector<vector<int16_t>> Iref, Idref;                             // per level: n x win^2, n x win^2 x 2

    static void build_pyramid(const uint8_t* img, int w, int h, int stride, int max_level, int win, vector<LkLevel>& pyr) {
        vector<uint8_t> cur((size_t)w * h);
        for (int y = 0; y < h; ++y) std::memcpy(&cur[(size_t)y * w], img + (size_t)y * stride, w);
        pyr.clear();
        pyr.emplace_back();
        pyr.back().build(cur, w, h, win);
        int cw = w, ch = h;
        for (int l = 0; l < max_level; ++l) {
            const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
            if (nw <= win || nh <= win) break;
            vector<uint8_t> nxt;
            int ow, oh;
            lk_pyr_down(cur, cw, ch, nxt, ow, oh);
            cur.swap(nxt);
            cw = ow; ch = oh;
            pyr.emplace_back();
            pyr.back().build(cur, cw, ch, win);
        }
    }
    LK_NOFMA static void weights(float a, float b, int* iw) {
        const float one = 1.0f, s = (float)(1 << 14);
        iw[0] = (int)std::nearbyintf((one - a) * (one - b) * s);
        iw[1] = (int)std::nearbyintf(a * (one - b) * s);
        iw[2] = (int)std::nearbyintf((one - a) * b * s);
        iw[3] = (1 << 14) - iw[0] - iw[1] - iw[2];
    }
    static void sample(const LkLevel& L, int ix, int iy, const int* iw, int win, int* val, int* d) {
        const int p = L.pad;
        for (int y = 0; y < win; ++y) {
            const uint8_t* a0 = &L.I[(size_t)(iy + p + y) * L.stride + (ix + p)];
            const uint8_t* a1 = a0 + L.stride;
            for (int x = 0; x < win; ++x)
                val[y * win + x] = ((int)a0[x] * iw[0] + (int)a0[x + 1] * iw[1] + (int)a1[x] * iw[2] + (int)a1[x + 1] * iw[3] + (1 << 8)) >> 9;
            if (d) {
                const int16_t* b0 = &L.D[((size_t)(iy + p + y) * L.stride + (ix + p)) * 2];
                const int16_t* b1 = b0 + 2 * L.stride;
                for (int x = 0; x < win; ++x)
                    for (int c = 0; c < 2; ++c)
                        d[(y * win + x) * 2 + c] = ((int)b0[2 * x + c] * iw[0] + (int)b0[2 * x + 2 + c] * iw[1] + (int)b1[2 * x + c] * iw[2] + (int)b1[2 * x + 2 + c] * iw[3] + (1 << 13)) >> 14;
            }
        }
    }
    LK_NOFMA static void means(const int* val, int npx, float area, float& m1, float& m2) {
        const float flt_scale = 1.0f / (float)(1 << 20);
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < npx; ++k) { s1 = s1 + (float)val[k]; s2 = s2 + (float)(val[k] * val[k]); }
        m1 = (s1 * flt_scale) / area;
        m2 = (s2 * flt_scale) / area;
    }

    LK_NOFMA void set_reference(const uint8_t* img, int w, int h, int stride, int n_, const float* pts) {
        vector<LkLevel> pyr;
        build_pyramid(img, w, h, stride, max_level, win, pyr);
        n = n_;
        prev.assign(pts, pts + 2 * (size_t)n);
        const int nl = max_level + 1, npx = win * win;
        meanI.assign(nl, vector<float>(n, -1.f)); meanI2.assign(nl, vector<float>(n, -1.f));
        filled.assign(nl, vector<uint8_t>(n, 0));
        Iref.assign(nl, vector<int16_t>((size_t)n * npx, 0)); Idref.assign(nl, vector<int16_t>((size_t)n * npx * 2, 0));
        const float half = (float)((win - 1) * 0.5), area = (float)(win * win);
        const int gap = win / 2;
        vector<int> val(npx), d(2 * npx);
        for (int level = (int)pyr.size() - 1; level >= 0; --level) {
            const LkLevel& L = pyr[level];
            const float sf = (float)(1 << level);
            for (int i = 0; i < n; ++i) {
                const float px = pts[2 * i] / sf - half, py = pts[2 * i + 1] / sf - half;
                const int ix = (int)std::floor(px), iy = (int)std::floor(py);
                if (ix < -gap || ix >= L.w - gap || iy < -gap || iy >= L.h - gap) continue;
                int iw[4];
                weights(px - (float)ix, py - (float)iy, iw);
                sample(L, ix, iy, iw, win, val.data(), d.data());
                means(val.data(), npx, area, meanI[level][i], meanI2[level][i]);
                for (int k = 0; k < npx; ++k) Iref[level][(size_t)i * npx + k] = (int16_t)val[k];
                for (int k = 0; k < 2 * npx; ++k) Idref[level][(size_t)i * npx * 2 + k] = (int16_t)d[k];
                filled[level][i] = 1;
            }
        }
    }

    static bool usable(int s) { return s == 0 || s == 1 || s == 2; }

    LK_NOFMA int track(const uint8_t* img, int w, int h, int stride, float* pts, int32_t* status, int initial_flow, float min_ssim, float* ssim_out) {
        vector<LkLevel> pyr;
        build_pyramid(img, w, h, stride, max_level, win, pyr);
        const int npx = win * win, gap = win / 2 + 1, top = max_level;
        const float half = (float)((win - 1) * 0.5), area = (float)(win * win), flt_scale = 1.0f / (float)(1 << 20);
        vector<int> val(npx), d(2 * npx);
        for (int level = top; level >= 0; --level) {
            if (level >= (int)pyr.size()) continue;
            const LkLevel& L = pyr[level];
            const float inv = (float)(1.0 / (double)(1 << level));
            for (int i = 0; i < n; ++i) {
                if (!usable(status[i])) continue;
                const float pvx0 = prev[2 * i] * inv, pvy0 = prev[2 * i + 1] * inv;
                float nx0, ny0;
                if (level == top) { nx0 = initial_flow ? pts[2 * i] * inv : pvx0; ny0 = initial_flow ? pts[2 * i + 1] * inv : pvy0; }
                else { nx0 = pts[2 * i] * 2.0f; ny0 = pts[2 * i + 1] * 2.0f; }
                pts[2 * i] = nx0; pts[2 * i + 1] = ny0;
                const float pvx = pvx0 - half, pvy = pvy0 - half;
                const int ipx = (int)std::floor(pvx), ipy = (int)std::floor(pvy);
                if (ipx < -gap || ipx >= L.w - gap || ipy < -gap || ipy >= L.h - gap) { if (level == 0) status[i] = 4; continue; }
                if (!filled[level][i]) { if (level == 0) status[i] = 4; continue; }
                const float mI = meanI[level][i], mI2 = meanI2[level][i];
                const int16_t* Iw = &Iref[level][(size_t)i * npx];
                const int16_t* dI = &Idref[level][(size_t)i * npx * 2];
                const float sx0 = nx0, sy0 = ny0;
                float nx = nx0 - half, ny = ny0 - half, pdx = 0.f, pdy = 0.f;
                for (int j = 0; j < max_iters; ++j) {
                    const int ix = (int)std::floor(nx), iy = (int)std::floor(ny);
                    if (ix < -gap || ix >= L.w - gap || iy < -gap || iy >= L.h - gap) { if (level == 0) status[i] = 4; break; }
                    int iw[4];
                    weights(nx - (float)ix, ny - (float)iy, iw);
                    sample(L, ix, iy, iw, win, val.data(), d.data());
                    float mJ, mJ2;
                    means(val.data(), npx, area, mJ, mJ2);
                    const float alpha = std::sqrt(mI2 / mJ2);
                    const float beta = mI - alpha * mJ;
                    float b1 = 0.f, b2 = 0.f, A11 = 0.f, A22 = 0.f, A12 = 0.f;
                    for (int k = 0; k < npx; ++k) {
                        const float t = ((float)val[k] * alpha - (float)Iw[k]) - beta;
                        float df = std::trunc(t);
                        if (!std::isfinite(df)) df = 0.f;
                        const float diff = (float)(long long)df;
                        const float dx = (float)dI[2 * k] + (float)d[2 * k] * alpha;
                        const float dy = (float)dI[2 * k + 1] + (float)d[2 * k + 1] * alpha;
                        b1 = b1 + diff * dx; b2 = b2 + diff * dy;
                        A11 = A11 + dx * dx; A22 = A22 + dy * dy; A12 = A12 + dx * dy;
                    }
                    b1 = b1 * flt_scale; b2 = b2 * flt_scale; A11 = A11 * flt_scale; A22 = A22 * flt_scale; A12 = A12 * flt_scale;
                    float D = A11 * A22 - A12 * A12;
                    const float disc = (A11 - A22) * (A11 - A22) + (4.0f * A12) * A12;
                    const float me = ((A22 + A11) - std::sqrt(disc)) / (float)(2 * win * win);
                    if (me < min_eig || D < std::numeric_limits<float>::epsilon()) { if (level == 0) status[i] = 5; break; }
                    D = 1.0f / D;
                    const float dlx = (A12 * b2 - A22 * b1) * D, dly = (A12 * b1 - A11 * b2) * D;
                    nx = nx + dlx; ny = ny + dly;
                    pts[2 * i] = nx + half; pts[2 * i + 1] = ny + half;
                    if (pts[2 * i] < (float)(gap + 1) || pts[2 * i] >= (float)(L.w - 1 - gap) || pts[2 * i + 1] < (float)(gap + 1) || pts[2 * i + 1] >= (float)(L.h - 1 - gap)) {
                        if (level == 0) status[i] = 4;
                        break;
                    }
                    const double ddx = (double)(pts[2 * i] - sx0), ddy = (double)(pts[2 * i + 1] - sy0);
                    if (std::sqrt(ddx * ddx + ddy * ddy) > 10) { pts[2 * i] = sx0; pts[2 * i + 1] = sy0; if (level == 0) status[i] = 3; break; }
                    if ((double)dlx * (double)dlx + (double)dly * (double)dly <= (double)eps) break;
                    if (j > 0 && std::fabs((double)(dlx + pdx)) < 0.01 && std::fabs((double)(dly + pdy)) < 0.01) {
                        pts[2 * i] = pts[2 * i] - dlx * 0.5f; pts[2 * i + 1] = pts[2 * i + 1] - dly * 0.5f;
                        break;
                    }
                    pdx = dlx; pdy = dly;
                }
            }
        }
        // ---- SSIM gate at level 0 (LK:465-592)
        const LkLevel& L = pyr[0];
        const float C1 = (float)((0.01 * 255) * (0.01 * 255)), C2 = (float)((0.03 * 255) * (0.03 * 255));
        const float N_inv = 1.0f / (float)(win * win), N_inv_1 = 1.0f / (float)(win * win - 1);
        int good = 0;
        vector<float> cur(npx), ref(npx);
        vector<double> pxx(npx), pyy(npx), pxy(npx);
        for (int i = 0; i < n; ++i) {
            if (ssim_out) ssim_out[i] = std::numeric_limits<float>::quiet_NaN();
            if (!usable(status[i])) continue;
            if (std::isnan(pts[2 * i]) || std::isnan(pts[2 * i + 1])) { status[i] = 4; continue; }
            const float nx = pts[2 * i] - half, ny = pts[2 * i + 1] - half;
            const int ix = (int)std::floor(nx), iy = (int)std::floor(ny);
            if (ix < -gap || ix >= L.w - gap * 2 || iy < -gap || iy >= L.h - gap * 2) { status[i] = 4; continue; }
            int iw[4];
            weights(nx - (float)ix, ny - (float)iy, iw);
            sample(L, ix, iy, iw, win, val.data(), nullptr);
            double sr = 0, sc = 0;
            for (int k = 0; k < npx; ++k) {
                cur[k] = (float)std::min(255.0, std::max(0.0, std::nearbyint((double)val[k] / 32.0)));
                ref[k] = (float)std::nearbyint((double)Iref[0][(size_t)i * npx + k] / 32.0);
                sr += (double)ref[k]; sc += (double)cur[k];
            }
            const float mu_x = (float)sr * N_inv, mu_y = (float)sc * N_inv;
            for (int k = 0; k < npx; ++k) {
                const double xn = (double)(ref[k] - mu_x), yn = (double)(cur[k] - mu_y);
                pxx[k] = xn * xn; pyy[k] = yn * yn; pxy[k] = xn * yn;
            }
            const double sxx = lk_np_sum(pxx.data(), npx), syy = lk_np_sum(pyy.data(), npx), sxy_ = lk_np_sum(pxy.data(), npx);
            const float sx = std::sqrt((float)(sxx * (double)N_inv_1)), sy = std::sqrt((float)(syy * (double)N_inv_1));
            const float sxy = (float)(sxy_ * (double)N_inv_1);
            const float two = 2.0f;
            const float num = ((two * mu_x) * mu_y + C1) * ((two * sxy) + C2);
            const float den = ((mu_x * mu_x) + (mu_y * mu_y) + C1) * ((sx * sx) + (sy * sy) + C2);
            const float ssim = num / den;
            if (ssim_out) ssim_out[i] = ssim;
            if (ssim < min_ssim) status[i] = 5;
            else ++good;
        }
        return good;
    }
};
