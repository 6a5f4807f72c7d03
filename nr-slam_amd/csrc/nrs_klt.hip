// a21-a23: pyramidal Lucas-Kanade tracker on MI355X
// (reference modules/matching/lucas_kanade_tracker.cc: SetReferenceImage :47-168, Track :170-596,
//  PhotometricInformation round trip :598-631; cv::buildOpticalFlowPyramid call sites :50,:184).
//
// Layout / mapping:
//   * pyramid levels live padded by winSize (21) on every side: image border = reflect-101,
//     derivative border = 0, exactly the ROI-in-a-larger-buffer OpenCV hands to the tracker, so the
//     tracker's reads at negative coordinates are plain loads.  One kernel per level computes
//     pyrDown (5-tap [1 4 6 4 1], (sum+128)>>8) straight into the padded layout, one the Scharr pair.
//   * templates are point-major: [(point*levels + level)] x {441 x i16 intensity*32, 441 x (i16,i16)
//     derivative, meanI, meanI2, valid}: one contiguous 2.6 KB run per (point, level).
//   * one wave64 per point runs the whole coarse-to-fine iteration: lanes own 7 pixels of the 21x21
//     window each (bilinear fixed-point resampling, W_BITS = 14), per-pixel terms go to LDS and lanes
//     0..4 each run one *sequential* float accumulation over the 441 terms -- the reference's sums
//     are row-major float loops (LK:377-407) and float addition does not commute, so bit-identical
//     status codes need the same order.  Exact-integer sums (window mean) use wave reductions.
//   * the fp32 arithmetic is compiled with contraction off (separate mul/add, like the oracle).
#include <algorithm>
#include <cmath>
#include <vector>
#include "nrs_ctx.hpp"
#include "nrs_device.hpp"

namespace nrs {

constexpr int KW = 21, KA = KW * KW, KPAD = 21, KMAXL = 8;
constexpr int W_BITS = 14;

struct PyrLevel {
    uint8_t* img;      // (h + 2 pad) x (w + 2 pad)
    short2* der;       // same dims
    int w, h, stride;  // stride = w + 2 pad (elements)
};

struct Pyr {
    PyrLevel L[KMAXL];
    int n_levels;
};

struct KltState {
    int win = KW, max_level = 4, max_iters = 10;
    float eps = 1e-4f, min_eig = 1e-4f;
    int n = 0, cap = 0, levels = 5;          // levels = max_level + 1 slots per point
    DevBuf pyr_buf, img_in, mask_in;
    Pyr pyr;
    int pyr_w = 0, pyr_h = 0;
    DevBuf tI, tD, tMean, tValid, prev, pts, status, misc;
    // the archive: photometric information kept by a caller's key (the map point id) -- what the reference keeps in its Map per map
    // point (GetPhotometricInformation at keyframes, InsertPhotometricInformation when a point is reused: tracking.cc:383-391,457-460)
    // -- in HBM, so that templates never travel to the host and back (nrs_klt_archive_templates / nrs_klt_insert_archived)
    DevBuf aI, aD, aMean, aValid, a_idx;
    int a_cap = 0, a_levels = 0;
    std::vector<uint8_t> a_has;
};

__device__ __host__ inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

// level 0: caller image -> padded reflect-101 copy
__global__ void k_pyr_copy(const uint8_t* __restrict__ src, int sstride, PyrLevel L) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= L.stride) return;
    const int x = reflect101(xo - KPAD, L.w), y = reflect101(yo - KPAD, L.h);
    L.img[(size_t)yo * L.stride + xo] = src[(size_t)y * sstride + x];
}

// pyrDown of the previous (padded) level into the padded layout of this level
__global__ void k_pyr_down(PyrLevel P, PyrLevel L) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= L.stride) return;
    const int x = reflect101(xo - KPAD, L.w), y = reflect101(yo - KPAD, L.h);
    const int k[5] = {1, 4, 6, 4, 1};
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t* row = P.img + (size_t)(2 * y + j - 2 + KPAD) * P.stride + (2 * x - 2 + KPAD);
        acc += k[j] * (row[0] + 4 * row[1] + 6 * row[2] + 4 * row[3] + row[4]);
    }
    L.img[(size_t)yo * L.stride + xo] = (uint8_t)((acc + 128) >> 8);
}

// Scharr pair (calcSharrDeriv): reflect-101 at the image edge comes from the padded image,
// the derivative's own border is zero
__global__ void k_pyr_scharr(PyrLevel L) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= L.stride) return;
    const int x = xo - KPAD, y = yo - KPAD;
    short2 d = make_short2(0, 0);
    if (x >= 0 && x < L.w && y >= 0 && y < L.h) {
        const uint8_t* r0 = L.img + (size_t)(yo - 1) * L.stride + xo;
        const uint8_t* r1 = r0 + L.stride;
        const uint8_t* r2 = r1 + L.stride;
        int t0[3], t1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t0[c] = (r0[c - 1] + r2[c - 1]) * 3 + r1[c - 1] * 10;
            t1[c] = r2[c - 1] - r0[c - 1];
        }
        d.x = (short)(t0[2] - t0[0]);
        d.y = (short)((t1[2] + t1[0]) * 3 + t1[1] * 10);
    }
    L.der[(size_t)yo * L.stride + xo] = d;
}

__device__ inline int cv_round(float v) { return __float2int_rn(v); }          // round half to even

__device__ inline void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
#pragma clang fp contract(off)
    w00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w01 = cv_round(a * (1.f - b) * (float)(1 << W_BITS));
    w10 = cv_round((1.f - a) * b * (float)(1 << W_BITS));
    w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ inline void sample_px(const PyrLevel& L, int ix, int iy, int px, int w00, int w01, int w10, int w11,
                                 int& val, int& dx, int& dy, bool with_deriv) {
    const int y = px / KW, x = px - y * KW;
    const size_t o = (size_t)(iy + y + KPAD) * L.stride + (ix + x + KPAD);
    const uint8_t* s = L.img + o;
    val = descale(s[0] * w00 + s[1] * w01 + s[L.stride] * w10 + s[L.stride + 1] * w11, W_BITS - 5);
    if (with_deriv) {
        const short2 a = L.der[o], b = L.der[o + 1], c = L.der[o + L.stride], d = L.der[o + L.stride + 1];
        dx = descale(a.x * w00 + b.x * w01 + c.x * w10 + d.x * w11, W_BITS);
        dy = descale(a.y * w00 + b.y * w01 + c.y * w10 + d.y * w11, W_BITS);
    }
}

__device__ inline float seq_sum_441(const float* v) {          // row-major sequential accumulation
#pragma clang fp contract(off)
    float s = 0.f;
    for (int i = 0; i < KA; ++i) s += v[i];
    return s;
}

// ---------------------------------------------------------------------------------------------
// a21: SetReferenceImage, one wave per (point, level)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_klt_set_reference(Pyr pyr, int n, int levels, const float* __restrict__ pts,
                                                           const uint8_t* __restrict__ mask, int mw, int mh,
                                                           short* tI, short2* tD, float* tMean, uint8_t* tValid) {
#pragma clang fp contract(off)
    __shared__ float s_v[KA], s_v2[KA];
    const int lane = threadIdx.x;
    const int i = blockIdx.x / levels, level = blockIdx.x % levels;
    const size_t slot = (size_t)i * levels + level;
    if (lane == 0) { tValid[slot] = 0; tMean[2 * slot] = -1.f; tMean[2 * slot + 1] = -1.f; }
    if (level >= pyr.n_levels) return;
    const PyrLevel L = pyr.L[level];
    const float half = (KW - 1) * 0.5f;
    const float px = pts[2 * i] / (float)(1 << level) - half, py = pts[2 * i + 1] / (float)(1 << level) - half;
    const int ix = (int)floorf(px), iy = (int)floorf(py);
    const int gap = KW / 2;                                      // round(winSize.width/2), LK:58
    if (ix < -gap || ix >= L.w - gap || iy < -gap || iy >= L.h - gap) return;
    int w00, w01, w10, w11;
    bilinear_weights(px - (float)ix, py - (float)iy, w00, w01, w10, w11);
    bool masked = false;
    for (int p = lane; p < KA; p += 64) {
        int val, dx, dy;
        sample_px(L, ix, iy, p, w00, w01, w10, w11, val, dx, dy, true);
        tI[slot * KA + p] = (short)val;
        tD[slot * KA + p] = make_short2((short)dx, (short)dy);
        s_v[p] = (float)val;
        s_v2[p] = (float)(val * val);
        if (mask) {                                              // LK:125-131 (out-of-image reads: not masked)
            const int y = p / KW, x = p - y * KW;
            const int mx = (ix + x) << level, my = (iy + y) << level;
            if (mx >= 0 && mx < mw && my >= 0 && my < mh && mask[(size_t)my * mw + mx] == 0) masked = true;
        }
    }
    __syncthreads();
    if (__any(masked)) return;
    float r = 0.f;
    if (lane == 0) r = seq_sum_441(s_v);
    if (lane == 1) r = seq_sum_441(s_v2);
    const float FLT_SCALE = 1.f / (1 << 20);
    if (lane < 2) tMean[2 * slot + lane] = (r * FLT_SCALE) / (float)KA;
    if (lane == 0) tValid[slot] = 1;
}

// ---------------------------------------------------------------------------------------------
// a22: Track (coarse-to-fine iterations), one wave per point
// ---------------------------------------------------------------------------------------------
struct TrackArgs {
    Pyr pyr;
    int n, levels, max_level, max_iters;
    float eps, min_eig;
    int initial_flow;
    const float* prev;
    float* pts;            // in/out
    int* status;           // in/out
    const short* tI; const short2* tD; const float* tMean; const uint8_t* tValid;
};

__device__ inline bool usable(int s) { return s == NRS_TRACKED_WITH_3D || s == NRS_TRACKED || s == NRS_JUST_TRIANGULATED; }

__global__ __launch_bounds__(64) void k_klt_track(TrackArgs a) {
#pragma clang fp contract(off)
    __shared__ float s_t[5][KA];          // per-pixel terms of b1, b2, A11, A22, A12
    __shared__ float s_j2[KA];
    __shared__ short s_I[KA];
    __shared__ short2 s_D[KA];
    const int lane = threadIdx.x, i = blockIdx.x;
    int st = a.status[i];
    if (!usable(st)) return;
    const float half = (KW - 1) * 0.5f;
    const int gap = KW / 2 + 1;                                   // LK:186
    const float FLT_SCALE = 1.f / (1 << 20);
    float ptx = a.pts[2 * i], pty = a.pts[2 * i + 1];
    const float rx = a.prev[2 * i], ry = a.prev[2 * i + 1];
    for (int level = a.max_level; level >= 0; --level) {
        if (!usable(st)) break;
        if (level >= a.pyr.n_levels) continue;
        const PyrLevel L = a.pyr.L[level];
        const float inv = (float)(1. / (1 << level));
        float pvx = rx * inv, pvy = ry * inv;
        float nx, ny;
        if (level == a.max_level) {
            if (a.initial_flow) { nx = ptx * inv; ny = pty * inv; }
            else { nx = pvx; ny = pvy; }
        } else { nx = ptx * 2.f; ny = pty * 2.f; }
        ptx = nx; pty = ny;
        pvx -= half; pvy -= half;
        const int ipx = (int)floorf(pvx), ipy = (int)floorf(pvy);
        if (ipx < -gap || ipx >= L.w - gap || ipy < -gap || ipy >= L.h - gap) {
            if (level == 0) st = NRS_OUT_IMAGE_BOUNDARIES;
            continue;
        }
        const size_t slot = (size_t)i * a.levels + level;
        if (!a.tValid[slot]) {
            if (level == 0) st = NRS_OUT_IMAGE_BOUNDARIES;
            continue;
        }
        const float meanI = a.tMean[2 * slot], meanI2 = a.tMean[2 * slot + 1];
        __syncthreads();
        for (int p = lane; p < KA; p += 64) { s_I[p] = a.tI[slot * KA + p]; s_D[p] = a.tD[slot * KA + p]; }
        const float sx0 = nx, sy0 = ny;                           // startCoordinates
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_iters; ++j) {
            const int ix = (int)floorf(nx), iy = (int)floorf(ny);
            if (ix < -gap || ix >= L.w - gap || iy < -gap || iy >= L.h - gap) {
                if (level == 0) st = NRS_OUT_IMAGE_BOUNDARIES;
                break;
            }
            int w00, w01, w10, w11;
            bilinear_weights(nx - (float)ix, ny - (float)iy, w00, w01, w10, w11);
            int jv[7], jx[7], jy[7];
            int sum = 0;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int p = lane + 64 * q;
                jv[q] = jx[q] = jy[q] = 0;
                if (p < KA) {
                    sample_px(L, ix, iy, p, w00, w01, w10, w11, jv[q], jx[q], jy[q], true);
                    sum += jv[q];
                    s_j2[p] = (float)(jv[q] * jv[q]);
                }
            }
            // window mean: every partial sum is an integer < 2^24, so the float loop of the
            // reference (LK:343) is exact and equals this integer reduction
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            __syncthreads();
            float m2 = 0.f;
            if (lane == 0) m2 = seq_sum_441(s_j2);
            m2 = __shfl(m2, 0, 64);
            const float meanJ = ((float)sum * FLT_SCALE) / (float)KA;
            const float meanJ2 = (m2 * FLT_SCALE) / (float)KA;
            const float alpha = sqrtf(meanI2 / meanJ2);
            const float beta = meanI - alpha * meanJ;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int p = lane + 64 * q;
                if (p < KA) {
                    const int diff = (int)((float)jv[q] * alpha - (float)s_I[p] - beta);    // truncation, LK:392
                    const float dx = (float)s_D[p].x + (float)jx[q] * alpha;
                    const float dy = (float)s_D[p].y + (float)jy[q] * alpha;
                    const float fd = (float)diff;
                    s_t[0][p] = fd * dx; s_t[1][p] = fd * dy;
                    s_t[2][p] = dx * dx; s_t[3][p] = dy * dy; s_t[4][p] = dx * dy;
                }
            }
            __syncthreads();
            float acc = 0.f;
            if (lane < 5) acc = seq_sum_441(s_t[lane]);
            const float b1 = __shfl(acc, 0, 64) * FLT_SCALE, b2 = __shfl(acc, 1, 64) * FLT_SCALE;
            const float A11 = __shfl(acc, 2, 64) * FLT_SCALE, A22 = __shfl(acc, 3, 64) * FLT_SCALE;
            const float A12 = __shfl(acc, 4, 64) * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * KW * KW);
            if (minEig < a.min_eig || D < 1.1920929e-07f) {
                // reference `continue`s: nothing changed, every remaining iteration repeats this outcome
                if (level == 0) st = NRS_BAD_FEATURE;
                break;
            }
            D = 1.f / D;
            const float dlx = (A12 * b2 - A22 * b1) * D, dly = (A12 * b1 - A11 * b2) * D;
            nx += dlx; ny += dly;
            ptx = nx + half; pty = ny + half;
            if (ptx < gap + 1 || ptx >= L.w - 1 - gap || pty < gap + 1 || pty >= L.h - 1 - gap) {
                if (level == 0) st = NRS_OUT_IMAGE_BOUNDARIES;
                break;
            }
            const double ddx = (double)(ptx - sx0), ddy = (double)(pty - sy0);
            if (sqrt(ddx * ddx + ddy * ddy) > 10) {
                ptx = sx0; pty = sy0;
                if (level == 0) st = NRS_BAD;
                break;
            }
            if ((double)dlx * (double)dlx + (double)dly * (double)dly <= (double)a.eps) break;
            if (j > 0 && fabs((double)(dlx + pdx)) < 0.01 && fabs((double)(dly + pdy)) < 0.01) {
                ptx -= dlx * 0.5f; pty -= dly * 0.5f;
                break;
            }
            pdx = dlx; pdy = dly;
        }
    }
    if (lane == 0) { a.pts[2 * i] = ptx; a.pts[2 * i + 1] = pty; a.status[i] = st; }
}

// ---------------------------------------------------------------------------------------------
// SSIM gate at level 0 (LK:465-592), one wave per point
// ---------------------------------------------------------------------------------------------
__device__ inline int div32_rne(int v) {                         // saturate_cast<short>(v / 32.0), v >= 0
    int q = v >> 5;
    const int r = v & 31;
    if (r > 16 || (r == 16 && (q & 1))) ++q;
    return q;
}

__global__ __launch_bounds__(64) void k_klt_ssim(Pyr pyr, int n, int levels, float* pts, int* status, const short* tI,
                                                  const float* tMean, float min_ssim, float* ssim_out, int* n_good) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x, i = blockIdx.x;
    int st = status[i];
    if (!usable(st)) return;
    const PyrLevel L = pyr.L[0];
    const float half = (KW - 1) * 0.5f;
    const int gap = KW / 2 + 1;
    const float ptx = pts[2 * i], pty = pts[2 * i + 1];
    if (isnan(ptx) || isnan(pty)) { if (lane == 0) status[i] = NRS_OUT_IMAGE_BOUNDARIES; return; }
    const float nx = ptx - half, ny = pty - half;
    const int ix = (int)floorf(nx), iy = (int)floorf(ny);
    if (ix < -gap || ix >= L.w - gap * 2 || iy < -gap || iy >= L.h - gap * 2) {
        if (lane == 0) status[i] = NRS_OUT_IMAGE_BOUNDARIES;
        return;
    }
    int w00, w01, w10, w11;
    bilinear_weights(nx - (float)ix, ny - (float)iy, w00, w01, w10, w11);
    const size_t slot = (size_t)i * levels;
    int cur[7], ref[7], sc = 0, sr = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int p = lane + 64 * q;
        cur[q] = ref[q] = 0;
        if (p < KA) {
            int val, dx, dy;
            sample_px(L, ix, iy, p, w00, w01, w10, w11, val, dx, dy, false);
            cur[q] = min(255, max(0, div32_rne(val)));
            const int rv = tI[slot * KA + p];
            ref[q] = rv >= 0 ? div32_rne(rv) : -div32_rne(-rv);
            sc += cur[q]; sr += ref[q];
        }
    }
    for (int off = 32; off > 0; off >>= 1) { sc += __shfl_xor(sc, off, 64); sr += __shfl_xor(sr, off, 64); }
    const float N_inv = 1.f / (float)KA, N_inv_1 = 1.f / (float)(KA - 1);
    const float mu_x = (float)sr * N_inv, mu_y = (float)sc * N_inv;
    double xx = 0, yy = 0, xy = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int p = lane + 64 * q;
        if (p < KA) {
            const double xn = (double)((float)ref[q] - mu_x), yn = (double)((float)cur[q] - mu_y);
            xx += xn * xn; yy += yn * yn; xy += xn * yn;
        }
    }
    xx = wave_sum(xx); yy = wave_sum(yy); xy = wave_sum(xy);
    const float sx = sqrtf((float)(xx * (double)N_inv_1)), sy = sqrtf((float)(yy * (double)N_inv_1));
    const float sxy = (float)(xy * (double)N_inv_1);
    const float C1 = (float)((0.01 * 255) * (0.01 * 255)), C2 = (float)((0.03 * 255) * (0.03 * 255));
    const float ssim = ((2.f * mu_x * mu_y + C1) * (2.f * sxy + C2)) /
                       ((mu_x * mu_x + mu_y * mu_y + C1) * (sx * sx + sy * sy + C2));
    if (lane == 0) {
        if (ssim_out) ssim_out[i] = ssim;
        if (ssim < min_ssim) status[i] = NRS_BAD_FEATURE;
        else atomicAdd(n_good, 1);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static KltState* klt(nrs_ctx* c) {
    if (!c->klt) c->klt = new (std::nothrow) KltState();
    return c->klt;
}

void klt_free(nrs_ctx* c) {
    if (!c->klt) return;
    KltState* k = c->klt;
    DevBuf* bufs[] = {&k->pyr_buf, &k->img_in, &k->mask_in, &k->tI, &k->tD, &k->tMean, &k->tValid, &k->prev, &k->pts, &k->status, &k->misc, &k->aI, &k->aD, &k->aMean, &k->aValid, &k->a_idx};
    for (auto b : bufs) c->release(*b);
    delete k;
    c->klt = nullptr;
}

static int build_pyramid(nrs_ctx* c, KltState* k, const uint8_t* img, int w, int h, int stride) {
    // level sizes (buildOpticalFlowPyramid stops when the next level would not exceed winSize)
    int ws[KMAXL], hs[KMAXL], nl = 1;
    ws[0] = w; hs[0] = h;
    for (int l = 1; l <= k->max_level; ++l) {
        const int nw = (ws[l - 1] + 1) / 2, nh = (hs[l - 1] + 1) / 2;
        if (nw <= KW || nh <= KW) break;
        ws[l] = nw; hs[l] = nh;
        nl = l + 1;
    }
    size_t bytes = 0, off[KMAXL][2];
    for (int l = 0; l < nl; ++l) {
        const size_t px = (size_t)(ws[l] + 2 * KPAD) * (hs[l] + 2 * KPAD);
        off[l][0] = bytes; bytes += (px + 255) / 256 * 256;
        off[l][1] = bytes; bytes += (px * sizeof(short2) + 255) / 256 * 256;
    }
    NRS_TRY(c->ensure(k->pyr_buf, bytes));
    NRS_TRY(c->ensure(k->img_in, (size_t)stride * h));
    for (int l = 0; l < nl; ++l) {
        PyrLevel& L = k->pyr.L[l];
        L.w = ws[l]; L.h = hs[l]; L.stride = ws[l] + 2 * KPAD;
        L.img = k->pyr_buf.as<uint8_t>() + off[l][0];
        L.der = reinterpret_cast<short2*>(k->pyr_buf.as<char>() + off[l][1]);
    }
    k->pyr.n_levels = nl;
    NRS_HIP(c, hipMemcpyAsync(k->img_in.p, img, (size_t)stride * h, hipMemcpyHostToDevice, c->stream));
    for (int l = 0; l < nl; ++l) {
        const PyrLevel& L = k->pyr.L[l];
        const dim3 b(256), g((L.stride + 255) / 256, L.h + 2 * KPAD);
        if (l == 0) hipLaunchKernelGGL(k_pyr_copy, g, b, 0, c->stream, k->img_in.as<uint8_t>(), stride, L);
        else hipLaunchKernelGGL(k_pyr_down, g, b, 0, c->stream, k->pyr.L[l - 1], L);
        hipLaunchKernelGGL(k_pyr_scharr, g, b, 0, c->stream, L);
    }
    NRS_HIP(c, hipGetLastError());
    return NRS_OK;
}

static int reserve_points(nrs_ctx* c, KltState* k, int n, bool keep) {
    if (n <= k->cap) return NRS_OK;
    const int cap = std::max(n + n / 2, 256);
    const size_t L = (size_t)k->levels;
    DevBuf nI, nD, nM, nV, nP;
    {
        int rc = c->ensure(nI, sizeof(short) * KA * L * cap);
        if (rc == NRS_OK) rc = c->ensure(nD, sizeof(short2) * KA * L * cap);
        if (rc == NRS_OK) rc = c->ensure(nM, sizeof(float) * 2 * L * cap);
        if (rc == NRS_OK) rc = c->ensure(nV, L * cap);
        if (rc == NRS_OK) rc = c->ensure(nP, sizeof(float) * 2 * cap);
        if (rc != NRS_OK) {                                        // nothing of a half-made set is kept
            c->release(nI); c->release(nD); c->release(nM); c->release(nV); c->release(nP);
            return rc;
        }
    }
    if (keep && k->n > 0) {
        const size_t m = (size_t)k->n;
        NRS_HIP(c, hipMemcpyAsync(nI.p, k->tI.p, sizeof(short) * KA * L * m, hipMemcpyDeviceToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(nD.p, k->tD.p, sizeof(short2) * KA * L * m, hipMemcpyDeviceToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(nM.p, k->tMean.p, sizeof(float) * 2 * L * m, hipMemcpyDeviceToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(nV.p, k->tValid.p, L * m, hipMemcpyDeviceToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(nP.p, k->prev.p, sizeof(float) * 2 * m, hipMemcpyDeviceToDevice, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->release(k->tI); c->release(k->tD); c->release(k->tMean); c->release(k->tValid); c->release(k->prev);
    k->tI = nI; k->tD = nD; k->tMean = nM; k->tValid = nV; k->prev = nP;
    k->cap = cap;
    return NRS_OK;
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_klt_configure(nrs_ctx* c, const nrs_klt_config* cfg) {
    if (!c || !cfg) return NRS_ERR_INVALID;
    if (cfg->win_size != KW) return c->fail(NRS_ERR_INVALID, "only the reference's 21x21 window is supported (SLAM/system.cc:78)");
    if (cfg->max_level < 0 || cfg->max_level >= KMAXL || cfg->max_iters <= 0) return c->fail(NRS_ERR_INVALID, "bad KLT configuration");
    KltState* k = klt(c);
    if (!k) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    if (k->n > 0 && cfg->max_level != k->max_level) return c->fail(NRS_ERR_STATE, "cannot change max_level while templates are stored (call nrs_klt_clear)");
    if (cfg->max_level + 1 != k->levels && k->cap > 0) {
        // the template buffers are sized cap x levels: another level count invalidates them (k->n == 0 here)
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        c->release(k->tI); c->release(k->tD); c->release(k->tMean); c->release(k->tValid); c->release(k->prev);
        k->cap = 0;
    }
    k->max_level = cfg->max_level; k->levels = cfg->max_level + 1;
    k->max_iters = cfg->max_iters; k->eps = cfg->epsilon; k->min_eig = cfg->min_eig_threshold;
    return NRS_OK;
}

extern "C" int nrs_klt_clear(nrs_ctx* c) {
    if (!c) return NRS_ERR_INVALID;
    if (c->klt) c->klt->n = 0;
    return NRS_OK;
}

extern "C" int nrs_klt_num_points(nrs_ctx* c) { return (c && c->klt) ? c->klt->n : 0; }

extern "C" int nrs_klt_set_reference(nrs_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                                     const uint8_t* mask, int32_t n, const float* xy) {
    if (!c) return NRS_ERR_INVALID;
    if (!img || w <= KW || h <= KW || stride < w || n < 0 || (n > 0 && !xy)) return c->fail(NRS_ERR_INVALID, "nrs_klt_set_reference: bad argument");
    NRS_HIP(c, hipSetDevice(c->device));
    KltState* k = klt(c);
    if (!k) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    NRS_TRY(build_pyramid(c, k, img, w, h, stride));
    k->n = 0;
    NRS_TRY(reserve_points(c, k, n, false));
    k->n = n;
    if (n == 0) return NRS_OK;
    NRS_HIP(c, hipMemcpyAsync(k->prev.p, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    const uint8_t* dmask = nullptr;
    if (mask) {
        NRS_TRY(c->ensure(k->mask_in, (size_t)w * h));
        NRS_HIP(c, hipMemcpyAsync(k->mask_in.p, mask, (size_t)w * h, hipMemcpyHostToDevice, c->stream));
        dmask = k->mask_in.as<uint8_t>();
    }
    hipLaunchKernelGGL(k_klt_set_reference, dim3(n * k->levels), dim3(64), 0, c->stream, k->pyr, n, k->levels,
                       k->prev.as<float>(), dmask, w, h, k->tI.as<short>(), k->tD.as<short2>(), k->tMean.as<float>(),
                       k->tValid.as<uint8_t>());
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

extern "C" int nrs_klt_track(nrs_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t n,
                             float* xy, int32_t* status, int32_t use_initial_flow, float min_ssim, int32_t* n_good,
                             float* ssim) {
    if (!c) return NRS_ERR_INVALID;
    KltState* k = c->klt;
    if (!k) return c->fail(NRS_ERR_STATE, "nrs_klt_track before nrs_klt_set_reference");
    if (!img || w <= KW || h <= KW || stride < w || n != k->n || (n > 0 && (!xy || !status)))
        return c->fail(NRS_ERR_INVALID, "nrs_klt_track: bad argument (n must equal the number of stored templates: %d)", k->n);
    NRS_HIP(c, hipSetDevice(c->device));
    NRS_TRY(build_pyramid(c, k, img, w, h, stride));
    if (n_good) *n_good = 0;
    if (n == 0) return NRS_OK;
    NRS_TRY(c->ensure(k->pts, sizeof(float) * 2 * n));
    NRS_TRY(c->ensure(k->status, sizeof(int) * n));
    NRS_TRY(c->ensure(k->misc, sizeof(int) * 4 + sizeof(float) * n));
    NRS_HIP(c, hipMemcpyAsync(k->pts.p, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(k->status.p, status, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemsetAsync(k->misc.p, 0, sizeof(int) * 4, c->stream));
    TrackArgs a;
    a.pyr = k->pyr; a.n = n; a.levels = k->levels; a.max_level = k->max_level; a.max_iters = k->max_iters;
    a.eps = k->eps; a.min_eig = k->min_eig; a.initial_flow = use_initial_flow ? 1 : 0;
    a.prev = k->prev.as<float>(); a.pts = k->pts.as<float>(); a.status = k->status.as<int>();
    a.tI = k->tI.as<short>(); a.tD = k->tD.as<short2>(); a.tMean = k->tMean.as<float>(); a.tValid = k->tValid.as<uint8_t>();
    hipLaunchKernelGGL(k_klt_track, dim3(n), dim3(64), 0, c->stream, a);
    float* d_ssim = reinterpret_cast<float*>(k->misc.as<char>() + sizeof(int) * 4);
    hipLaunchKernelGGL(k_klt_ssim, dim3(n), dim3(64), 0, c->stream, k->pyr, n, k->levels, a.pts, a.status, a.tI, a.tMean,
                       min_ssim, ssim ? d_ssim : nullptr, k->misc.as<int>());
    NRS_HIP(c, hipGetLastError());
    int good = 0;
    NRS_HIP(c, hipMemcpyAsync(xy, k->pts.p, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(status, k->status.p, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(&good, k->misc.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (ssim) NRS_HIP(c, hipMemcpyAsync(ssim, d_ssim, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (n_good) *n_good = good;
    return NRS_OK;
}

extern "C" int nrs_klt_get_template(nrs_ctx* c, int32_t idx, float xy[2], int16_t* gray, int16_t* grad, float* mean,
                                    uint8_t* valid) {
    if (!c) return NRS_ERR_INVALID;
    KltState* k = c->klt;
    if (!k || idx < 0 || idx >= k->n) return c->fail(NRS_ERR_INVALID, "template index out of range");
    if (!xy || !gray || !grad || !mean || !valid) return c->fail(NRS_ERR_INVALID, "null output");
    const size_t L = (size_t)k->levels, s = (size_t)idx * L;
    NRS_HIP(c, hipMemcpy(xy, k->prev.as<float>() + 2 * idx, sizeof(float) * 2, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(gray, k->tI.as<short>() + s * KA, sizeof(short) * KA * L, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(grad, k->tD.as<short2>() + s * KA, sizeof(short2) * KA * L, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(mean, k->tMean.as<float>() + 2 * s, sizeof(float) * 2 * L, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(valid, k->tValid.as<uint8_t>() + s, L, hipMemcpyDeviceToHost));
    return NRS_OK;
}

extern "C" int nrs_klt_insert_template(nrs_ctx* c, const float xy[2], const int16_t* gray, const int16_t* grad,
                                       const float* mean, const uint8_t* valid) {
    if (!c) return NRS_ERR_INVALID;
    if (!xy || !gray || !grad || !mean || !valid) return c->fail(NRS_ERR_INVALID, "null input");
    KltState* k = klt(c);
    if (!k) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    NRS_TRY(reserve_points(c, k, k->n + 1, true));
    const size_t L = (size_t)k->levels, s = (size_t)k->n * L;
    NRS_HIP(c, hipMemcpy(k->prev.as<float>() + 2 * k->n, xy, sizeof(float) * 2, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tI.as<short>() + s * KA, gray, sizeof(short) * KA * L, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tD.as<short2>() + s * KA, grad, sizeof(short2) * KA * L, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tMean.as<float>() + 2 * s, mean, sizeof(float) * 2 * L, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tValid.as<uint8_t>() + s, valid, L, hipMemcpyHostToDevice));
    k->n += 1;
    return NRS_OK;
}

extern "C" int nrs_klt_get_templates(nrs_ctx* c, int32_t first, int32_t count, float* xy, int16_t* gray, int16_t* grad,
                                     float* mean, uint8_t* valid) {
    if (!c) return NRS_ERR_INVALID;
    KltState* k = c->klt;
    if (!k || first < 0 || count < 0 || first + count > k->n) return c->fail(NRS_ERR_INVALID, "template range out of bounds");
    if (count == 0) return NRS_OK;
    if (!xy || !gray || !grad || !mean || !valid) return c->fail(NRS_ERR_INVALID, "null output");
    const size_t L = (size_t)k->levels, s = (size_t)first * L, n = (size_t)count;
    NRS_HIP(c, hipMemcpy(xy, k->prev.as<float>() + 2 * first, sizeof(float) * 2 * n, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(gray, k->tI.as<short>() + s * KA, sizeof(short) * KA * L * n, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(grad, k->tD.as<short2>() + s * KA, sizeof(short2) * KA * L * n, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(mean, k->tMean.as<float>() + 2 * s, sizeof(float) * 2 * L * n, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(valid, k->tValid.as<uint8_t>() + s, L * n, hipMemcpyDeviceToHost));
    return NRS_OK;
}

extern "C" int nrs_klt_insert_templates(nrs_ctx* c, int32_t count, const float* xy, const int16_t* gray,
                                        const int16_t* grad, const float* mean, const uint8_t* valid) {
    if (!c) return NRS_ERR_INVALID;
    if (count < 0) return c->fail(NRS_ERR_INVALID, "count < 0");
    if (count == 0) return NRS_OK;
    if (!xy || !gray || !grad || !mean || !valid) return c->fail(NRS_ERR_INVALID, "null input");
    KltState* k = klt(c);
    if (!k) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    NRS_TRY(reserve_points(c, k, k->n + count, true));
    const size_t L = (size_t)k->levels, s = (size_t)k->n * L, n = (size_t)count;
    NRS_HIP(c, hipMemcpy(k->prev.as<float>() + 2 * k->n, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tI.as<short>() + s * KA, gray, sizeof(short) * KA * L * n, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tD.as<short2>() + s * KA, grad, sizeof(short2) * KA * L * n, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tMean.as<float>() + 2 * s, mean, sizeof(float) * 2 * L * n, hipMemcpyHostToDevice));
    NRS_HIP(c, hipMemcpy(k->tValid.as<uint8_t>() + s, valid, L * n, hipMemcpyHostToDevice));
    k->n += count;
    return NRS_OK;
}


// ---- the template archive (device to device) ----------------------------------------------------------------------------------
// one workgroup per template: `lv` levels of entry src_idx[i] (src_levels levels an entry) into entry dst_idx[i] (dst_levels)
__global__ __launch_bounds__(256) void k_klt_copy_templates(int n, const int* __restrict__ src_idx, const int* __restrict__ dst_idx, int lv, int src_levels,
                                                            int dst_levels, const short* __restrict__ sI, const short2* __restrict__ sD, const float* __restrict__ sM,
                                                            const uint8_t* __restrict__ sV, short* dI, short2* dD, float* dM, uint8_t* dV) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const size_t so = (size_t)src_idx[i] * src_levels, d_o = (size_t)dst_idx[i] * dst_levels;
    for (int e = threadIdx.x; e < lv * KA; e += 256) { dI[d_o * KA + e] = sI[so * KA + e]; dD[d_o * KA + e] = sD[so * KA + e]; }
    for (int e = threadIdx.x; e < 2 * lv; e += 256) dM[2 * d_o + e] = sM[2 * so + e];
    for (int e = threadIdx.x; e < lv; e += 256) dV[d_o + e] = sV[so + e];
}

// The archive is DENSE by key (a map point id indexes its slot: no lookup on the device): KA x levels x 6 bytes per key, so the key range is bounded --
// 2^20 map points = 2.9 GB at 5 levels; the reference's maps hold thousands (include/nrs.h nrs_klt_archive_templates)
constexpr int NRS_KLT_ARCHIVE_MAX_KEY = 1 << 20;
static int klt_archive_reserve(nrs_ctx* c, KltState* k, int cap_want) {
    if (k->a_levels != k->levels && k->a_cap > 0) {                // (another level count: the archive starts over)
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        c->release(k->aI); c->release(k->aD); c->release(k->aMean); c->release(k->aValid);
        k->a_cap = 0; k->a_has.clear();
    }
    k->a_levels = k->levels;
    if (cap_want <= k->a_cap) return NRS_OK;
    const int cap = std::max(cap_want + cap_want / 2, 256);
    const size_t L = (size_t)k->a_levels;
    DevBuf nI, nD, nM, nV;
    int rc = c->ensure(nI, sizeof(short) * KA * L * cap);
    if (rc == NRS_OK) rc = c->ensure(nD, sizeof(short2) * KA * L * cap);
    if (rc == NRS_OK) rc = c->ensure(nM, sizeof(float) * 2 * L * cap);
    if (rc == NRS_OK) rc = c->ensure(nV, L * cap);
    if (rc != NRS_OK) { c->release(nI); c->release(nD); c->release(nM); c->release(nV); return rc; }
    if (k->a_cap > 0) {
        const size_t m = (size_t)k->a_cap;
        hipError_t he = hipMemcpyAsync(nI.p, k->aI.p, sizeof(short) * KA * L * m, hipMemcpyDeviceToDevice, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(nD.p, k->aD.p, sizeof(short2) * KA * L * m, hipMemcpyDeviceToDevice, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(nM.p, k->aMean.p, sizeof(float) * 2 * L * m, hipMemcpyDeviceToDevice, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(nV.p, k->aValid.p, L * m, hipMemcpyDeviceToDevice, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) {                                    // (the new buffers do not outlive a failed copy; the archive stays as it was)
            c->release(nI); c->release(nD); c->release(nM); c->release(nV);
            return c->fail(NRS_ERR_HIP, "template archive: copy to the grown buffers failed: %s", hipGetErrorString(he));
        }
    }
    c->release(k->aI); c->release(k->aD); c->release(k->aMean); c->release(k->aValid);
    k->aI = nI; k->aD = nD; k->aMean = nM; k->aValid = nV;
    k->a_cap = cap;
    k->a_has.resize((size_t)cap, 0);
    return NRS_OK;
}

extern "C" int nrs_klt_archive_templates(nrs_ctx* c, int32_t n, const int32_t* slots, const int32_t* keys) {
    if (!c) return NRS_ERR_INVALID;
    if (n < 0 || (n > 0 && (!slots || !keys))) return c->fail(NRS_ERR_INVALID, "nrs_klt_archive_templates: bad argument");
    if (n == 0) return NRS_OK;
    KltState* k = c->klt;
    if (!k) return c->fail(NRS_ERR_STATE, "nrs_klt_archive_templates: no tracker state");
    int kmax = -1;
    for (int i = 0; i < n; ++i) {
        if (slots[i] < 0 || slots[i] >= k->n) return c->fail(NRS_ERR_INVALID, "nrs_klt_archive_templates: slot %d out of range", slots[i]);
        if (keys[i] < 0 || keys[i] >= NRS_KLT_ARCHIVE_MAX_KEY) return c->fail(NRS_ERR_INVALID, "nrs_klt_archive_templates: key %d out of range (the archive is dense by key: keys below %d)", keys[i], NRS_KLT_ARCHIVE_MAX_KEY);
        kmax = std::max(kmax, keys[i]);
    }
    NRS_HIP(c, hipSetDevice(c->device));
    NRS_TRY(klt_archive_reserve(c, k, kmax + 1));
    // a key listed more than once: its LAST occurrence stands (the copies run side by side: two of them must not share a destination)
    std::vector<int> u_slot, u_key;
    {
        std::vector<int> last((size_t)kmax + 1, -1);
        for (int i = 0; i < n; ++i) last[keys[i]] = i;
        for (int i = 0; i < n; ++i) if (last[keys[i]] == i) { u_slot.push_back(slots[i]); u_key.push_back(keys[i]); }
    }
    const int m = (int)u_key.size();
    NRS_TRY(c->ensure(k->a_idx, sizeof(int) * 2 * (size_t)m));
    int* d_idx = k->a_idx.as<int>();
    NRS_HIP(c, hipMemcpyAsync(d_idx, u_slot.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(d_idx + m, u_key.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_klt_copy_templates, dim3(m), dim3(256), 0, c->stream, m, d_idx, d_idx + m, k->levels, k->levels, k->a_levels,
                       k->tI.as<short>(), k->tD.as<short2>(), k->tMean.as<float>(), k->tValid.as<uint8_t>(),
                       k->aI.as<short>(), k->aD.as<short2>(), k->aMean.as<float>(), k->aValid.as<uint8_t>());
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipStreamSynchronize(c->stream));                   // (the caller's index arrays are free again; the archive is complete for any stream)
    for (int i = 0; i < n; ++i) k->a_has[keys[i]] = 1;
    return NRS_OK;
}

extern "C" int nrs_klt_insert_archived(nrs_ctx* c, nrs_ctx* src, int32_t n, const int32_t* keys, const float* xy) {
    if (!c || !src) return NRS_ERR_INVALID;
    if (n < 0 || (n > 0 && (!keys || !xy))) return c->fail(NRS_ERR_INVALID, "nrs_klt_insert_archived: bad argument");
    if (n == 0) return NRS_OK;
    if (c->device != src->device) return c->fail(NRS_ERR_INVALID, "nrs_klt_insert_archived: the two contexts are on different devices");
    KltState* ks = src->klt;
    if (!ks || ks->a_cap == 0) return c->fail(NRS_ERR_STATE, "nrs_klt_insert_archived: the source context holds no archive");
    KltState* k = klt(c);
    if (!k) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    if (k->levels > ks->a_levels) return c->fail(NRS_ERR_INVALID, "nrs_klt_insert_archived: the archive holds %d levels, this tracker needs %d", ks->a_levels, k->levels);
    for (int i = 0; i < n; ++i)
        if (keys[i] < 0 || keys[i] >= ks->a_cap || !ks->a_has[keys[i]]) return c->fail(NRS_ERR_INVALID, "nrs_klt_insert_archived: nothing archived under key %d", keys[i]);
    NRS_HIP(c, hipSetDevice(c->device));
    NRS_TRY(reserve_points(c, k, k->n + n, true));
    NRS_TRY(c->ensure(k->a_idx, sizeof(int) * 2 * (size_t)n));
    std::vector<int> dst((size_t)n);
    for (int i = 0; i < n; ++i) dst[i] = k->n + i;
    int* d_idx = k->a_idx.as<int>();
    NRS_HIP(c, hipMemcpyAsync(d_idx, keys, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(d_idx + n, dst.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(k->prev.as<float>() + 2 * (size_t)k->n, xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_klt_copy_templates, dim3(n), dim3(256), 0, c->stream, n, d_idx, d_idx + n, k->levels, ks->a_levels, k->levels,
                       ks->aI.as<short>(), ks->aD.as<short2>(), ks->aMean.as<float>(), ks->aValid.as<uint8_t>(),
                       k->tI.as<short>(), k->tD.as<short2>(), k->tMean.as<float>(), k->tValid.as<uint8_t>());
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    k->n += n;
    return NRS_OK;
}
