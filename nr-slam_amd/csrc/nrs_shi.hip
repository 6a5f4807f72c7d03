// f3: Shi-Tomasi corner extraction (reference modules/features/shi_tomasi.cc:38-409) and the caller's
// mask filter (modules/tracking/tracking.cc:118-134) behind the C ABI.
//
// The reference makes ONE sequential pass over the image with rolling three-row pointers and running
// column / row sums; what that pass leaves in its buffers is restated here per output cell (oracle:
// oracle/shi_oracle.py keeps both forms and holds them to each other), so that every cell is one
// independent thread:
//   k_shi_grad    int16 un-normalised Sobel sums exactly as the pass stores them: gradient row i >= 4 is
//                 built from image rows i-2, i-1, i; rows 2 and 3 from (1,2,2) and (2,2,3) (:256-270);
//                 row 0 has its own two-row formula over columns 1..rows-2 (:170-192); X columns 0 and
//                 cols-1 and Y rows 0 and rows-1 are never written (zero).
//   k_shi_score   score row r (0..rows-4), columns 1..cols-2: float32 min eigenvalue of the 3x3 box
//                 tensor of gradient rows r..r+2 (:293-313 writes scores.ptr(i-2)).  All tensor sums are
//                 integers < 2^24: exact in float32 in any order; the eigenvalue is the reference's
//                 float32 expression with separate multiplies / adds and a correctly rounded sqrtf.
//   k_shi_last    the last-row pass (:318-344): overwrites score row rows-4, columns 1..rows-2, from
//                 gradient rows rows-3..rows-1, reading the last row's X gradients BEFORE that pass has
//                 rewritten them (the previous call's values, except column 1); then k_shi_last_grad
//                 stores the new last row.  Buffers therefore persist between calls, as in the reference.
//   k_shi_mark    already extracted keypoints -> score -1 (:93-96)
//   k_shi_nms     IsLocalMaximum (:123-160) per pixel; k_shi_row_count / k_shi_row_scan / k_shi_emit
//                 compact the maxima in the row-major order of GetKeyPoints (:77-88) and number them.
// The pass indexes columns up to rows-1 in its first / last row loops: width >= height is required
// (the reference writes out of bounds otherwise).
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>
#include "nrs_ctx.hpp"

namespace nrs {

struct ShiState {
    int nms = 5;
    int w = 0, h = 0;
    int next_id = 0;
    DevBuf img, xg, yg, scores, flags, rowcnt, rowoff, prev, out_xy, out_id;
};

__device__ inline int shi_I(const uint8_t* __restrict__ im, int w, int r, int c) { return (int)im[(size_t)r * w + c]; }

// 1-2-1 row sum of image row r at column c; 2-2 at the two ends (:205-207,:247-249)
__device__ inline int shi_rs(const uint8_t* __restrict__ im, int w, int r, int c) {
    if (c == 0) return 2 * shi_I(im, w, r, 0) + 2 * shi_I(im, w, r, 1);
    if (c == w - 1) return 2 * shi_I(im, w, r, w - 1) + 2 * shi_I(im, w, r, w - 2);
    return shi_I(im, w, r, c - 1) + 2 * shi_I(im, w, r, c) + shi_I(im, w, r, c + 1);
}

// image rows behind gradient row i (1 <= i <= rows-2)
__device__ inline void shi_trip(int i, int& a, int& b, int& c) {
    if (i == 1) { a = 0; b = 1; c = 2; }
    else if (i == 2) { a = 1; b = 2; c = 2; }
    else if (i == 3) { a = 2; b = 2; c = 3; }
    else { a = i - 2; b = i - 1; c = i; }
}

__global__ void k_shi_grad(const uint8_t* __restrict__ im, int w, int h, int16_t* __restrict__ xg, int16_t* __restrict__ yg) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= w || r >= h - 1) return;                              // the last row belongs to the last-row pass
    int gx = 0, gy = 0;
    if (r == 0) {
        if (c >= 1 && c <= h - 2) {
            auto C0 = [&](int k) { return k <= 2 ? 3 * shi_I(im, w, 0, k) + shi_I(im, w, 1, k) : 2 * shi_I(im, w, 0, k) + 2 * shi_I(im, w, 1, k); };
            gx = C0(c + 1) - C0(c - 1);
        }
    } else {
        int a, b, d;
        shi_trip(r, a, b, d);
        if (c >= 1 && c <= w - 2) {
            auto Cs = [&](int k) { return shi_I(im, w, a, k) + 2 * shi_I(im, w, b, k) + shi_I(im, w, d, k); };
            gx = Cs(c + 1) - Cs(c - 1);
        }
        if (r == 1) {
            int top = shi_rs(im, w, 0, c);
            if (c == 1) top = shi_I(im, w, 0, 0) + 2 * shi_I(im, w, 0, 1) + shi_I(im, w, 2, 2);   // (:221 reads pIm[2][2])
            gy = shi_rs(im, w, 2, c) - top;
        } else if (r == 2) gy = shi_rs(im, w, 2, c) - shi_rs(im, w, 1, c);
        else if (r == 3) gy = shi_rs(im, w, 3, c) - shi_rs(im, w, 2, c);
        else gy = shi_rs(im, w, r, c) - shi_rs(im, w, r - 2, c);
    }
    xg[(size_t)r * w + c] = (int16_t)gx;
    yg[(size_t)r * w + c] = (int16_t)gy;
}

// ComputeMinEigenValue (:402-409): float32, separate roundings
__device__ inline float shi_eig(int sxx, int sxy, int syy) {
    // plain operators under contract(off): the __f*_rn intrinsics are header inlines whose operations keep
    // the header's contraction flag and do get fused; sqrtf is correctly rounded with the build's
    // -fhip-fp32-correctly-rounded-divide-sqrt (__fsqrt_rn is not)
#pragma clang fp contract(off)
    const float inv = 1.f / 9.f;
    const float t0 = (float)sxx * inv, t1 = (float)sxy * inv, t2 = (float)syy * inv;
    const float tr = t0 + t2;
    const float p02 = t0 * t2, p11 = t1 * t1;
    const float det = p02 - p11;
    const float trtr = tr * tr, det4 = 4.f * det;
    const float root = trtr - det4;
    const float diff = tr - sqrtf(root);
    return diff * 0.5f;
}

__global__ void k_shi_score(const int16_t* __restrict__ xg, const int16_t* __restrict__ yg, int w, int h, float* __restrict__ sc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c < 1 || c > w - 2 || r > h - 4) return;
    int sxx = 0, sxy = 0, syy = 0;
#pragma unroll
    for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const int x = xg[(size_t)(r + dr) * w + c + dc], y = yg[(size_t)(r + dr) * w + c + dc];
            sxx += x * x; sxy += x * y; syy += y * y;
        }
    sc[(size_t)r * w + c] = shi_eig(sxx, sxy, syy);
}

// X gradient the last-row pass stores at column j (1 <= j <= rows-2): seq[j+2] - seq[j], where seq is the
// order in which the pass forms its column sums: three two-row sums, then the three-row sums from column 2 on
__device__ inline int shi_last_x(const uint8_t* __restrict__ im, int w, int h, int j) {
    int a, b, d;
    shi_trip(h - 2, a, b, d);
    auto seq = [&](int k) {
        return k <= 2 ? 3 * shi_I(im, w, b, k) + shi_I(im, w, d, k)
                      : shi_I(im, w, a, k - 1) + 2 * shi_I(im, w, b, k - 1) + shi_I(im, w, d, k - 1);
    };
    return seq(j + 2) - seq(j);
}

__global__ void k_shi_last(const uint8_t* __restrict__ im, const int16_t* __restrict__ xg, const int16_t* __restrict__ yg,
                           int w, int h, float* __restrict__ sc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (j > h - 2) return;
    const int new1 = (int)(int16_t)shi_last_x(im, w, h, 1);
    int sxx = 0, sxy = 0, syy = 0;
    for (int dc = -1; dc <= 1; ++dc) {
        const int cc = j + dc;
        for (int dr = 0; dr < 2; ++dr) {
            const int x = xg[(size_t)(h - 3 + dr) * w + cc], y = yg[(size_t)(h - 3 + dr) * w + cc];
            sxx += x * x; sxy += x * y; syy += y * y;
        }
        const int xl = cc == 1 ? new1 : (int)xg[(size_t)(h - 1) * w + cc];      // Y gradient of the last row is zero
        sxx += xl * xl;
    }
    sc[(size_t)(h - 4) * w + j] = shi_eig(sxx, sxy, syy);
}

__global__ void k_shi_last_grad(const uint8_t* __restrict__ im, int w, int h, int16_t* __restrict__ xg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (j > h - 2) return;
    xg[(size_t)(h - 1) * w + j] = (int16_t)shi_last_x(im, w, h, j);
}

__global__ void k_shi_mark(const int* __restrict__ cells, int n, float* __restrict__ sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sc[cells[i]] = -1.f;
}

__global__ void k_shi_nms(const float* __restrict__ sc, int w, int h, int nms, uint8_t* __restrict__ flags) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= w) return;
    const float cur = sc[(size_t)r * w + c];
    bool ok = !(cur == -1.f) && !(cur < 80.f);                     // a NaN score passes both tests, as in the reference
    // IsLocalMaximum returns false for a mark anywhere in the 31x31 window OR a larger score in the inner
    // window; the order of the two scans does not matter: the (small) inner window goes first, it
    // eliminates all but the local maxima, and only those look at the 961 cells for marks
    if (ok) {
        const int r0 = max(0, r - nms), r1 = min(h - 1, r + nms), c0 = max(0, c - nms), c1 = min(w - 1, c + nms);
        for (int i = r0; i <= r1 && ok; ++i)
            for (int j = c0; j <= c1; ++j) {
                const float v = sc[(size_t)i * w + j];
                if (v == -1.f || v > cur) { ok = false; break; }
            }
    }
    if (ok) {
        const int r0 = max(0, r - 15), r1 = min(h - 1, r + 15), c0 = max(0, c - 15), c1 = min(w - 1, c + 15);
        for (int i = r0; i <= r1 && ok; ++i)
            for (int j = c0; j <= c1; ++j)
                if (sc[(size_t)i * w + j] == -1.f) { ok = false; break; }
    }
    flags[(size_t)r * w + c] = ok ? 1 : 0;
}

__global__ void k_shi_row_count(const uint8_t* __restrict__ flags, int w, int* __restrict__ rowcnt) {
    __shared__ int lds[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    int n = 0;
    for (int c = tid; c < w; c += 256) n += flags[(size_t)r * w + c];
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((tid & 63) == 0) lds[tid >> 6] = n;
    __syncthreads();
    if (tid == 0) rowcnt[r] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ void k_shi_row_scan(const int* __restrict__ rowcnt, int h, int* __restrict__ rowoff /* h + 1 */) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int r = 0; r < h; ++r) { rowoff[r] = s; s += rowcnt[r]; }
        rowoff[h] = s;
    }
}

// one workgroup per row; columns in ascending order: 256-column chunks, ballot prefix inside a chunk
__global__ void k_shi_emit(const uint8_t* __restrict__ flags, int w, const int* __restrict__ rowoff, int first_id,
                           float* __restrict__ out_xy, int* __restrict__ out_id) {
    __shared__ int wcnt[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = rowoff[r];
    for (int c0 = 0; c0 < w; c0 += 256) {
        const int c = c0 + tid;
        const bool f = c < w && flags[(size_t)r * w + c] != 0;
        const unsigned long long m = __ballot(f);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wave; ++q) before += wcnt[q];
        const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (f) {
            const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
            out_xy[2 * (size_t)pos] = (float)c;
            out_xy[2 * (size_t)pos + 1] = (float)r;
            out_id[pos] = first_id + pos;
        }
        base += total;
        __syncthreads();
    }
}

void shi_free(nrs_ctx* c) {
    if (!c->shi) return;
    ShiState* s = c->shi;
    DevBuf* bufs[] = {&s->img, &s->xg, &s->yg, &s->scores, &s->flags, &s->rowcnt, &s->rowoff, &s->prev, &s->out_xy, &s->out_id};
    for (auto b : bufs) c->release(*b);
    delete s;
    c->shi = nullptr;
}

static int shi_resize(nrs_ctx* c, ShiState* s, int w, int h) {          // ShiTomasi::ResizeBuffers (:56-66): zeroed buffers
    const size_t n = (size_t)w * h;
    NRS_TRY(c->ensure(s->img, n));
    NRS_TRY(c->ensure(s->xg, 2 * n));
    NRS_TRY(c->ensure(s->yg, 2 * n));
    NRS_TRY(c->ensure(s->scores, 4 * n));
    NRS_TRY(c->ensure(s->flags, n));
    NRS_TRY(c->ensure(s->rowcnt, sizeof(int) * (size_t)h));
    NRS_TRY(c->ensure(s->rowoff, sizeof(int) * ((size_t)h + 1)));
    NRS_HIP(c, hipMemsetAsync(s->xg.p, 0, 2 * n, c->stream));
    NRS_HIP(c, hipMemsetAsync(s->yg.p, 0, 2 * n, c->stream));
    NRS_HIP(c, hipMemsetAsync(s->scores.p, 0, 4 * n, c->stream));
    s->w = w; s->h = h;
    return NRS_OK;
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_shi_configure(nrs_ctx* c, int32_t nms_window) {
    if (!c) return NRS_ERR_INVALID;
    if (nms_window < 0 || nms_window > 15) return c->fail(NRS_ERR_INVALID, "nrs_shi_configure: window must be in [0, 15]");
    shi_free(c);
    c->shi = new (std::nothrow) ShiState();
    if (!c->shi) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    c->shi->nms = nms_window;
    return NRS_OK;
}

extern "C" int nrs_shi_extract(nrs_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                               const uint8_t* mask, int32_t mask_stride, int32_t n_prev, const float* prev_xy,
                               int32_t capacity, float* out_xy, int32_t* out_id, int32_t* n_out) {
    if (!c) return NRS_ERR_INVALID;
    if (!img || !n_out || w < h || h < 5 || stride < w || n_prev < 0 || (n_prev > 0 && !prev_xy) || capacity < 0 ||
        (capacity > 0 && (!out_xy || !out_id)) || (mask && mask_stride < w))
        return c->fail(NRS_ERR_INVALID, "nrs_shi_extract: bad argument (width >= height >= 5 required)");
    NRS_HIP(c, hipSetDevice(c->device));
    if (!c->shi) NRS_TRY(nrs_shi_configure(c, 5));                  // ShiTomasi::Options default (shi_tomasi.h:33)
    ShiState* s = c->shi;
    // cells of the already extracted keypoints: round() is half away from zero (:94-95)
    std::vector<int> cells((size_t)n_prev);
    for (int i = 0; i < n_prev; ++i) {
        const long x = lroundf(prev_xy[2 * i]), y = lroundf(prev_xy[2 * i + 1]);
        if (!(x >= 0 && x < w && y >= 0 && y < h)) return c->fail(NRS_ERR_INVALID, "nrs_shi_extract: keypoint %d outside the image", i);
        cells[i] = (int)(y * w + x);
    }
    if (s->w != w || s->h != h) NRS_TRY(shi_resize(c, s, w, h));
    NRS_HIP(c, hipMemcpy2DAsync(s->img.p, (size_t)w, img, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, c->stream));
    const dim3 blk(256), grid((w + 255) / 256, h);
    const uint8_t* im = s->img.as<uint8_t>();
    int16_t* xg = s->xg.as<int16_t>();
    int16_t* yg = s->yg.as<int16_t>();
    float* sc = s->scores.as<float>();
    hipLaunchKernelGGL(k_shi_grad, grid, blk, 0, c->stream, im, w, h, xg, yg);
    hipLaunchKernelGGL(k_shi_score, grid, blk, 0, c->stream, xg, yg, w, h, sc);
    hipLaunchKernelGGL(k_shi_last, dim3((h + 255) / 256), blk, 0, c->stream, im, xg, yg, w, h, sc);
    hipLaunchKernelGGL(k_shi_last_grad, dim3((h + 255) / 256), blk, 0, c->stream, im, w, h, xg);
    if (n_prev > 0) {
        NRS_TRY(c->ensure(s->prev, sizeof(int) * (size_t)n_prev));
        NRS_HIP(c, hipMemcpyAsync(s->prev.p, cells.data(), sizeof(int) * (size_t)n_prev, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_shi_mark, dim3((n_prev + 255) / 256), blk, 0, c->stream, s->prev.as<int>(), n_prev, sc);
    }
    hipLaunchKernelGGL(k_shi_nms, grid, blk, 0, c->stream, sc, w, h, s->nms, s->flags.as<uint8_t>());
    hipLaunchKernelGGL(k_shi_row_count, dim3(h), blk, 0, c->stream, s->flags.as<uint8_t>(), w, s->rowcnt.as<int>());
    hipLaunchKernelGGL(k_shi_row_scan, dim3(1), dim3(64), 0, c->stream, s->rowcnt.as<int>(), h, s->rowoff.as<int>());
    NRS_HIP(c, hipGetLastError());
    int total = 0;
    NRS_HIP(c, hipMemcpyAsync(&total, s->rowoff.as<int>() + h, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));                   // (the cells vector is consumed by now, too)
    std::vector<float> xy(2 * (size_t)total);
    std::vector<int> ids((size_t)total);
    if (total > 0) {
        NRS_TRY(c->ensure(s->out_xy, sizeof(float) * 2 * (size_t)total));
        NRS_TRY(c->ensure(s->out_id, sizeof(int) * (size_t)total));
        hipLaunchKernelGGL(k_shi_emit, dim3(h), blk, 0, c->stream, s->flags.as<uint8_t>(), w, s->rowoff.as<int>(), s->next_id,
                           s->out_xy.as<float>(), s->out_id.as<int>());
        NRS_HIP(c, hipGetLastError());
        NRS_HIP(c, hipMemcpyAsync(xy.data(), s->out_xy.p, sizeof(float) * xy.size(), hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipMemcpyAsync(ids.data(), s->out_id.p, sizeof(int) * ids.size(), hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
    }
    s->next_id += total;                                           // ids are consumed before the mask filter (tracking.cc:121-131)
    int kept = 0;
    for (int i = 0; i < total; ++i) {
        if (mask && !mask[(size_t)xy[2 * i + 1] * mask_stride + (size_t)xy[2 * i]]) continue;
        if (kept < capacity) { out_xy[2 * kept] = xy[2 * i]; out_xy[2 * kept + 1] = xy[2 * i + 1]; out_id[kept] = ids[i]; }
        ++kept;
    }
    *n_out = kept;                                                 // > capacity: the output was truncated
    return NRS_OK;
}

extern "C" int nrs_shi_buffers(nrs_ctx* c, float* scores, int16_t* xgrad, int16_t* ygrad) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->shi || c->shi->w == 0) return c->fail(NRS_ERR_STATE, "no image has been processed");
    ShiState* s = c->shi;
    const size_t n = (size_t)s->w * s->h;
    if (scores) NRS_HIP(c, hipMemcpyAsync(scores, s->scores.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
    if (xgrad) NRS_HIP(c, hipMemcpyAsync(xgrad, s->xg.p, 2 * n, hipMemcpyDeviceToHost, c->stream));
    if (ygrad) NRS_HIP(c, hipMemcpyAsync(ygrad, s->yg.p, 2 * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}
