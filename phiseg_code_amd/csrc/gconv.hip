// Layers of tfwrapper/layers.py that no shipped PHiSeg experiment calls -- plain direct kernels,
// fp32 accumulation, correct for any shape, not tuned:
//   * general 2-D convolution with stride and dilation, SAME padding, HWIO filter: conv2D(strides=...) (layers.py:94-145) and
//     dilated_conv2D = tf.nn.atrous_conv2d (layers.py:378-425), forward and both gradients;
//   * maxpool2D = tf.nn.max_pool 2x2 / stride 2 / SAME (layers.py:18-28), forward and gradient;
//   * spatial window copy (zero padding / centre crop): pad_to_size (layers.py:625-650) and the crop of crop_and_concat
//     (layers.py:586-622), its own gradient with the offsets negated;
//   * dropout (layers.py:653-668, tf.nn.dropout): keep mask from the Philox stream contract, scaled by 1 / keep_prob.
//
// [TF 1.12 semantics] SAME: Ho = ceil(H / s); with the effective kernel extent ke = (k - 1) d + 1 the total padding is
// max((Ho - 1) s + ke - H, 0), pad_before = total / 2 (the extra row / column goes to the bottom / right).
#include "phx_common.h"
#include "philox.h"

namespace {

struct GGeo {
    int B, H, W, Cin, Cout, kh, kw, sh, sw, dh, dw, Ho, Wo, pt, pl;
};

template <typename TX, typename TY>
__global__ void k_gconv_fwd(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                            TY* __restrict__ y, GGeo g, int act) {
    const size_t n = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout);
        size_t r = i / g.Cout;
        const int ox = (int)(r % g.Wo); r /= g.Wo;
        const int oy = (int)(r % g.Ho);
        const int b = (int)(r / g.Ho);
        float acc = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int iy = oy * g.sh + ky * g.dh - g.pt;
            if (iy < 0 || iy >= g.H) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int ix = ox * g.sw + kx * g.dw - g.pl;
                if (ix < 0 || ix >= g.W) continue;
                const size_t xo = (((size_t)b * g.H + iy) * g.W + ix) * g.Cin;
                const float* wp = w + (size_t)(ky * g.kw + kx) * g.Cin * g.Cout + co;
                for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(ldf<TX>(x, xo + ci), wp[(size_t)ci * g.Cout], acc);
            }
        }
        stf<TY>(y, i, act_fwd(acc, act));
    }
}

// dx[b, iy, ix, ci] = sum over (ky, kx, co) with (iy + pt - ky d) divisible by s of dy[b, (iy + pt - ky d) / s, ., co] w[ky, kx, ci, co]
template <typename TD, typename TX>
__global__ void k_gconv_dgrad(const TD* __restrict__ dy, const float* __restrict__ w, TX* __restrict__ dx, GGeo g) {
    const size_t n = (size_t)g.B * g.H * g.W * g.Cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % g.Cin);
        size_t r = i / g.Cin;
        const int ix = (int)(r % g.W); r /= g.W;
        const int iy = (int)(r % g.H);
        const int b = (int)(r / g.H);
        float acc = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int ty = iy + g.pt - ky * g.dh;
            if (ty < 0 || ty % g.sh != 0) continue;
            const int oy = ty / g.sh;
            if (oy >= g.Ho) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int tx = ix + g.pl - kx * g.dw;
                if (tx < 0 || tx % g.sw != 0) continue;
                const int ox = tx / g.sw;
                if (ox >= g.Wo) continue;
                const size_t yo = (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout;
                const float* wp = w + ((size_t)(ky * g.kw + kx) * g.Cin + ci) * g.Cout;
                for (int co = 0; co < g.Cout; ++co) acc = fmaf(ldf<TD>(dy, yo + co), wp[co], acc);
            }
        }
        stf<TX>(dx, i, acc);
    }
}

// dw[ky, kx, ci, co] += sum over output pixels of x[b, oy s + ky d - pt, ox s + kx d - pl, ci] dy[b, oy, ox, co]: one block per
// (tap, ci), threads = 32 output channels x 8 pixel lanes, fixed-order LDS reduction (deterministic)
template <typename TX, typename TD>
__global__ __launch_bounds__(256) void k_gconv_wgrad(const TX* __restrict__ x, const TD* __restrict__ dy, float* __restrict__ dw,
                                                     GGeo g) {
    const int ci = blockIdx.x % g.Cin, tap = blockIdx.x / g.Cin, ky = tap / g.kw, kx = tap % g.kw;
    const int P = g.B * g.Ho * g.Wo;
    __shared__ float red[256];
    for (int co0 = 0; co0 < g.Cout; co0 += 32) {
        const int co = co0 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
        float acc = 0.f;
        if (co < g.Cout)
            for (int p = pl; p < P; p += 8) {
                const int ox = p % g.Wo, oy = (p / g.Wo) % g.Ho, b = p / (g.Wo * g.Ho);
                const int iy = oy * g.sh + ky * g.dh - g.pt, ix = ox * g.sw + kx * g.dw - g.pl;
                if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
                acc = fmaf(ldf<TX>(x, (((size_t)b * g.H + iy) * g.W + ix) * g.Cin + ci), ldf<TD>(dy, (size_t)p * g.Cout + co), acc);
            }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < 32 && co < g.Cout) {
            float t = 0.f;
            for (int q = 0; q < 8; ++q) t += red[q * 32 + threadIdx.x];
            dw[((size_t)tap * g.Cin + ci) * g.Cout + co] += t;
        }
        __syncthreads();
    }
}

int make_ggeo(GGeo* g, int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw) {
    PHX_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0,
                PHX_E_SHAPE, "gconv2d: bad shape");
    g->B = B; g->H = H; g->W = W; g->Cin = Cin; g->Cout = Cout; g->kh = kh; g->kw = kw; g->sh = sh; g->sw = sw; g->dh = dh; g->dw = dw;
    g->Ho = (H + sh - 1) / sh; g->Wo = (W + sw - 1) / sw;
    const int th = (g->Ho - 1) * sh + (kh - 1) * dh + 1 - H, tw = (g->Wo - 1) * sw + (kw - 1) * dw + 1 - W;
    g->pt = (th > 0 ? th : 0) / 2;
    g->pl = (tw > 0 ? tw : 0) / 2;
    return PHX_OK;
}

// ---- max pool 2x2 / stride 2 / SAME: window rows 2 oy, 2 oy + 1 (bottom / right padding only), first maximum wins ------------
template <typename T>
__global__ void k_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t n = (size_t)B * Ho * Wo * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float m = -INFINITY;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                if (iy < H && ix < W) m = fmaxf(m, ldf<T>(x, (((size_t)b * H + iy) * W + ix) * C + c));
            }
        stf<T>(y, i, m);
    }
}
// dx = dy routed to the first element of the window (row-major) that equals its maximum
template <typename T>
__global__ void k_maxpool_bwd(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t n = (size_t)B * Ho * Wo * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float m = -INFINITY;
        int am = 0;
        for (int q = 0; q < 4; ++q) {
            const int iy = 2 * oy + (q >> 1), ix = 2 * ox + (q & 1);
            if (iy < H && ix < W) {
                const float v = ldf<T>(x, (((size_t)b * H + iy) * W + ix) * C + c);
                if (v > m) { m = v; am = q; }
            }
        }
        const float g = ldf<T>(dy, i);
        for (int q = 0; q < 4; ++q) {
            const int iy = 2 * oy + (q >> 1), ix = 2 * ox + (q & 1);
            if (iy < H && ix < W) stf<T>(dx, (((size_t)b * H + iy) * W + ix) * C + c, q == am ? g : 0.f);
        }
    }
}

// dst[b, y, x, :] = src[b, y + oy, x + ox, :] inside the source, 0 outside (oy, ox < 0: zero padding; > 0: crop)
template <typename T>
__global__ void k_spatial_window(const T* __restrict__ src, T* __restrict__ dst, int B, int Hs, int Ws, int Hd, int Wd, int C, int oy,
                                 int ox) {
    const size_t n = (size_t)B * Hd * Wd * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % Wd); r /= Wd;
        const int y = (int)(r % Hd);
        const int b = (int)(r / Hd);
        const int sy = y + oy, sx = x + ox;
        const bool in = sy >= 0 && sy < Hs && sx >= 0 && sx < Ws;
        stf<T>(dst, i, in ? ldf<T>(src, (((size_t)b * Hs + sy) * Ws + sx) * C + c) : 0.f);
    }
}

// general strided window with a channel offset (the skip path of the residual units, layers.py:465-470: tf.pad along the channel
// axis and identity[:, ::2, ::2, :]):  dst[b, y, x, c] = src[b, y sy + oy, x sx + ox, c + oc] inside the source, 0 outside
struct WGeo {
    int B, Hs, Ws, Cs, Hd, Wd, Cd, sy, sx, oy, ox, oc;
};
template <typename T>
__global__ void k_window4_fwd(const T* __restrict__ src, T* __restrict__ dst, WGeo g) {
    const size_t n = (size_t)g.B * g.Hd * g.Wd * g.Cd;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.Cd);
        size_t r = i / g.Cd;
        const int x = (int)(r % g.Wd); r /= g.Wd;
        const int y = (int)(r % g.Hd);
        const int b = (int)(r / g.Hd);
        const int yy = y * g.sy + g.oy, xx = x * g.sx + g.ox, cc = c + g.oc;
        const bool in = yy >= 0 && yy < g.Hs && xx >= 0 && xx < g.Ws && cc >= 0 && cc < g.Cs;
        stf<T>(dst, i, in ? ldf<T>(src, (((size_t)b * g.Hs + yy) * g.Ws + xx) * g.Cs + cc) : 0.f);
    }
}
// its gradient: dsrc[b, Y, X, C] = ddst[b, (Y - oy) / sy, (X - ox) / sx, C - oc] where that position exists, else 0
template <typename T>
__global__ void k_window4_bwd(const T* __restrict__ ddst, T* __restrict__ dsrc, WGeo g) {
    const size_t n = (size_t)g.B * g.Hs * g.Ws * g.Cs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int C = (int)(i % g.Cs);
        size_t r = i / g.Cs;
        const int X = (int)(r % g.Ws); r /= g.Ws;
        const int Y = (int)(r % g.Hs);
        const int b = (int)(r / g.Hs);
        const int ty = Y - g.oy, tx = X - g.ox, c = C - g.oc;
        float v = 0.f;
        if (ty >= 0 && tx >= 0 && ty % g.sy == 0 && tx % g.sx == 0 && c >= 0 && c < g.Cd) {
            const int y = ty / g.sy, x = tx / g.sx;
            if (y < g.Hd && x < g.Wd) v = ldf<T>(ddst, (((size_t)b * g.Hd + y) * g.Wd + x) * g.Cd + c);
        }
        stf<T>(dsrc, i, v);
    }
}
// y = act(a + b)   (tf.add + activation of the residual units, layers.py:474-475)
template <typename T>
__global__ void k_add_act(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n, int act) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        stf<T>(y, i, act_fwd(ldf<T>(a, i) + ldf<T>(b, i), act));
}

// y = x * keep[b, e] / keep_prob, keep = (u < keep_prob), u = (word (e % 4) of Philox block e / 4 of (seed, step, stream, sample b) >> 8) / 2^24
// (the same keep mask is recomputed by the backward launch: dx = dy * keep / keep_prob)
template <typename T>
__global__ void k_dropout(const T* __restrict__ x, T* __restrict__ y, size_t per_sample, int B, float keep_prob, uint64_t seed,
                          const int32_t* __restrict__ step_dev, uint32_t stream_id, int sample_offset) {
    const size_t nblk = (per_sample + 3) / 4, n = (size_t)B * nblk;
    const float inv = 1.f / keep_prob;
    const uint32_t step = (uint32_t)*step_dev;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / nblk);
        const size_t blk = i % nblk;
        unsigned o[4];
        philox4x32_10((uint32_t)blk, (uint32_t)(b + sample_offset), stream_id, step, (uint32_t)seed, (uint32_t)(seed >> 32), o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = blk * 4 + k;
            if (e < per_sample) {
                const bool keep = (float)(o[k] >> 8) * (1.0f / 16777216.0f) < keep_prob;       // 24-bit uniform: exact in fp32
                const size_t idx = (size_t)b * per_sample + e;
                stf<T>(y, idx, keep ? ldf<T>(x, idx) * inv : 0.f);
            }
        }
    }
}

}  // namespace

extern "C" {

int phx_gconv2d_out_size(int H, int W, int sh, int sw, int* Ho, int* Wo) {
    PHX_REQUIRE(H > 0 && W > 0 && sh > 0 && sw > 0 && Ho && Wo, PHX_E_INVAL, "gconv2d_out_size: bad argument");
    *Ho = (H + sh - 1) / sh;
    *Wo = (W + sw - 1) / sw;
    return PHX_OK;
}

int phx_gconv2d_fwd(const void* x, int x_dt, const float* w_hwio, const float* bias, void* y, int y_dt, int B, int H, int W,
                    int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw, int act, void* stream) {
    GGeo g;
    const int rc = make_ggeo(&g, B, H, W, Cin, Cout, kh, kw, sh, sw, dh, dw);
    if (rc != PHX_OK) return rc;
    const size_t n = (size_t)B * g.Ho * g.Wo * Cout;
    PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(y_dt, TY, {
        hipLaunchKernelGGL((k_gconv_fwd<TX, TY>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const TX*)x, w_hwio, bias, (TY*)y, g, act);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_gconv2d_dgrad(const void* dy, int dy_dt, const float* w_hwio, void* dx, int dx_dt, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, int dh, int dw, void* stream) {
    GGeo g;
    const int rc = make_ggeo(&g, B, H, W, Cin, Cout, kh, kw, sh, sw, dh, dw);
    if (rc != PHX_OK) return rc;
    const size_t n = (size_t)B * H * W * Cin;
    PHX_DT_SWITCH(dy_dt, TD, PHX_DT_SWITCH(dx_dt, TX, {
        hipLaunchKernelGGL((k_gconv_dgrad<TD, TX>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const TD*)dy, w_hwio, (TX*)dx, g);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

/* dw_hwio += filter gradient; the bias gradient is the channel sum of dy (phx_channel_sum_accumulate) */
int phx_gconv2d_wgrad(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwio, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, int dh, int dw, void* stream) {
    GGeo g;
    const int rc = make_ggeo(&g, B, H, W, Cin, Cout, kh, kw, sh, sw, dh, dw);
    if (rc != PHX_OK) return rc;
    PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(dy_dt, TD, {
        hipLaunchKernelGGL((k_gconv_wgrad<TX, TD>), dim3(kh * kw * Cin), dim3(256), 0, (hipStream_t)stream, (const TX*)x,
                           (const TD*)dy, dw_hwio, g);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_maxpool2x2_fwd(const void* x, int dt, void* y, int B, int H, int W, int C, void* stream) {
    PHX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, PHX_E_SHAPE, "maxpool2x2: bad shape");
    const size_t n = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_maxpool_fwd<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                           B, H, W, C);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_maxpool2x2_bwd(const void* x, const void* dy, int dt, void* dx, int B, int H, int W, int C, void* stream) {
    PHX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, PHX_E_SHAPE, "maxpool2x2: bad shape");
    const size_t n = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_maxpool_bwd<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                           (const T*)dy, (T*)dx, B, H, W, C);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_spatial_window(const void* src, void* dst, int dt, int B, int Hs, int Ws, int Hd, int Wd, int C, int off_y, int off_x,
                       void* stream) {
    PHX_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0, PHX_E_SHAPE, "spatial_window: bad shape");
    const size_t n = (size_t)B * Hd * Wd * C;
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_spatial_window<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)src,
                           (T*)dst, B, Hs, Ws, Hd, Wd, C, off_y, off_x);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

static int make_wgeo(WGeo* g, int B, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int sy, int sx, int oy, int ox, int oc) {
    PHX_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cs > 0 && Hd > 0 && Wd > 0 && Cd > 0 && sy > 0 && sx > 0, PHX_E_SHAPE, "window4: bad shape");
    g->B = B; g->Hs = Hs; g->Ws = Ws; g->Cs = Cs; g->Hd = Hd; g->Wd = Wd; g->Cd = Cd; g->sy = sy; g->sx = sx; g->oy = oy; g->ox = ox; g->oc = oc;
    return PHX_OK;
}
int phx_window4_fwd(const void* src, void* dst, int dt, int B, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int sy, int sx,
                    int off_y, int off_x, int off_c, void* stream) {
    WGeo g;
    const int rc = make_wgeo(&g, B, Hs, Ws, Cs, Hd, Wd, Cd, sy, sx, off_y, off_x, off_c);
    if (rc != PHX_OK) return rc;
    const size_t n = (size_t)B * Hd * Wd * Cd;
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_window4_fwd<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)src,
                           (T*)dst, g);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_window4_bwd(const void* ddst, void* dsrc, int dt, int B, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int sy, int sx,
                    int off_y, int off_x, int off_c, void* stream) {
    WGeo g;
    const int rc = make_wgeo(&g, B, Hs, Ws, Cs, Hd, Wd, Cd, sy, sx, off_y, off_x, off_c);
    if (rc != PHX_OK) return rc;
    const size_t n = (size_t)B * Hs * Ws * Cs;
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_window4_bwd<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)ddst,
                           (T*)dsrc, g);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_add_act(const void* a, const void* b, void* y, int dt, size_t n, int act, void* stream) {
    PHX_REQUIRE(a && b && y && n > 0, PHX_E_INVAL, "add_act: bad argument");
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_add_act<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)a,
                           (const T*)b, (T*)y, n, act);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_dropout(const void* x, void* y, int dt, size_t per_sample, int B, float keep_prob, uint64_t seed, const int32_t* step_dev,
                int stream_id, int sample_offset, void* stream) {
    PHX_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f && B > 0 && per_sample > 0 && step_dev, PHX_E_INVAL, "dropout: bad argument");
    const size_t n = (size_t)B * ((per_sample + 3) / 4);
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_dropout<T>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                           per_sample, B, keep_prob, seed, step_dev, (uint32_t)stream_id, sample_offset);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
