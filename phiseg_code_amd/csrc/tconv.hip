// tf.nn.conv2d_transpose (tfwrapper/layers.py:197-258: transposed_conv2D, default 4x4 kernel, stride 2, SAME -> 2x up-sampling),
// forward and the two gradients, NHWC, filter [kh][kw][Cout][Cin] as TF stores it.  No shipped PHiSeg experiment calls this
// layer: the kernels are the plain direct form (one thread per output element, fp32
// accumulation), correct for any kernel size / stride with SAME padding, not tuned.
//
// [TF 1.12 semantics] conv2d_transpose is the gradient of conv2d with respect to its input: for the stride-s SAME convolution
// that maps [Ho, Wo, Cout] -> [H, W, Cin] (Ho = H * s) the padding is pad_total = max((H - 1) s + k - Ho, 0), pad_before =
// pad_total / 2, so   out[b, oy, ox, co] = sum_{iy, ky: iy s + ky - pad = oy} sum_ci x[b, iy, ix, ci] w[ky, kx, co, ci].
#include "phx_common.h"

namespace {

struct TGeo {
    int B, H, W, Cin, Cout, kh, kw, sh, sw, Ho, Wo, pt, pl;
};

template <typename TX, typename TY>
__global__ void k_tconv_fwd(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                            TY* __restrict__ y, TGeo g, int act) {
    const size_t n = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout);
        size_t r = i / g.Cout;
        const int ox = (int)(r % g.Wo); r /= g.Wo;
        const int oy = (int)(r % g.Ho);
        const int b = (int)(r / g.Ho);
        float acc = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int ty = oy + g.pt - ky;
            if (ty < 0 || ty % g.sh != 0) continue;
            const int iy = ty / g.sh;
            if (iy >= g.H) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int tx = ox + g.pl - kx;
                if (tx < 0 || tx % g.sw != 0) continue;
                const int ix = tx / g.sw;
                if (ix >= g.W) continue;
                const size_t xo = (((size_t)b * g.H + iy) * g.W + ix) * g.Cin;
                const float* wp = w + ((size_t)(ky * g.kw + kx) * g.Cout + co) * g.Cin;
                for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(ldf<TX>(x, xo + ci), wp[ci], acc);
            }
        }
        stf<TY>(y, i, act_fwd(acc, act));
    }
}

// dx[b, iy, ix, ci] = sum_{ky, kx, co} dy[b, iy s + ky - pad, ix s + kx - pad, co] w[ky, kx, co, ci]  (a strided convolution)
template <typename TD, typename TX>
__global__ void k_tconv_dgrad(const TD* __restrict__ dy, const float* __restrict__ w, TX* __restrict__ dx, TGeo g) {
    const size_t n = (size_t)g.B * g.H * g.W * g.Cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % g.Cin);
        size_t r = i / g.Cin;
        const int ix = (int)(r % g.W); r /= g.W;
        const int iy = (int)(r % g.H);
        const int b = (int)(r / g.H);
        float acc = 0.f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int oy = iy * g.sh + ky - g.pt;
            if (oy < 0 || oy >= g.Ho) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int ox = ix * g.sw + kx - g.pl;
                if (ox < 0 || ox >= g.Wo) continue;
                const size_t yo = (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout;
                const float* wp = w + (size_t)(ky * g.kw + kx) * g.Cout * g.Cin + ci;
                for (int co = 0; co < g.Cout; ++co) acc = fmaf(ldf<TD>(dy, yo + co), wp[(size_t)co * g.Cin], acc);
            }
        }
        stf<TX>(dx, i, acc);
    }
}

// dw[ky, kx, co, ci] += sum_{b, iy, ix} x[b, iy, ix, ci] dy[b, iy s + ky - pad, ix s + kx - pad, co]: one block per (ky, kx, co),
// threads over ci and pixel lanes, fixed-order reduction through LDS (no atomics: deterministic by construction)
template <typename TX, typename TD>
__global__ __launch_bounds__(256) void k_tconv_wgrad(const TX* __restrict__ x, const TD* __restrict__ dy, float* __restrict__ dw,
                                                     TGeo g) {
    const int co = blockIdx.x % g.Cout, tap = blockIdx.x / g.Cout, ky = tap / g.kw, kx = tap % g.kw;
    const int P = g.B * g.H * g.W;
    __shared__ float red[256];
    for (int ci0 = 0; ci0 < g.Cin; ci0 += 32) {
        const int ci = ci0 + (threadIdx.x & 31), pl = threadIdx.x >> 5;       // 32 channels x 8 pixel lanes
        float acc = 0.f;
        if (ci < g.Cin)
            for (int p = pl; p < P; p += 8) {
                const int ix = p % g.W, iy = (p / g.W) % g.H, b = p / (g.W * g.H);
                const int oy = iy * g.sh + ky - g.pt, ox = ix * g.sw + kx - g.pl;
                if (oy < 0 || oy >= g.Ho || ox < 0 || ox >= g.Wo) continue;
                acc = fmaf(ldf<TX>(x, (size_t)p * g.Cin + ci), ldf<TD>(dy, (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout + co), acc);
            }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < 32 && ci < g.Cin) {
            float t = 0.f;
            for (int q = 0; q < 8; ++q) t += red[q * 32 + threadIdx.x];
            dw[((size_t)tap * g.Cout + co) * g.Cin + ci] += t;
        }
        __syncthreads();
    }
}

int make_geo(TGeo* g, int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw) {
    PHX_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0, PHX_E_SHAPE, "tconv2d: bad shape");
    g->B = B; g->H = H; g->W = W; g->Cin = Cin; g->Cout = Cout; g->kh = kh; g->kw = kw; g->sh = sh; g->sw = sw;
    g->Ho = H * sh; g->Wo = W * sw;
    const int th = (H - 1) * sh + kh - g->Ho, tw = (W - 1) * sw + kw - g->Wo;
    g->pt = (th > 0 ? th : 0) / 2;
    g->pl = (tw > 0 ? tw : 0) / 2;
    return PHX_OK;
}

}  // namespace

extern "C" {

int phx_tconv2d_fwd(const void* x, int x_dt, const float* w_hwoi, const float* bias, void* y, int y_dt, int B, int H, int W,
                    int Cin, int Cout, int kh, int kw, int sh, int sw, int act, void* stream) {
    TGeo g;
    const int rc = make_geo(&g, B, H, W, Cin, Cout, kh, kw, sh, sw);
    if (rc != PHX_OK) return rc;
    const size_t n = (size_t)B * g.Ho * g.Wo * Cout;
    PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(y_dt, TY, {
        hipLaunchKernelGGL((k_tconv_fwd<TX, TY>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const TX*)x, w_hwoi, bias, (TY*)y, g, act);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_tconv2d_dgrad(const void* dy, int dy_dt, const float* w_hwoi, void* dx, int dx_dt, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, void* stream) {
    TGeo g;
    const int rc = make_geo(&g, B, H, W, Cin, Cout, kh, kw, sh, sw);
    if (rc != PHX_OK) return rc;
    const size_t n = (size_t)B * H * W * Cin;
    PHX_DT_SWITCH(dy_dt, TD, PHX_DT_SWITCH(dx_dt, TX, {
        hipLaunchKernelGGL((k_tconv_dgrad<TD, TX>), dim3(phx_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const TD*)dy, w_hwoi, (TX*)dx, g);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

/* dw_hwoi += filter gradient (accumulates: a variable may be used by several layers); the bias gradient is the channel sum of
 * dy (phx_channel_sum_accumulate) */
int phx_tconv2d_wgrad(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwoi, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, void* stream) {
    TGeo g;
    const int rc = make_geo(&g, B, H, W, Cin, Cout, kh, kw, sh, sw);
    if (rc != PHX_OK) return rc;
    PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(dy_dt, TD, {
        hipLaunchKernelGGL((k_tconv_wgrad<TX, TD>), dim3(kh * kw * Cout), dim3(256), 0, (hipStream_t)stream, (const TX*)x,
                           (const TD*)dy, dw_hwoi, g);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
