// Generic fp32-math convolution (tf.nn.conv2d SAME stride 1, tfwrapper/layers.py:122-135) for every
// shape the bf16 MFMA kernels do not take: the fp32 parity path, Cin in {1,2,3} image / latent inputs,
// the 1x1 / 3x3 mu, sigma, y_lvl heads (Cout = zdim0 or nlabels) and prob_unet2D's 1x1 recombination.
// One thread = one output pixel x COT output channels; input patch and filter slab staged in LDS.
#include "phx_common.h"


struct TileGeo {
    int tws, ths, tb;       // tile = tb images x (1<<ths) rows x (1<<tws) cols = 256 pixels
    int tiles_x, tiles_y, tiles_b;
};

static TileGeo make_geo(int B, int H, int W) {
    TileGeo g;
    int tw = 1, th = 1;
    g.tws = g.ths = 0;
    while (tw < W && tw < 16) { tw <<= 1; g.tws++; }
    while (th < H && th < 16) { th <<= 1; g.ths++; }
    g.tb = 256 / (tw * th);
    g.tiles_x = (W + tw - 1) / tw;
    g.tiles_y = (H + th - 1) / th;
    // tiny maps (2x2 .. 8x8): fewer images per tile so that at least ~64 blocks exist (threads beyond tb idle)
    while (g.tb > 1 && g.tiles_x * g.tiles_y * ((B + g.tb - 1) / g.tb) < 64) g.tb >>= 1;
    g.tiles_b = (B + g.tb - 1) / g.tb;
    return g;
}

// CIT = input channels per LDS stage (8; 32 for wide inputs with <= 4 output channels: a quarter of the serial stages)
template <typename TI, typename TO, int COT, int KS, int CIT>
__global__ __launch_bounds__(256) void k_conv_direct(const TI* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, TO* __restrict__ y,
                                                     float* __restrict__ stats, int B, int H, int W, int Cx, int Cy,
                                                     int wCin, int wCout, int act, int tflip, TileGeo g) {
    constexpr int PAD = KS / 2;
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int pw = tw + KS - 1, ph = th + KS - 1;
    const int npatch = g.tb * ph * pw;
    extern __shared__ float smem[];
    float* sx = smem;                         // [CIT][npatch]
    float* sw = smem + CIT * npatch;         // [KS*KS][CIT][COT]

    int t = blockIdx.x;
    const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
    const int b0 = t * g.tb;
    const int co0 = blockIdx.y * COT;

    const int m = threadIdx.x;
    const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
    const int ox = tx0 + lx, oy = ty0 + ly, ob = b0 + lb;
    const bool valid = ox < W && oy < H && ob < B && lb < g.tb;
    const int pbase = (lb < g.tb ? (lb * ph + ly) * pw + lx : 0);

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.f;

    for (int c0 = 0; c0 < Cx; c0 += CIT) {
        __syncthreads();
        // stage the input patch (zero padded), channel fastest in the global read
        for (int i = threadIdx.x; i < npatch * CIT; i += 256) {
            const int ci = i % CIT, pp = i / CIT;
            const int px = pp % pw, py = (pp / pw) % ph, pb = pp / (pw * ph);
            const int gx = tx0 + px - PAD, gy = ty0 + py - PAD, gb = b0 + pb;
            float v = 0.f;
            if (gx >= 0 && gx < W && gy >= 0 && gy < H && gb < B && c0 + ci < Cx)
                v = ldf<TI>(x, (((size_t)gb * H + gy) * W + gx) * Cx + c0 + ci);
            sx[ci * npatch + pp] = v;
        }
        // stage the filter slab
        for (int i = threadIdx.x; i < KS * KS * CIT * COT; i += 256) {
            const int j = i % COT, ci = (i / COT) % CIT, tap = i / (COT * CIT);
            const int inc = c0 + ci, outc = co0 + j;
            float v = 0.f;
            if (inc < Cx && outc < Cy) {
                if (!tflip) v = w[((size_t)tap * wCin + inc) * wCout + outc];
                else v = w[((size_t)(KS * KS - 1 - tap) * wCin + outc) * wCout + inc];
            }
            sw[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int poff = pbase + kh * pw + kw;
                const float* wt = sw + (kh * KS + kw) * CIT * COT;
#pragma unroll
                for (int ci = 0; ci < CIT; ++ci) {
                    const float xv = sx[ci * npatch + poff];
#pragma unroll
                    for (int j = 0; j < COT; ++j) acc[j] = fmaf(xv, wt[ci * COT + j], acc[j]);
                }
            }
    }
    float s1[COT], s2[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = co0 + j;
        float v = acc[j];
        if (co < Cy) {
            if (bias) v += bias[co];
            v = act_fwd(v, act);
            if (valid) stf<TO>(y, (((size_t)ob * H + oy) * W + ox) * Cy + co, v);
        }
        const float r = (valid && co < Cy) ? roundf_as<TO>(v) : 0.f;
        s1[j] = r;
        s2[j] = r * r;
    }
    if (stats) {
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            const float a = wave_sum(s1[j]), bq = wave_sum(s2[j]);
            if ((threadIdx.x & 63) == 0 && co0 + j < Cy) {
                atomicAdd(&stats[(co0 + j) * 2], a);
                atomicAdd(&stats[(co0 + j) * 2 + 1], bq);
            }
        }
    }
}

// dw[tap][ci][co] += sum_pixels x[p+tap][ci]*dy[p][co];  thread = (ci in 8) x (co in 32), KS*KS accumulators
#define WG_CI 8
#define WG_CO 32
template <typename TX, typename TD, int KS>
__global__ __launch_bounds__(256) void k_conv_direct_wgrad(const TX* __restrict__ x, const TD* __restrict__ dy,
                                                           float* __restrict__ dw, float* __restrict__ dbias, int B,
                                                           int H, int W, int Cin, int Cout, TileGeo g, int tiles_per_block,
                                                           float* __restrict__ part) {
    constexpr int PAD = KS / 2;
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int pw = tw + KS - 1, ph = th + KS - 1;
    const int npatch = g.tb * ph * pw;
    extern __shared__ float smem[];
    float* sx = smem;                       // [WG_CI][npatch]
    float* sd = smem + WG_CI * npatch;      // [256][WG_CO]
    const int ci_l = threadIdx.x / WG_CO, co_l = threadIdx.x % WG_CO;
    const int ci0 = blockIdx.y * WG_CI, co0 = blockIdx.z * WG_CO;
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    float acc[KS * KS];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) acc[k] = 0.f;
    float accb = 0.f;

    for (int tt = 0; tt < tiles_per_block; ++tt) {
        int t = blockIdx.x * tiles_per_block + tt;
        if (t >= ntiles) break;
        const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
        const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
        const int b0 = t * g.tb;
        __syncthreads();
        for (int i = threadIdx.x; i < npatch * WG_CI; i += 256) {
            const int ci = i % WG_CI, pp = i / WG_CI;
            const int px = pp % pw, py = (pp / pw) % ph, pb = pp / (pw * ph);
            const int gx = tx0 + px - PAD, gy = ty0 + py - PAD, gb = b0 + pb;
            float v = 0.f;
            if (gx >= 0 && gx < W && gy >= 0 && gy < H && gb < B && ci0 + ci < Cin)
                v = ldf<TX>(x, (((size_t)gb * H + gy) * W + gx) * Cin + ci0 + ci);
            sx[ci * npatch + pp] = v;
        }
        for (int i = threadIdx.x; i < 256 * WG_CO; i += 256) {
            const int co = i % WG_CO, m = i / WG_CO;
            const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
            const int ox = tx0 + lx, oy = ty0 + ly, ob = b0 + lb;
            float v = 0.f;
            if (ox < W && oy < H && ob < B && lb < g.tb && co0 + co < Cout)
                v = ldf<TD>(dy, (((size_t)ob * H + oy) * W + ox) * Cout + co0 + co);
            sd[i] = v;
        }
        __syncthreads();
        const int mtile = g.tb << (g.tws + g.ths);
        for (int m = 0; m < mtile; ++m) {
            const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
            const int pbase = (lb * ph + ly) * pw + lx;
            const float d = sd[m * WG_CO + co_l];
            accb += d;
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)
                    acc[kh * KS + kw] = fmaf(sx[ci_l * npatch + pbase + kh * pw + kw], d, acc[kh * KS + kw]);
        }
    }
    const int ci = ci0 + ci_l, co = co0 + co_l;
    if (part) {
        // ordered mode (phx_conv2d_direct_wgrad_ordered): pixel slice blockIdx.x leaves its sums in part[slice][KS*KS*Cin*Cout + Cout];
        // k_direct_wgrad_fold adds the slices in index order
        float* ps = part + (size_t)blockIdx.x * ((size_t)KS * KS * Cin * Cout + Cout);
        if (ci < Cin && co < Cout) {
#pragma unroll
            for (int k = 0; k < KS * KS; ++k) ps[((size_t)k * Cin + ci) * Cout + co] = acc[k];
        }
        if (blockIdx.y == 0 && ci_l == 0 && co < Cout) ps[(size_t)KS * KS * Cin * Cout + co] = accb;
        return;
    }
    if (ci < Cin && co < Cout) {
#pragma unroll
        for (int k = 0; k < KS * KS; ++k) atomicAdd(&dw[((size_t)k * Cin + ci) * Cout + co], acc[k]);
    }
    if (dbias && blockIdx.y == 0 && ci_l == 0 && co < Cout) atomicAdd(&dbias[co], accb);
}
// dw[i] += part[0][i] + part[1][i] + ... in slice order (i < nw); dbias[c] += the same over the slices' bias tails
__global__ void k_direct_wgrad_fold(const float* __restrict__ part, int nslice, size_t nw, int Cout, float* __restrict__ dw,
                                    float* __restrict__ dbias) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = nw + Cout;
    if (i >= stride) return;
    float a = 0.f;
    for (int s = 0; s < nslice; ++s) a += part[(size_t)s * stride + i];
    if (i < nw) dw[i] += a;
    else if (dbias) dbias[i - nw] += a;
}

// ---- small maps (B*H*W <= 4096 pixels) with a handful of channels on one side: the top-level 3x3 mu convolution (192 -> 2
// at 2x2) and its backward.  k_conv_direct gives every output pixel to ONE thread (1728 x COT serial FMAs, six staged
// 32-channel chunks: 60 us for 1.7 MFLOP, on the critical chain of both directions).  Here the reduction is spread out:
// one wave per output pixel, lanes over the wide channel axis, everything a thread needs fetched in one latency round.
template <int KS>
__device__ __forceinline__ size_t tiny_widx(int tap, int ci, int co, int wCin, int wCout, int tflip) {
    // forward: w[tap][ci][co];  data gradient: input channel ci of this convolution is the filter's OUTPUT channel
    return tflip ? ((size_t)(KS * KS - 1 - tap) * wCin + co) * wCout + ci : ((size_t)tap * wCin + ci) * wCout + co;
}
// few output channels (Cy <= 4), many input channels: lanes split the input channels, wave-sum at the end
template <typename TI, typename TO, int KS>
__global__ __launch_bounds__(256) void k_conv_tiny_narrow_out(const TI* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, TO* __restrict__ y, int B,
                                                              int H, int W, int Cx, int Cy, int wCin, int wCout, int act,
                                                              int tflip) {
    constexpr int PAD = KS / 2;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pix >= B * H * W) return;
    const int ox = pix % W, oy = (pix / W) % H, b = pix / (W * H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int iy = oy + kh - PAD, ix = ox + kw - PAD;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const size_t xo = (((size_t)b * H + iy) * W + ix) * Cx;
                for (int c = lane; c < Cx; c += 64) {
                    const float xv = ldf<TI>(x, xo + c);
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < Cy) acc[o] = fmaf(xv, w[tiny_widx<KS>(kh * KS + kw, c, o, wCin, wCout, tflip)], acc[o]);
                }
            }
        }
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = wave_sum(acc[o]);
    if (lane == 0)
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Cy) stf<TO>(y, (size_t)pix * Cy + o, act_fwd(acc[o] + (bias ? bias[o] : 0.f), act));
}
// few input channels (Cx <= 4), many output channels: lanes over the output channels, no reduction across lanes
template <typename TI, typename TO, int KS>
__global__ __launch_bounds__(256) void k_conv_tiny_wide_out(const TI* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, TO* __restrict__ y, int B,
                                                            int H, int W, int Cx, int Cy, int wCin, int wCout, int act,
                                                            int tflip) {
    constexpr int PAD = KS / 2;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pix >= B * H * W) return;
    const int ox = pix % W, oy = (pix / W) % H, b = pix / (W * H);
    for (int o = lane; o < Cy; o += 64) {
        float acc = bias ? bias[o] : 0.f;
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int iy = oy + kh - PAD, ix = ox + kw - PAD;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const size_t xo = (((size_t)b * H + iy) * W + ix) * Cx;
                    for (int c = 0; c < Cx; ++c)
                        acc = fmaf(ldf<TI>(x, xo + c), w[tiny_widx<KS>(kh * KS + kw, c, o, wCin, wCout, tflip)], acc);
                }
            }
        stf<TO>(y, (size_t)pix * Cy + o, act_fwd(acc, act));
    }
}
// filter gradient, Cout <= 4: thread = (tap, input channel), block.y = chunk of 16 output pixels; dw / dbias by atomics
template <typename TX, typename TD, int KS>
__global__ __launch_bounds__(256) void k_conv_tiny_wgrad(const TX* __restrict__ x, const TD* __restrict__ dy,
                                                         float* __restrict__ dw, float* __restrict__ dbias, int B, int H,
                                                         int W, int Cin, int Cout) {
    constexpr int PAD = KS / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // tap * Cin + ci
    const int P = B * H * W, p0 = blockIdx.y * 16;
    if (idx < KS * KS * Cin) {
        const int tap = idx / Cin, ci = idx - tap * Cin, kh = tap / KS, kw = tap % KS;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int p = p0 + q;
            if (p < P) {
                const int ox = p % W, oy = (p / W) % H, b = p / (W * H);
                const int iy = oy + kh - PAD, ix = ox + kw - PAD;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const float xv = ldf<TX>(x, (((size_t)b * H + iy) * W + ix) * Cin + ci);
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < Cout) acc[o] = fmaf(xv, ldf<TD>(dy, (size_t)p * Cout + o), acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Cout) atomicAdd(&dw[(size_t)idx * Cout + o], acc[o]);
    }
    if (dbias && blockIdx.x == 0 && threadIdx.x < Cout) {
        float a = 0.f;
        for (int q = 0; q < 16; ++q)
            if (p0 + q < P) a += ldf<TD>(dy, (size_t)(p0 + q) * Cout + threadIdx.x);
        atomicAdd(&dbias[threadIdx.x], a);
    }
}
static bool tiny_map(int B, int H, int W) {
    return (long)B * H * W <= 4096;
}

extern "C" {

int phx_conv2d_direct(const void* x, int x_dt, const float* w_hwio, const float* bias, void* y, int y_dt, int B, int H,
                      int W, int Cin, int Cout, int ksize, int act, int transpose_flip, float* stats, void* stream) {
    PHX_REQUIRE(ksize == 1 || ksize == 3, PHX_E_SHAPE, "conv2d_direct: ksize must be 1 or 3");
    PHX_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PHX_E_SHAPE, "conv2d_direct: bad shape");
    const int Cx = transpose_flip ? Cout : Cin, Cy = transpose_flip ? Cin : Cout;
    TileGeo g = make_geo(B, H, W);
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int npatch = g.tb * (th + ksize - 1) * (tw + ksize - 1);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
#define CD_LAUNCH(COT, KS, CIT)                                                                                      \
    do {                                                                                                             \
        const size_t sh = (size_t)(CIT * npatch + KS * KS * CIT * COT) * sizeof(float);                              \
        hipLaunchKernelGGL((k_conv_direct<TI, TO, COT, KS, CIT>), dim3(ntiles, (Cy + COT - 1) / COT), dim3(256), sh, \
                           (hipStream_t)stream, (const TI*)x, w_hwio, bias, (TO*)y, stats, B, H, W, Cx, Cy, Cin, Cout, \
                           act, transpose_flip, g);                                                                  \
    } while (0)
    if (tiny_map(B, H, W) && !stats && (Cy <= 4 || Cx <= 4)) {
        const int P = B * H * W;
#define CT_LAUNCH(KERN, KS)                                                                                          \
    hipLaunchKernelGGL((KERN<TI, TO, KS>), dim3((P + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const TI*)x, w_hwio, \
                       bias, (TO*)y, B, H, W, Cx, Cy, Cin, Cout, act, transpose_flip)
        PHX_DT_SWITCH(x_dt, TI, PHX_DT_SWITCH(y_dt, TO, {
            if (Cy <= 4) { if (ksize == 3) CT_LAUNCH(k_conv_tiny_narrow_out, 3); else CT_LAUNCH(k_conv_tiny_narrow_out, 1); }
            else { if (ksize == 3) CT_LAUNCH(k_conv_tiny_wide_out, 3); else CT_LAUNCH(k_conv_tiny_wide_out, 1); }
        }));
#undef CT_LAUNCH
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    PHX_DT_SWITCH(x_dt, TI, PHX_DT_SWITCH(y_dt, TO, {
        if (Cy <= 4 && Cx >= 64) { if (ksize == 3) CD_LAUNCH(4, 3, 32); else CD_LAUNCH(4, 1, 32); }
        else if (Cy <= 4) { if (ksize == 3) CD_LAUNCH(4, 3, 8); else CD_LAUNCH(4, 1, 8); }
        else { if (ksize == 3) CD_LAUNCH(16, 3, 8); else CD_LAUNCH(16, 1, 8); }
    }));
#undef CD_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

// pixel slices of the ordered form: enough that slices x channel blocks fill the chip, at most 64
static int ordered_slices(int B, int H, int W, int Cin, int Cout) {
    TileGeo g = make_geo(B, H, W);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    const int cblocks = ((Cin + WG_CI - 1) / WG_CI) * ((Cout + WG_CO - 1) / WG_CO);
    int want = 256 / cblocks;
    if (want > 64) want = 64;
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    const int tpb = (ntiles + want - 1) / want;
    return (ntiles + tpb - 1) / tpb;
}
size_t phx_conv2d_direct_wgrad_ordered_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const int ns = ordered_slices(B, H, W, Cin, Cout);
    return ns > 1 ? (size_t)ns * ((size_t)ksize * ksize * Cin * Cout + Cout) * sizeof(float) : 0;
}
int phx_conv2d_direct_wgrad_ordered(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwio, float* dbias,
                                    void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int ksize,
                                    void* stream) {
    PHX_REQUIRE(ksize == 1 || ksize == 3, PHX_E_SHAPE, "conv2d_direct_wgrad_ordered: ksize must be 1 or 3");
    PHX_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PHX_E_SHAPE, "conv2d_direct_wgrad_ordered: bad shape");
    const size_t need = phx_conv2d_direct_wgrad_ordered_ws_bytes(B, H, W, Cin, Cout, ksize);
    PHX_REQUIRE(workspace_bytes >= need && (need == 0 || workspace != nullptr), PHX_E_INVAL,
                "conv2d_direct_wgrad_ordered: workspace smaller than phx_conv2d_direct_wgrad_ordered_ws_bytes");
    TileGeo g = make_geo(B, H, W);
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int npatch = g.tb * (th + ksize - 1) * (tw + ksize - 1);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    const int ns = ordered_slices(B, H, W, Cin, Cout);
    const int tpb = (ntiles + ns - 1) / ns;
    // one slice: the block adds straight into dw (a single add per element, as in phx_conv2d_direct_wgrad's deterministic launch)
    float* part = ns > 1 ? (float*)workspace : nullptr;
    const size_t sh = (size_t)(WG_CI * npatch + 256 * WG_CO) * sizeof(float);
    PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(dy_dt, TD, {
        if (ksize == 3)
            hipLaunchKernelGGL((k_conv_direct_wgrad<TX, TD, 3>), dim3(ns, (Cin + WG_CI - 1) / WG_CI, (Cout + WG_CO - 1) / WG_CO),
                               dim3(256), sh, (hipStream_t)stream, (const TX*)x, (const TD*)dy, dw_hwio, dbias, B, H, W,
                               Cin, Cout, g, tpb, part);
        else
            hipLaunchKernelGGL((k_conv_direct_wgrad<TX, TD, 1>), dim3(ns, (Cin + WG_CI - 1) / WG_CI, (Cout + WG_CO - 1) / WG_CO),
                               dim3(256), sh, (hipStream_t)stream, (const TX*)x, (const TD*)dy, dw_hwio, dbias, B, H, W,
                               Cin, Cout, g, tpb, part);
    }));
    PHX_CHECK_LAUNCH();
    if (part) {
        const size_t nw = (size_t)ksize * ksize * Cin * Cout, tot = nw + Cout;
        hipLaunchKernelGGL(k_direct_wgrad_fold, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, ns, nw,
                           Cout, dw_hwio, dbias);
        PHX_CHECK_LAUNCH();
    }
    return PHX_OK;
}

int phx_conv2d_direct_wgrad(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwio, float* dbias, int B,
                            int H, int W, int Cin, int Cout, int ksize, void* stream) {
    PHX_REQUIRE(ksize == 1 || ksize == 3, PHX_E_SHAPE, "conv2d_direct_wgrad: ksize must be 1 or 3");
    if (tiny_map(B, H, W) && Cout <= 4 && !phx_deterministic()) {       // (its pixel groups meet in atomics)
        const int P = B * H * W;
#define CTW_LAUNCH(KS)                                                                                               \
    hipLaunchKernelGGL((k_conv_tiny_wgrad<TX, TD, KS>), dim3((KS * KS * Cin + 255) / 256, (P + 15) / 16), dim3(256), 0, \
                       (hipStream_t)stream, (const TX*)x, (const TD*)dy, dw_hwio, dbias, B, H, W, Cin, Cout)
        PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(dy_dt, TD, { if (ksize == 3) CTW_LAUNCH(3); else CTW_LAUNCH(1); }));
#undef CTW_LAUNCH
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    TileGeo g = make_geo(B, H, W);
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int npatch = g.tb * (th + ksize - 1) * (tw + ksize - 1);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    int tpb = (ntiles + 255) / 256;          // <= 256 blocks along the pixel axis -> bounded atomic traffic
    if (tpb < 1) tpb = 1;
    if (phx_deterministic()) tpb = ntiles;   // one block walks every pixel tile: a single add per filter element
    const int gx = (ntiles + tpb - 1) / tpb;
    const size_t sh = (size_t)(WG_CI * npatch + 256 * WG_CO) * sizeof(float);
    PHX_DT_SWITCH(x_dt, TX, PHX_DT_SWITCH(dy_dt, TD, {
        if (ksize == 3)
            hipLaunchKernelGGL((k_conv_direct_wgrad<TX, TD, 3>), dim3(gx, (Cin + WG_CI - 1) / WG_CI, (Cout + WG_CO - 1) / WG_CO),
                               dim3(256), sh, (hipStream_t)stream, (const TX*)x, (const TD*)dy, dw_hwio, dbias, B, H, W,
                               Cin, Cout, g, tpb, (float*)nullptr);
        else
            hipLaunchKernelGGL((k_conv_direct_wgrad<TX, TD, 1>), dim3(gx, (Cin + WG_CI - 1) / WG_CI, (Cout + WG_CO - 1) / WG_CO),
                               dim3(256), sh, (hipStream_t)stream, (const TX*)x, (const TD*)dy, dw_hwio, dbias, B, H, W,
                               Cin, Cout, g, tpb, (float*)nullptr);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
