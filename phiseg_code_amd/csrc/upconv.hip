// bilinear_upsample2D -> conv2D 3x3 SAME (tfwrapper/layers.py:336-345 into :123; likelihoods.py:200-204 `post_z*_ups` -> `post_z*_ups_c`)
// WITHOUT the up-sampled tensor: the elementwise half of the phase form (tools/polyphase_proto.py).
//
//   conv3x3(up2(x), W)[2i + a, 2j + b] = conv3x3(x, Weff)[i, j, (a, b, :)]      for every hi-res pixel outside the FRAME,
//   Weff[dh, dw, ci, (a, b, co)] = sum_{kh, kw} M[a][dh][kh] M[b][dw][kw] W[kh, kw, ci, co]
//
// (TF 1.12 legacy resize: up[2k] = x[k], up[2k + 1] = (x[k] + x[min(k + 1, n - 1)]) / 2 per axis), i.e. ONE ordinary 3x3 convolution
// at the low resolution with 4 Cout output channels, whose output [B, h, w, (a, b, co)] is the hi-res map with its pixels in another
// order -- a channels-last tensor of 4 B h w pixels for the per-channel normalisation kernels as they are.  The frame (hi rows /
// columns 0, 2n - 2, 2n - 1: conv2D zero-pads the UP-SAMPLED map and the resize clamps at the far edge) comes from the ORDINARY
// kernel run on two small gathered images: per image six rows (zeros, up[0], up[1], up[2n - 3], up[2n - 2], up[2n - 1]) stacked into one
// tall image [1, 6 B, 2w, Cin] -- the zero row is the padding between neighbours; outputs of rows 1, 4, 5 of every six are hi rows
// 0, 2n - 2, 2n - 1 -- and the same for the columns with the filter transposed.  This file holds everything that is not a matrix
// launch: the filter maps (and their transposes for the filter gradient), the frame gather / scatter and their adjoints, the
// depth-to-space permutation.  bf16 activations, 16-byte channel vectors (C % 8 == 0).
#include "phx_common.h"

namespace {

// packed-filter element address (include/phx.h, phx_pack_conv3x3_bf16): tap t, row n, reduction channel k, N rows
__device__ __forceinline__ size_t pk_idx(int t, int n, int k, int N) { return ((((size_t)(k >> 5) * 9 + t) * N + n) << 5) + (k & 31); }

// M[a][dh][kh]: weight of filter tap kh in the low-resolution tap dh of output phase a (dh, kh in 0..2 for -1..1)
__device__ __forceinline__ float phase_m(int a, int dh, int kh) {
    const int i = dh * 3 + kh;
    if (a == 0) return (i == 0 || i == 3 || i == 5 || i == 8) ? 0.5f : (i == 4 ? 1.f : 0.f);       // [[.5 0 0] [.5 1 .5] [0 0 .5]]
    return (i == 3 || i == 8) ? 1.f : ((i == 4 || i == 7) ? 0.5f : 0.f);                             // [[0 0 0] [1 .5 0] [0 .5 1]]
}

// weight of low-res sample i in up-sampled position r of an axis of n samples
__device__ __forceinline__ float up_w(int r, int i, int n) {
    if (!(r & 1)) return i == (r >> 1) ? 1.f : 0.f;
    const int y0 = r >> 1, y1 = min(y0 + 1, n - 1);
    return (i == y0 ? 0.5f : 0.f) + (i == y1 ? 0.5f : 0.f);
}

__device__ __forceinline__ void unpack8(const uint4 q, float (&v)[8]) {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[2 * k] = __uint_as_float(w[k] << 16);
        v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7]));
}

// ---- filters ---------------------------------------------------------------------------------------------------------------
// fp32 W[3][3][Ci][Co] -> packed bf16 Weff (forward: N = 4 Co rows, K = Ci; data gradient: N = Ci rows, K = 4 Co, taps flipped) and
// packed bf16 W^T (W^T[kh][kw] = W[kw][kh]: the column frame's filter; forward and data-gradient forms)
__global__ void k_upconv_pack(const float* __restrict__ w, unsigned short* __restrict__ ef, unsigned short* __restrict__ ed,
                              unsigned short* __restrict__ tf, unsigned short* __restrict__ td, const float* __restrict__ bias,
                              float* __restrict__ bias4, int Ci, int Co) {
    const size_t ne = (size_t)9 * Ci * 4 * Co, nt = (size_t)9 * Ci * Co;
    if (bias4 && blockIdx.x == 0)                     // the convolution bias for the 4 Cout packed columns (group / instance norm keep it)
        for (int i = threadIdx.x; i < 4 * Co; i += blockDim.x) bias4[i] = bias[i % Co];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ne + nt; i += (size_t)gridDim.x * blockDim.x) {
        if (i < ne) {
            const int c4 = (int)(i % (4 * Co)), ci = (int)((i / (4 * Co)) % Ci), t = (int)(i / ((size_t)4 * Co * Ci));
            const int a = c4 / (2 * Co), b = (c4 / Co) & 1, co = c4 % Co, dh = t / 3, dw = t % 3;
            float v = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float m = phase_m(a, dh, kh) * phase_m(b, dw, kw);
                    if (m != 0.f) v += m * w[((size_t)(kh * 3 + kw) * Ci + ci) * Co + co];
                }
            const unsigned short r = f2bf(v);
            ef[pk_idx(t, c4, ci, 4 * Co)] = r;
            if (ed) ed[pk_idx(8 - t, ci, c4, Ci)] = r;
        } else {
            const size_t j = i - ne;
            const int co = (int)(j % Co), ci = (int)((j / Co) % Ci), t = (int)(j / ((size_t)Co * Ci));
            const int kh = t / 3, kw = t % 3;
            const unsigned short r = f2bf(w[((size_t)(kw * 3 + kh) * Ci + ci) * Co + co]);
            tf[pk_idx(t, co, ci, Co)] = r;
            if (td) td[pk_idx(8 - t, ci, co, Ci)] = r;
        }
    }
}

// the transposes: dW[kh][kw][ci][co] += sum_{dh, dw, a, b} M[a][dh][kh] M[b][dw][kw] dWeff[dh][dw][ci][(a, b, co)] + dWt[kw][kh][ci][co]
__global__ void k_upconv_fold(const float* __restrict__ de, const float* __restrict__ dt, float* __restrict__ dw, int Ci, int Co) {
    const size_t n = (size_t)9 * Ci * Co;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co), ci = (int)((i / Co) % Ci), t = (int)(i / ((size_t)Co * Ci));
        const int kh = t / 3, kw = t % 3;
        float acc = dt ? dt[((size_t)(kw * 3 + kh) * Ci + ci) * Co + co] : 0.f;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dwi = 0; dwi < 3; ++dwi)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) {
                    const float m = phase_m(ab >> 1, dh, kh) * phase_m(ab & 1, dwi, kw);
                    if (m != 0.f) acc += m * de[((size_t)(dh * 3 + dwi) * Ci + ci) * 4 * Co + (size_t)ab * Co + co];
                }
        dw[i] += acc;
    }
}

// ---- the frame -------------------------------------------------------------------------------------------------------------
// hi-res row / column of slot s of a frame group (slot 0 is the zero row): s = 1, 2 -> 0, 1;  s = 3, 4, 5 -> 2n - 3, 2n - 2, 2n - 1
__device__ __forceinline__ int frame_pos(int s, int n) { return s <= 2 ? s - 1 : 2 * n - 6 + s; }

// up2(x)[r][c] for one channel vector, the expressions of k_bilinear_up2x_fwd (elementwise.hip)
__device__ __forceinline__ void up_at(const unsigned short* __restrict__ x, size_t img, int r, int c, int h, int w, int C, int cv,
                                      float (&o)[8]) {
    const int y0 = r >> 1, x0 = c >> 1, y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float fy = (r & 1) ? 0.5f : 0.f, fx = (c & 1) ? 0.5f : 0.f;
    float a[8], b[8], cc[8], d[8];
    unpack8(*reinterpret_cast<const uint4*>(x + ((img * h + y0) * w + x0) * C + (size_t)cv * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(x + ((img * h + y0) * w + x1) * C + (size_t)cv * 8), b);
    unpack8(*reinterpret_cast<const uint4*>(x + ((img * h + y1) * w + x0) * C + (size_t)cv * 8), cc);
    unpack8(*reinterpret_cast<const uint4*>(x + ((img * h + y1) * w + x1) * C + (size_t)cv * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float top = a[j] + (b[j] - a[j]) * fx, bot = cc[j] + (d[j] - cc[j]) * fx;
        o[j] = top + (bot - top) * fy;
    }
}

// x [B][h][w][C] -> frows [B][6][2w][C], fcols [B][6][2h][C] (the columns as rows: fcols[b][s][r] = up2(x)[r][column of slot s])
__global__ void k_upconv_frame_gather(const unsigned short* __restrict__ x, unsigned short* __restrict__ frows,
                                      unsigned short* __restrict__ fcols, int B, int h, int w, int C) {
    const int CV = C / 8;
    const size_t nr = (size_t)B * 6 * 2 * w * CV, nc = (size_t)B * 6 * 2 * h * CV;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < nr + nc; it += (size_t)gridDim.x * blockDim.x) {
        const bool rows = it < nr;
        size_t r = rows ? it : it - nr;
        const int len = rows ? 2 * w : 2 * h;
        const int cv = (int)(r % CV); r /= CV;
        const int p = (int)(r % len); r /= len;
        const int s = (int)(r % 6);
        const size_t b = r / 6;
        float o[8];
        if (s == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        } else if (rows) {
            up_at(x, b, frame_pos(s, h), p, h, w, C, cv, o);
        } else {
            up_at(x, b, p, frame_pos(s, w), h, w, C, cv, o);
        }
        *reinterpret_cast<uint4*>((rows ? frows : fcols) + ((b * 6 + s) * len + p) * C + (size_t)cv * 8) = pack8(o);
    }
}

// packed address of hi-res pixel (p, q) of image b: [B][h][w][(a, b, co)]
__device__ __forceinline__ size_t packed_at(size_t b, int p, int q, int h, int w, int Co) {
    return (((b * h + (p >> 1)) * w + (q >> 1)) * 4 + (size_t)((p & 1) * 2 + (q & 1))) * Co;
}

// fr [B][6][2w][Co], fc [B][6][2h][Co] (the frame convolutions' outputs) -> the frame pixels of y [B][h][w][4 Co]
__global__ void k_upconv_frame_scatter(const unsigned short* __restrict__ fr, const unsigned short* __restrict__ fc,
                                       unsigned short* __restrict__ y, int B, int h, int w, int Co) {
    const int CV = Co / 8;
    const size_t nr = (size_t)B * 3 * 2 * w * CV, nc = (size_t)B * 3 * 2 * h * CV;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < nr + nc; it += (size_t)gridDim.x * blockDim.x) {
        const bool rows = it < nr;
        size_t r = rows ? it : it - nr;
        const int len = rows ? 2 * w : 2 * h;
        const int cv = (int)(r % CV); r /= CV;
        const int p = (int)(r % len); r /= len;
        const int s3 = (int)(r % 3);
        const size_t b = r / 3;
        const int s = s3 == 0 ? 1 : s3 + 3;                        // slots 1, 4, 5
        if (rows) {
            const int row = frame_pos(s, h);
            *reinterpret_cast<uint4*>(y + packed_at(b, row, p, h, w, Co) + (size_t)cv * 8) =
                *reinterpret_cast<const uint4*>(fr + ((b * 6 + s) * len + p) * Co + (size_t)cv * 8);
        } else {
            if (p == 0 || p >= 2 * h - 2) continue;                // the corners belong to the row frame
            const int col = frame_pos(s, w);
            *reinterpret_cast<uint4*>(y + packed_at(b, p, col, h, w, Co) + (size_t)cv * 8) =
                *reinterpret_cast<const uint4*>(fc + ((b * 6 + s) * len + p) * Co + (size_t)cv * 8);
        }
    }
}

// the adjoint of the scatter: dy's frame pixels -> dfr / dfc (zeros in the rows that are not outputs), and ZEROED in dy (the phase
// convolution's result was overwritten there, so no gradient flows through it)
__global__ void k_upconv_frame_gather_dy(unsigned short* __restrict__ dy, unsigned short* __restrict__ dfr,
                                         unsigned short* __restrict__ dfc, int B, int h, int w, int Co) {
    const int CV = Co / 8;
    const size_t nr = (size_t)B * 6 * 2 * w * CV, nc = (size_t)B * 6 * 2 * h * CV;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < nr + nc; it += (size_t)gridDim.x * blockDim.x) {
        const bool rows = it < nr;
        size_t r = rows ? it : it - nr;
        const int len = rows ? 2 * w : 2 * h;
        const int cv = (int)(r % CV); r /= CV;
        const int p = (int)(r % len); r /= len;
        const int s = (int)(r % 6);
        const size_t b = r / 6;
        uint4 v = z;
        const bool live = (s == 1 || s >= 4) && (rows || (p != 0 && p < 2 * h - 2));
        if (live) {
            uint4* src = reinterpret_cast<uint4*>(dy + (rows ? packed_at(b, frame_pos(s, h), p, h, w, Co) : packed_at(b, p, frame_pos(s, w), h, w, Co))
                                                  + (size_t)cv * 8);
            v = *src;
            *src = z;
        }
        *reinterpret_cast<uint4*>((rows ? dfr : dfc) + ((b * 6 + s) * len + p) * Co + (size_t)cv * 8) = v;
    }
}

// the adjoint of the gather: dx [B][h][w][C] += (d up2 / d x)^T of the frame rows' / columns' gradients (the border pixels only)
__global__ void k_upconv_frame_scatter_dx(const unsigned short* __restrict__ dfrows, const unsigned short* __restrict__ dfcols,
                                          unsigned short* __restrict__ dx, int B, int h, int w, int C) {
    const int CV = C / 8;
    // border pixels of an image, enumerated: rows 0, 1, h - 2, h - 1 in full, then columns 0, 1, w - 2, w - 1 of the other rows
    const int nrow = min(4, h), ncol = min(4, w), inner = max(h - 4, 0);
    const int per = nrow * w + inner * ncol;
    const size_t n = (size_t)B * per * CV;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < n; it += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        size_t r = it / CV;
        const int q = (int)(r % per);
        const size_t b = r / per;
        int i, j;
        if (q < nrow * w) {
            const int k = q / w;
            j = q - k * w;
            i = (h <= 4) ? k : (k < 2 ? k : h - 4 + k);
        } else {
            const int k = (q - nrow * w) / ncol, c = (q - nrow * w) - k * ncol;
            i = 2 + k;
            j = (w <= 4) ? c : (c < 2 ? c : w - 4 + c);
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int s = 1; s < 6; ++s) {
            const float wy = up_w(frame_pos(s, h), i, h);
            if (wy != 0.f)
                for (int c = max(2 * j - 1, 0); c <= min(2 * j + 1, 2 * w - 1); ++c) {
                    const float ww = wy * up_w(c, j, w);
                    if (ww == 0.f) continue;
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(dfrows + ((b * 6 + s) * 2 * w + c) * C + (size_t)cv * 8), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += ww * v[e];
                }
            const float wx = up_w(frame_pos(s, w), j, w);
            if (wx != 0.f)
                for (int rr = max(2 * i - 1, 0); rr <= min(2 * i + 1, 2 * h - 1); ++rr) {
                    const float ww = wx * up_w(rr, i, h);
                    if (ww == 0.f) continue;
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(dfcols + ((b * 6 + s) * 2 * h + rr) * C + (size_t)cv * 8), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += ww * v[e];
                }
        }
        uint4* dst = reinterpret_cast<uint4*>(dx + ((b * h + i) * w + j) * C + (size_t)cv * 8);
        float o[8];
        unpack8(*dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += acc[e];
        *dst = pack8(o);
    }
}

// ---- the pixel order ---------------------------------------------------------------------------------------------------------
// packed [B][h][w][(a, b, c)] <-> hi-res [B][2h][2w][c]; to_hi: packed -> hi-res, else hi-res -> packed
__global__ void k_depth_space2(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int B, int h, int w, int C,
                               int to_hi) {
    const int CV = C / 8;
    const size_t n = (size_t)B * 4 * h * w * CV;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < n; it += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        size_t r = it / CV;
        const int q = (int)(r % (2 * w)); r /= 2 * w;
        const int p = (int)(r % (2 * h));
        const size_t b = r / (2 * h);
        const size_t hi = it * 8, pk = packed_at(b, p, q, h, w, C) + (size_t)cv * 8;
        if (to_hi) *reinterpret_cast<uint4*>(dst + hi) = *reinterpret_cast<const uint4*>(src + pk);
        else *reinterpret_cast<uint4*>(dst + pk) = *reinterpret_cast<const uint4*>(src + hi);
    }
}

}  // namespace

#define UPCONV_LAUNCH(kern, items, ...)                                                                          \
    do {                                                                                                       \
        hipLaunchKernelGGL(kern, dim3(phx_grid_for((items), 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        PHX_CHECK_LAUNCH();                                                                                    \
        return PHX_OK;                                                                                         \
    } while (0)

extern "C" {

int phx_upconv_supported(int B, int h, int w, int Cin, int Cout) {
    return (B > 0 && h >= 4 && w >= 4 && Cin % 32 == 0 && Cout % 32 == 0 && (size_t)B * 4 * h * w * (Cin > Cout ? Cin : Cout) < 2147483648ull) ? 1 : 0;
}
int phx_upconv_pack(const float* w_hwio, void* weff_fwd, void* weff_dgrad, void* wt_fwd, void* wt_dgrad, const float* bias, float* bias4,
                    int Cin, int Cout, void* stream) {
    PHX_REQUIRE(w_hwio && weff_fwd && wt_fwd && Cin % 32 == 0 && Cout % 32 == 0, PHX_E_INVAL, "upconv_pack: Cin % 32 == 0, Cout % 32 == 0");
    PHX_REQUIRE((bias == nullptr) == (bias4 == nullptr), PHX_E_INVAL, "upconv_pack: bias and bias4 go together");
    UPCONV_LAUNCH(k_upconv_pack, (size_t)9 * Cin * 5 * Cout, w_hwio, (unsigned short*)weff_fwd, (unsigned short*)weff_dgrad,
                  (unsigned short*)wt_fwd, (unsigned short*)wt_dgrad, bias, bias4, Cin, Cout);
}
int phx_upconv_fold_wgrad(const float* dweff, const float* dwt, float* dw_hwio, int Cin, int Cout, void* stream) {
    PHX_REQUIRE(dweff && dw_hwio, PHX_E_INVAL, "upconv_fold_wgrad: null argument");
    UPCONV_LAUNCH(k_upconv_fold, (size_t)9 * Cin * Cout, dweff, dwt, dw_hwio, Cin, Cout);
}
int phx_upconv_frame_gather(const void* x, void* f_rows, void* f_cols, int B, int h, int w, int C, void* stream) {
    PHX_REQUIRE(C % 8 == 0 && h >= 2 && w >= 2, PHX_E_SHAPE, "upconv_frame_gather: C % 8 == 0, h, w >= 2");
    UPCONV_LAUNCH(k_upconv_frame_gather, (size_t)B * 12 * (h + w) * (C / 8), (const unsigned short*)x, (unsigned short*)f_rows,
                  (unsigned short*)f_cols, B, h, w, C);
}
int phx_upconv_frame_scatter(const void* fr, const void* fc, void* y_packed, int B, int h, int w, int Cout, void* stream) {
    PHX_REQUIRE(Cout % 8 == 0 && h >= 2 && w >= 2, PHX_E_SHAPE, "upconv_frame_scatter: Cout % 8 == 0, h, w >= 2");
    UPCONV_LAUNCH(k_upconv_frame_scatter, (size_t)B * 6 * (h + w) * (Cout / 8), (const unsigned short*)fr, (const unsigned short*)fc,
                  (unsigned short*)y_packed, B, h, w, Cout);
}
int phx_upconv_frame_gather_dy(void* dy_packed, void* dfr, void* dfc, int B, int h, int w, int Cout, void* stream) {
    PHX_REQUIRE(Cout % 8 == 0 && h >= 2 && w >= 2, PHX_E_SHAPE, "upconv_frame_gather_dy: Cout % 8 == 0, h, w >= 2");
    UPCONV_LAUNCH(k_upconv_frame_gather_dy, (size_t)B * 12 * (h + w) * (Cout / 8), (unsigned short*)dy_packed, (unsigned short*)dfr,
                  (unsigned short*)dfc, B, h, w, Cout);
}
int phx_upconv_frame_scatter_dx(const void* df_rows, const void* df_cols, void* dx, int B, int h, int w, int C, void* stream) {
    PHX_REQUIRE(C % 8 == 0 && h >= 4 && w >= 4, PHX_E_SHAPE, "upconv_frame_scatter_dx: C % 8 == 0, h, w >= 4");
    UPCONV_LAUNCH(k_upconv_frame_scatter_dx, (size_t)B * (4 * w + (h - 4) * 4) * (C / 8), (const unsigned short*)df_rows,
                  (const unsigned short*)df_cols, (unsigned short*)dx, B, h, w, C);
}
int phx_depth_to_space2(const void* packed, void* hi, int B, int h, int w, int C, void* stream) {
    PHX_REQUIRE(C % 8 == 0, PHX_E_SHAPE, "depth_to_space2: C % 8 == 0");
    UPCONV_LAUNCH(k_depth_space2, (size_t)B * 4 * h * w * (C / 8), (const unsigned short*)packed, (unsigned short*)hi, B, h, w, C, 1);
}
int phx_space_to_depth2(const void* hi, void* packed, int B, int h, int w, int C, void* stream) {
    PHX_REQUIRE(C % 8 == 0, PHX_E_SHAPE, "space_to_depth2: C % 8 == 0");
    UPCONV_LAUNCH(k_depth_space2, (size_t)B * 4 * h * w * (C / 8), (const unsigned short*)hi, (unsigned short*)packed, B, h, w, C, 0);
}

}  // extern "C"
