// HBM-bound kernels of the PHiSeg step: normalisation (batch/group/instance), affine+activation,
// pooling, TF1 bilinear resize, concat/split, casts, posterior input assembly, global pooling.
// All are streaming kernels: 16-byte vector access along the NHWC channel axis when C % 8 == 0.
#include <stdlib.h>
#include <type_traits>

#include "phx_common.h"

// ---- 8-wide vector access ----------------------------------------------------------------------
template <typename T, int V> struct VecIO;
template <> struct VecIO<float, 8> {
    static __device__ __forceinline__ void load(const float* p, size_t i, float o[8]) {
        const float4* q = reinterpret_cast<const float4*>(p + i);
        float4 a = q[0], b = q[1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, size_t i, const float o[8]) {
        float4* q = reinterpret_cast<float4*>(p + i);
        q[0] = make_float4(o[0], o[1], o[2], o[3]);
        q[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
};
template <> struct VecIO<bf16_t, 8> {
    static __device__ __forceinline__ void load(const bf16_t* p, size_t i, float o[8]) {
        uint4 r = *reinterpret_cast<const uint4*>(p + i);
        unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = __uint_as_float(w[k] << 16);
            o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, size_t i, const float o[8]) {
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = f2bf_pk(o[2 * k], o[2 * k + 1]);
        *reinterpret_cast<uint4*>(p + i) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <typename T> struct VecIO<T, 1> {
    static __device__ __forceinline__ void load(const T* p, size_t i, float o[1]) { o[0] = ldf<T>(p, i); }
    static __device__ __forceinline__ void store(T* p, size_t i, const float o[1]) { stf<T>(p, i, o[0]); }
};

#define PHX_VEC_SWITCH(C, V, ...)                         \
    do {                                                  \
        if ((C) % 8 == 0) { constexpr int V = 8; __VA_ARGS__; } \
        else { constexpr int V = 1; __VA_ARGS__; }        \
    } while (0)

// =================================================================================================
// normalisation statistics: sums[ns][c][2] += {sum x, sum x^2} over the P pixels of sample-group ns
template <typename T, int V>
__global__ void k_norm_stats(const T* __restrict__ x, float* __restrict__ sums, float* __restrict__ pivot, int P,
                             int C, int PL, int chunk) {
    const int CV = C / V;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float red[];   // [PL][C][2]
    float s1[V], s2[V], pv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s1[j] = s2[j] = pv[j] = 0.f;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    if (pl < PL) {
        if (pivot) {      // shifted sums: pivot = first pixel of the sample-group -> no catastrophic cancellation
            VecIO<T, V>::load(x, ((size_t)ns * P) * C + (size_t)cv * V, pv);
            if (blockIdx.x == 0 && pl == 0) {
#pragma unroll
                for (int j = 0; j < V; ++j) pivot[(size_t)ns * C + cv * V + j] = pv[j];
            }
        }
        // four pixels per trip, loads first: a thread's trips are serially dependent, so the loads in flight per
        // thread -- not the block count -- set the speed on the mid-size maps
        int p = p0 + pl;
        for (; p + 3 * PL < p1; p += 4 * PL) {
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) VecIO<T, V>::load(x, ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) { const float d = v[u][j] - pv[j]; s1[j] += d; s2[j] += d * d; }
        }
        for (; p < p1; p += PL) {
            float v[V];
            VecIO<T, V>::load(x, ((size_t)ns * P + p) * C + (size_t)cv * V, v);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = v[j] - pv[j]; s1[j] += d; s2[j] += d * d; }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            red[(pl * C + cv * V + j) * 2 + 0] = s1[j];
            red[(pl * C + cv * V + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float a = 0.f;
        for (int q = 0; q < PL; ++q) a += red[q * 2 * C + i];
        atomicAdd(&sums[(size_t)ns * 2 * C + i], a);
    }
}

// =================================================================================================
// Small-map batch norm (P = B*H*W <= 4096 pixels: the H <= 8 levels).  There the three-launch forward (statistics,
// finalisation + apply) and two-launch backward (reduction, apply) are a chain of 5-10 us latency-bound launches for a
// tensor of at most 1.5 MB.  Here ONE launch does the whole layer: a block owns 16 channels (a 32-byte slice of every
// pixel row) for ALL pixels, keeps its slice in registers (NIT x 16 B per thread), and reduces over pixels inside the
// block -- no atomics, no second launch, exact two-pass variance.  1024 threads = 512 pixel lanes x 2 channel vectors.
__device__ __forceinline__ void bf16x8_unpack(const uint4& r, float o[8]) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = __uint_as_float(w[k] << 16);
        o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
}
// sum of s[0..NV) over the 512 pixel lanes that share this thread's channel vector (threadIdx.x & 1); red: [17][2][NV]
template <int NV>
__device__ __forceinline__ void small_block_sum(float s[NV], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 2; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < NV; ++j) s[j] += __shfl_xor(s[j], m);
    __syncthreads();                                   // red may still be read from the previous call
    if (lane < 2)
#pragma unroll
        for (int j = 0; j < NV; ++j) red[(wave * 2 + lane) * NV + j] = s[j];
    __syncthreads();
    float* tot = red + 16 * 2 * NV;                    // [2][NV]
    if (threadIdx.x < 2 * NV) {
        const int vv = threadIdx.x / NV, j = threadIdx.x % NV;
        float a = 0.f;
        for (int w = 0; w < 16; ++w) a += red[(w * 2 + vv) * NV + j];
        tot[vv * NV + j] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) s[j] = tot[(threadIdx.x & 1) * NV + j];
}

// XF32 (round 5): the pre-normalisation tensor x is fp32 -- what the split-K convolution of these small maps leaves (its fp32 slices,
// summed) -- and the statistics and the normalisation run on the UNROUNDED values.  A batch-norm layer at 2 x 2 / 4 x 4 normalises a
// few dozen to a few hundred values per channel; when their spread is small against their mean, the bf16 rounding of x (2^-9 of the
// MEAN) is a large fraction of the spread the normalisation blows up to unit variance: measured as a +40 % bias of the two coarsest
// KL terms after 200 training steps (tools/convergence_study.py).  These tensors are < 1 MB: fp32 is free.
template <bool XF32>
struct BnsVec {                                        // eight channels of one pixel in registers: packed bf16 or fp32
    uint4 r; float f[XF32 ? 8 : 1];
    __device__ __forceinline__ void zero() {
        r = make_uint4(0, 0, 0, 0);
        if constexpr (XF32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
    }
    __device__ __forceinline__ void load(const void* x, size_t i) {
        if constexpr (XF32) {
            typedef __attribute__((ext_vector_type(4))) float f32x4_t;
            const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>((const float*)x + i), a1 = *reinterpret_cast<const f32x4_t*>((const float*)x + i + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = a0[j]; f[4 + j] = a1[j]; }
        } else r = *reinterpret_cast<const uint4*>((const bf16_t*)x + i);
    }
    __device__ __forceinline__ void unpack(float o[8]) const {
        if constexpr (XF32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f[j];
        } else bf16x8_unpack(r, o);
    }
};
template <int NIT, bool XF32>
__global__ __launch_bounds__(1024) void k_bn_small_fwd(const void* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                       float* mean_out, float* rstd_out, float* scale_out,
                                                       float* shift_out, float* moving_mean, float* moving_var,
                                                       float momentum, int P, int C, int act) {
    __shared__ float red[17 * 2 * 8];
    const int v = threadIdx.x & 1, pl = threadIdx.x >> 1;
    const int c0 = blockIdx.x * 16 + v * 8;
    BnsVec<XF32> r[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = pl + it * 512;
        r[it].zero();
        if (p < P) r[it].load(x, (size_t)p * C + c0);
    }
    float s[8], mu[8], gm[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; gm[j] = gamma[c0 + j]; be[j] = beta[c0 + j]; }   // (loaded ahead of the reductions)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {                  // (pixels past P hold zeros)
        float f[8];
        r[it].unpack(f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
    small_block_sum<8>(s, red);
    const float invP = 1.f / (float)P;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = s[j] * invP; s[j] = 0.f; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (pl + it * 512 < P) {
            float f[8];
            r[it].unpack(f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[j] - mu[j]; s[j] = fmaf(d, d, s[j]); }
        }
    }
    small_block_sum<8>(s, red);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        const float var = s[j] * invP, rs = rsqrtf(var + eps);
        sc[j] = gm[j] * rs;
        sh[j] = be[j] - mu[j] * sc[j];
        if (pl == 0) {
            mean_out[c] = mu[j];
            rstd_out[c] = rs;
            scale_out[c] = sc[j];
            shift_out[c] = sh[j];
            if (momentum > 0.f && moving_mean) {          // TF1 fused-batch-norm moving update (unbiased variance)
                const float m = (float)P;
                moving_mean[c] -= (moving_mean[c] - mu[j]) * momentum;
                moving_var[c] -= (moving_var[c] - var * (m / fmaxf(m - 1.f, 1.f))) * momentum;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = pl + it * 512;
        if (p < P) {
            float f[8];
            r[it].unpack(f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = act_fwd(fmaf(f[j], sc[j], sh[j]), act);
            VecIO<bf16_t, 8>::store(y, (size_t)p * C + c0, f);
        }
    }
}

template <int NIT, bool XF32>
__global__ __launch_bounds__(1024) void k_bn_small_bwd(const bf16_t* __restrict__ dA, const void* __restrict__ x,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, bf16_t* __restrict__ dx,
                                                       float* dgamma, float* dbeta, int P, int C, int act) {
    __shared__ float red[17 * 2 * 16];
    const int v = threadIdx.x & 1, pl = threadIdx.x >> 1;
    const int c0 = blockIdx.x * 16 + v * 8;
    // the slice stays in registers between the two passes when that is <= 4 x 16 B per thread; larger slices are read
    // again (from L2) -- 2 x 8 x 16 B per thread on top of the unpacked working set does not fit 128 registers
    constexpr bool KEEP = NIT <= 2;
    constexpr int NB = KEEP ? NIT : 2;                 // register buffers (reads go two pixels at a time when not kept)
    BnsVec<XF32> rx[NB];
    uint4 rd[NB];
    auto fetch = [&](int it) {
        const int p = pl + it * 512;
        rx[it % NB].zero();
        rd[it % NB] = make_uint4(0, 0, 0, 0);
        if (p < P) {
            rx[it % NB].load(x, (size_t)p * C + c0);
            rd[it % NB] = *reinterpret_cast<const uint4*>(dA + (size_t)p * C + c0);
        }
    };
    if constexpr (KEEP) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) fetch(it);
    }
    float sc[8], sh[8], mu[8], rs[8], gmv[8], s[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c]; gmv[j] = gamma[c];
        s[j] = s[8 + j] = 0.f;
    }
    auto accumulate = [&](int it) {                     // (pixels past P: dA = 0 -> g = 0)
        float xf[8], df[8];
        rx[it % NB].unpack(xf);
        bf16x8_unpack(rd[it % NB], df);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = df[j] * act_grad_pre(fmaf(xf[j], sc[j], sh[j]), act);
            s[j] += g;
            s[8 + j] = fmaf(g * (xf[j] - mu[j]), rs[j], s[8 + j]);
        }
    };
    if constexpr (KEEP) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) accumulate(it);
    } else {
#pragma unroll 1
        for (int it = 0; it < NIT; it += 2) { fetch(it); fetch(it + 1); accumulate(it); accumulate(it + 1); }
    }
    small_block_sum<16>(s, red);
    const float inv_m = 1.f / (float)P;
    float ca[8], cb[8], cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        const float gm = gmv[j];
        ca[j] = rs[j] * gm;
        cc[j] = -rs[j] * rs[j] * gm * s[8 + j] * inv_m;
        cb[j] = -rs[j] * gm * s[j] * inv_m - cc[j] * mu[j];
        if (pl == 0) {                                  // (accumulate: a variable may be used by several layers)
            atomicAdd(&dbeta[c], s[j]);
            atomicAdd(&dgamma[c], s[8 + j]);
        }
    }
    auto apply = [&](int it) {
        const int p = pl + it * 512;
        if (p < P) {
            float xf[8], df[8], o[8];
            rx[it % NB].unpack(xf);
            bf16x8_unpack(rd[it % NB], df);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gq = df[j] * act_grad_pre(fmaf(xf[j], sc[j], sh[j]), act);
                o[j] = fmaf(ca[j], gq, fmaf(cc[j], xf[j], cb[j]));
            }
            VecIO<bf16_t, 8>::store(dx, (size_t)p * C + c0, o);
        }
    };
    if constexpr (KEEP) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) apply(it);
    } else {
#pragma unroll 1
        for (int it = 0; it < NIT; it += 2) { fetch(it); fetch(it + 1); apply(it); apply(it + 1); }
    }
}

// ---- the WIDE form of the one-launch batch norm (round 5): P <= 1024 pixels (the 2 x 2 / 4 x 4 levels at batch 64), fp32
// pre-normalisation tensor given as the nz split-K slices the convolution left -- this launch is also the split-K finishing pass.
// A block of 256 threads owns FOUR channels of all pixels (one float4 per pixel and slice), so a 192-channel layer is 48 blocks on
// 48 CUs instead of 12: the layer is a few hundred KB to a few MB of slices, and what a launch of this size costs is the number of
// loads a CU has in flight, not arithmetic (k_splitk_finish + k_bn_small_fwd: 4.8 + 10.6 us and a launch boundary; this: one launch).
// The slices are added in slice order, as k_splitk_finish adds them: xsum (written for the backward pass when nz > 1) is bit-identical.
__device__ __forceinline__ f32x4 wide_block_sum(f32x4 s, float* red /* [4 waves][4] */) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], m);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) *reinterpret_cast<f32x4*>(red + wave * 4) = s;
    __syncthreads();
    f32x4 t = *reinterpret_cast<const f32x4*>(red);
#pragma unroll
    for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + w * 4);
    return t;
}
template <int NPT>
__global__ __launch_bounds__(256) void k_bn_wide_fwd(const float* __restrict__ xs, int nz, size_t zstride, float* __restrict__ xsum,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     bf16_t* __restrict__ y, float* mean_out, float* rstd_out, float* scale_out,
                                                     float* shift_out, float* moving_mean, float* moving_var, float momentum,
                                                     int P, int C, int act) {
    __shared__ __attribute__((aligned(16))) float red[2][16];
    const int c0 = blockIdx.x * 4;
    f32x4 v[NPT];
#pragma unroll
    for (int it = 0; it < NPT; ++it) {
        const int p = threadIdx.x + it * 256;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p < P) v[it] = *reinterpret_cast<const f32x4*>(xs + (size_t)p * C + c0);
    }
    for (int z0 = 1; z0 < nz; z0 += 4) {                 // four slices of every pixel in flight; added in slice order
        f32x4 t[NPT][4];
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int p = threadIdx.x + it * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[it][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p < P && z0 + k < nz) t[it][k] = *reinterpret_cast<const f32x4*>(xs + (size_t)(z0 + k) * zstride + (size_t)p * C + c0);
            }
        }
#pragma unroll
        for (int it = 0; it < NPT; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (z0 + k < nz) v[it] += t[it][k];
    }
    if (nz > 1 && xsum != nullptr) {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int p = threadIdx.x + it * 256;
            if (p < P) *reinterpret_cast<f32x4*>(xsum + (size_t)p * C + c0) = v[it];
        }
    }
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 s = v[0];
#pragma unroll
    for (int it = 1; it < NPT; ++it) s += v[it];                         // (pixels past P hold zeros)
    const float invP = 1.f / (float)P;
    const f32x4 mu = wide_block_sum(s, red[0]) * invP;
    s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPT; ++it)
        if ((int)threadIdx.x + it * 256 < P) { const f32x4 d = v[it] - mu; s += d * d; }
    const f32x4 var = wide_block_sum(s, red[1]) * invP;
    f32x4 sc, sh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float rs = rsqrtf(var[j] + eps);
        sc[j] = gm[j] * rs;
        sh[j] = be[j] - mu[j] * sc[j];
        if (threadIdx.x == 0) {
            const int c = c0 + j;
            mean_out[c] = mu[j];
            rstd_out[c] = rs;
            scale_out[c] = sc[j];
            shift_out[c] = sh[j];
            if (momentum > 0.f && moving_mean) {          // TF1 fused-batch-norm moving update (unbiased variance)
                const float m = (float)P;
                moving_mean[c] -= (moving_mean[c] - mu[j]) * momentum;
                moving_var[c] -= (moving_var[c] - var[j] * (m / fmaxf(m - 1.f, 1.f))) * momentum;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NPT; ++it) {
        const int p = threadIdx.x + it * 256;
        if (p < P) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_fwd(fmaf(v[it][j], sc[j], sh[j]), act);
            *reinterpret_cast<uint2*>(y + (size_t)p * C + c0) = make_uint2(f2bf_pk(o[0], o[1]), f2bf_pk(o[2], o[3]));
        }
    }
}
// ... and its backward: dA (bf16, or the nzd fp32 split-K slices the consumer's data gradient left -- das != NULL), the fp32 x
template <int NPT>
__global__ __launch_bounds__(256) void k_bn_wide_bwd(const bf16_t* __restrict__ dA, const float* __restrict__ das, int nzd, size_t zstride,
                                                     const float* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     bf16_t* __restrict__ dx, float* dgamma, float* dbeta, int P, int C, int act) {
    __shared__ __attribute__((aligned(16))) float red[2][16];
    const int c0 = blockIdx.x * 4;
    f32x4 xv[NPT], g[NPT];
#pragma unroll
    for (int it = 0; it < NPT; ++it) {
        const int p = threadIdx.x + it * 256;
        xv[it] = g[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p < P) {
            xv[it] = *reinterpret_cast<const f32x4*>(x + (size_t)p * C + c0);
            if (das == nullptr) {
                const uint2 r = *reinterpret_cast<const uint2*>(dA + (size_t)p * C + c0);
                g[it] = f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
            } else g[it] = *reinterpret_cast<const f32x4*>(das + (size_t)p * C + c0);
        }
    }
    if (das != nullptr) {
        for (int z0 = 1; z0 < nzd; z0 += 4) {
            f32x4 t[NPT][4];
#pragma unroll
            for (int it = 0; it < NPT; ++it) {
                const int p = threadIdx.x + it * 256;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[it][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p < P && z0 + k < nzd) t[it][k] = *reinterpret_cast<const f32x4*>(das + (size_t)(z0 + k) * zstride + (size_t)p * C + c0);
                }
            }
#pragma unroll
            for (int it = 0; it < NPT; ++it)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (z0 + k < nzd) g[it] += t[it][k];
        }
        // (the bf16 tensor the finishing pass would have written is what every other consumer of this gradient sees: round the
        // same way, so that the two forms of this launch agree bit for bit)
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const unsigned w0 = f2bf_pk(g[it][0], g[it][1]), w1 = f2bf_pk(g[it][2], g[it][3]);
            g[it] = f32x4{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
        }
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0), sh = *reinterpret_cast<const f32x4*>(shift + c0);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), rs = *reinterpret_cast<const f32x4*>(rstd + c0);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c0);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPT; ++it) {                  // (pixels past P: dA = 0 -> g = 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[it][j] *= act_grad_pre(fmaf(xv[it][j], sc[j], sh[j]), act);
            s1[j] += g[it][j];
            s2[j] = fmaf(g[it][j] * (xv[it][j] - mu[j]), rs[j], s2[j]);
        }
    }
    s1 = wide_block_sum(s1, red[0]);
    s2 = wide_block_sum(s2, red[1]);
    const float inv_m = 1.f / (float)P;
    f32x4 ca, cb, cc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ca[j] = rs[j] * gm[j];
        cc[j] = -rs[j] * rs[j] * gm[j] * s2[j] * inv_m;
        cb[j] = -rs[j] * gm[j] * s1[j] * inv_m - cc[j] * mu[j];
        if (threadIdx.x == 0) {                         // (accumulate: a variable may be used by several layers)
            atomicAdd(&dbeta[c0 + j], s1[j]);
            atomicAdd(&dgamma[c0 + j], s2[j]);
        }
    }
#pragma unroll
    for (int it = 0; it < NPT; ++it) {
        const int p = threadIdx.x + it * 256;
        if (p < P) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(ca[j], g[it][j], fmaf(cc[j], xv[it][j], cb[j]));
            *reinterpret_cast<uint2*>(dx + (size_t)p * C + c0) = make_uint2(f2bf_pk(o[0], o[1]), f2bf_pk(o[2], o[3]));
        }
    }
}

// =================================================================================================
// Group / instance norm on small maps (a sample has P = H*W <= 256 pixels: the H <= 16 levels), bf16 NHWC: the whole layer
// in one launch, like k_bn_small_* -- but the statistics are per SAMPLE, so nothing is reduced across waves: a wave owns
// (sample, 16-channel slice) pairs, lane = (sample slot, pixel lane, channel vector); LPS pixel lanes per sample (4, 16 or
// 32 -> 8, 2 or 1 samples per wave pass), NIT pixels per lane, reductions by wave shuffles only, no barrier in the forward
// pass.  MODE 0: one statistic per channel (instance norm); MODE 1: one per 16-channel group = the whole slice (group_norm2D's
// default groups for C >= 32).  A block (4 waves) walks a run of samples; the backward pass keeps dgamma / dbeta / dbias of
// its run in registers and adds them once per block.
template <int LPS> struct NormWaveMap {
    static constexpr int SPW = 32 / LPS;               // samples per wave pass
    int v, q, ss;
    __device__ __forceinline__ NormWaveMap() {
        const int l = threadIdx.x & 63;
        v = l & 1;
        q = (l >> 1) & (LPS - 1);
        ss = l / (2 * LPS);
    }
};
template <int LPS, int NV>
__device__ __forceinline__ void wave_pixel_sum(float s[NV]) {     // over the LPS pixel lanes of a sample slot (lane bits 1..)
#pragma unroll
    for (int m = 2; m <= LPS; m <<= 1)
#pragma unroll
        for (int j = 0; j < NV; ++j) s[j] += __shfl_xor(s[j], m);
}
__device__ __forceinline__ float slice16_total(const float t[8]) {   // sum over the 16 channels of the slice (both channel vectors)
    float g = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    return g + __shfl_xor(g, 1);
}

template <int LPS, int NIT, int MODE, bool SLICES>
__global__ __launch_bounds__(256) void k_norm_wave_fwd(bf16_t* __restrict__ x, const float* __restrict__ ws, int nz,
                                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      float* __restrict__ scale_out, float* __restrict__ shift_out, int NS,
                                                      int P, int C, int passes, int act) {
    const NormWaveMap<LPS> m;
    constexpr int SPW = NormWaveMap<LPS>::SPW;
    const int wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 16 + m.v * 8;
    const int G = gridDim.x;
    float gm[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gm[j] = gamma[c0 + j]; be[j] = beta[c0 + j]; }
    const float inv_m = 1.f / ((float)P * (MODE == 1 ? 16.f : 1.f));
    const float lo = act == PHX_ACT_RELU ? 0.f : -INFINITY;     // (uniform activation: see k_norm_wave_bwd)
    auto run = [&](auto softc) {
    constexpr bool SOFT = decltype(softc)::value;
    for (int i = 0; i < passes; ++i) {
        const int ns = ((blockIdx.y * passes + i) * 4 + wave) * SPW + m.ss;
        const bool live = ns < NS;
        const size_t base = (size_t)ns * P * C + c0;
        uint4 r[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p = m.q + it * LPS;
            r[it] = make_uint4(0, 0, 0, 0);
            if (live && p < P) {
                if constexpr (SLICES) {
                    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
                    const size_t zs = (size_t)NS * P * C;
                    const float* qp = ws + base + (size_t)p * C;
                    f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(qp), a1 = *reinterpret_cast<const f32x4_t*>(qp + 4);
                    for (int z = 1; z < nz; ++z) {
                        a0 += *reinterpret_cast<const f32x4_t*>(qp + (size_t)z * zs);
                        a1 += *reinterpret_cast<const f32x4_t*>(qp + (size_t)z * zs + 4);
                    }
                    if (bias) {
                        a0 += *reinterpret_cast<const f32x4_t*>(bias + c0);
                        a1 += *reinterpret_cast<const f32x4_t*>(bias + c0 + 4);
                    }
                    r[it] = make_uint4(f2bf_pk(a0[0], a0[1]), f2bf_pk(a0[2], a0[3]), f2bf_pk(a1[0], a1[1]), f2bf_pk(a1[2], a1[3]));
                    *reinterpret_cast<uint4*>(x + base + (size_t)p * C) = r[it];
                } else {
                    r[it] = *reinterpret_cast<const uint4*>(x + base + (size_t)p * C);
                }
            }
        }
        float s[8], mu[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {                  // (pixels past P hold zeros)
            float f[8];
            bf16x8_unpack(r[it], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
        wave_pixel_sum<LPS, 8>(s);
        if constexpr (MODE == 1) {
            const float g = slice16_total(s);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = g;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu[j] = s[j] * inv_m; s[j] = 0.f; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (m.q + it * LPS < P) {
                float f[8];
                bf16x8_unpack(r[it], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = f[j] - mu[j]; s[j] = fmaf(d, d, s[j]); }
            }
        }
        wave_pixel_sum<LPS, 8>(s);
        if constexpr (MODE == 1) {
            const float g = slice16_total(s);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = g;
        }
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            const float rs = rsqrtf(s[j] * inv_m + eps);
            sc[j] = gm[j] * rs;
            sh[j] = be[j] - mu[j] * sc[j];
            if (live && m.q == 0) {
                if constexpr (MODE == 1) {
                    if (m.v == 0 && j == 0) {
                        mean_out[(size_t)ns * G + blockIdx.x] = mu[j];
                        rstd_out[(size_t)ns * G + blockIdx.x] = rs;
                    }
                } else {
                    mean_out[(size_t)ns * C + c] = mu[j];
                    rstd_out[(size_t)ns * C + c] = rs;
                }
                scale_out[(size_t)ns * C + c] = sc[j];
                shift_out[(size_t)ns * C + c] = sh[j];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p = m.q + it * LPS;
            if (live && p < P) {
                float f[8];
                bf16x8_unpack(r[it], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float v = fmaf(f[j], sc[j], sh[j]); f[j] = SOFT ? act_fwd(v, PHX_ACT_SOFTPLUS) : fmaxf(v, lo); }
                VecIO<bf16_t, 8>::store(y, base + (size_t)p * C, f);
            }
        }
    }
    };
    if (act == PHX_ACT_SOFTPLUS) run(std::true_type()); else run(std::false_type());
}

// dbias: the gradient of a convolution bias in front of the normalisation, sum over samples and pixels of dx, in closed form
// per sample (a * sum g + c * sum (x - mean) - P * rstd * S0 / m; identically zero in MODE 0, not computed there)
template <int LPS, int NIT, int MODE>
__global__ __launch_bounds__(256) void k_norm_wave_bwd(const bf16_t* __restrict__ dA, const bf16_t* __restrict__ x,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, bf16_t* __restrict__ dx,
                                                      float* dgamma, float* dbeta, float* dbias, int NS, int P, int C,
                                                      int passes, int act) {
    constexpr int NA = MODE == 1 ? 24 : 16;
    const NormWaveMap<LPS> m;
    constexpr int SPW = NormWaveMap<LPS>::SPW;
    __shared__ float red[4 * SPW * 2 * NA];
    const int wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 16 + m.v * 8;
    const int G = gridDim.x;
    float gmv[8], acc[NA];
#pragma unroll
    for (int j = 0; j < 8; ++j) gmv[j] = gamma[c0 + j];
#pragma unroll
    for (int j = 0; j < NA; ++j) acc[j] = 0.f;
    const float inv_m = 1.f / ((float)P * (MODE == 1 ? 16.f : 1.f));
    // The activation is uniform per launch: ReLU / identity through a threshold, softplus in its own copy of the loop.  With
    // act_grad_pre(pre, act) per element this kernel carried two scalar branches per element (585 in the 256-pixel instantiation,
    // 8 777 instructions; 90 and 5 143 now).  NOT done in the streaming kernels below: measured slower there.
    const float thr = act == PHX_ACT_RELU ? 0.f : -INFINITY;
    auto run = [&](auto softc) {
    constexpr bool SOFT = decltype(softc)::value;
    for (int i = 0; i < passes; ++i) {
        const int ns = ((blockIdx.y * passes + i) * 4 + wave) * SPW + m.ss;
        const bool live = ns < NS;
        const size_t base = (size_t)ns * P * C + c0;
        uint4 rx[NIT], rd[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p = m.q + it * LPS;
            rx[it] = rd[it] = make_uint4(0, 0, 0, 0);
            if (live && p < P) {
                rx[it] = *reinterpret_cast<const uint4*>(x + base + (size_t)p * C);
                rd[it] = *reinterpret_cast<const uint4*>(dA + base + (size_t)p * C);
            }
        }
        float sc[8], sh[8], mu[8], rs[8], s[NA];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            const size_t si = MODE == 1 ? (size_t)ns * G + blockIdx.x : (size_t)ns * C + c;
            sc[j] = live ? scale[(size_t)ns * C + c] : 0.f;
            sh[j] = live ? shift[(size_t)ns * C + c] : 0.f;
            mu[j] = live ? mean[si] : 0.f;
            rs[j] = live ? rstd[si] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) s[j] = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {              // (pixels past P: dA = 0 -> g = 0)
            float xf[8], df[8];
            bf16x8_unpack(rx[it], xf);
            bf16x8_unpack(rd[it], df);
            const bool in = live && m.q + it * LPS < P;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pre = fmaf(xf[j], sc[j], sh[j]);
                const float g = SOFT ? df[j] * act_grad_pre(pre, PHX_ACT_SOFTPLUS) : (pre > thr ? df[j] : 0.f);
                s[j] += g;
                s[8 + j] = fmaf(g * (xf[j] - mu[j]), rs[j], s[8 + j]);
                if constexpr (MODE == 1) s[16 + j] += in ? xf[j] - mu[j] : 0.f;
            }
        }
        wave_pixel_sum<LPS, NA>(s);
        float S0[8], S1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { S0[j] = gmv[j] * s[j]; S1[j] = gmv[j] * s[8 + j]; }
        if constexpr (MODE == 1) {                      // S = sum over the group's 16 channels of gamma_c * {sum g, sum g xhat}
            const float a = slice16_total(S0), bq = slice16_total(S1);
#pragma unroll
            for (int j = 0; j < 8; ++j) { S0[j] = a; S1[j] = bq; }
        }
        float ca[8], cb[8], cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ca[j] = rs[j] * gmv[j];
            cc[j] = -rs[j] * rs[j] * S1[j] * inv_m;
            cb[j] = -rs[j] * S0[j] * inv_m - cc[j] * mu[j];
            acc[j] += s[j];
            acc[8 + j] += s[8 + j];
            if constexpr (MODE == 1) acc[16 + j] += fmaf(ca[j], s[j], fmaf(cc[j], s[16 + j], -(float)P * rs[j] * S0[j] * inv_m));
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p = m.q + it * LPS;
            if (live && p < P) {
                float xf[8], df[8], o[8];
                bf16x8_unpack(rx[it], xf);
                bf16x8_unpack(rd[it], df);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pre = fmaf(xf[j], sc[j], sh[j]);
                    const float gq = SOFT ? df[j] * act_grad_pre(pre, PHX_ACT_SOFTPLUS) : (pre > thr ? df[j] : 0.f);
                    o[j] = fmaf(ca[j], gq, fmaf(cc[j], xf[j], cb[j]));
                }
                VecIO<bf16_t, 8>::store(dx, base + (size_t)p * C, o);
            }
        }
    }
    };
    if (act == PHX_ACT_SOFTPLUS) run(std::true_type()); else run(std::false_type());
    // the block's run of samples: one add per channel (every pixel lane of a slot holds the slot's totals; lane q == 0 speaks)
    if (m.q == 0)
#pragma unroll
        for (int j = 0; j < NA; ++j) red[((wave * SPW + m.ss) * 2 + m.v) * NA + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 2 * NA) {
        const int vv = threadIdx.x / NA, j = threadIdx.x % NA;
        float a = 0.f;
        for (int w = 0; w < 4 * SPW; ++w) a += red[(w * 2 + vv) * NA + j];
        const int c = blockIdx.x * 16 + vv * 8 + (j & 7);
        if (j < 8) atomicAdd(&dbeta[c], a);
        else if (j < 16) atomicAdd(&dgamma[c], a);
        else if (dbias) atomicAdd(&dbias[c], a);
    }
}

static int norm_geometry(int P, int C, int V, int* PL, int* threads, int* chunk, int* nchunks, int NS, int nrep = 1) {
    int CV = C / V;
    if (CV > 256) return -1;
    *PL = 256 / CV;                              // (512 / 1024-thread blocks measured slower: the LDS reduction over PL lanes grows)
    *threads = CV * (*PL);
    int rows = (P + *PL - 1) / (*PL);
    // Every block ends in 2C same-address fp32 atomics (~45 ns each, serialised per address): the block count is a trade between
    // streaming parallelism and that tail.  The backward reduction spreads its blocks over nrep accumulator replicas and takes
    // more, shorter blocks (16 pixels per thread, 256..512 blocks); the forward statistics pass has one accumulator set -- 256
    // blocks cost it 11 us of atomics on the H <= 16 levels -- and keeps 8 pixels per thread with a floor of 128 blocks.
    const int ppt_r = 16, fl_r = 256, capv = 512, ppt_s = 8, fl_s = 128;        // (measured: tools/bench_norm.py)
    const int ppt_n = nrep > 1 ? ppt_r : ppt_s, fl = nrep > 1 ? fl_r : fl_s;
    int want = (rows + ppt_n - 1) / ppt_n;
    // (the floor counts blocks of the whole launch: with per-sample statistics, NS > 1, every sample gets its share -- a
    // floor per SAMPLE cut the 128 x 128 group-norm layers into thousands of blocks of two pixels per thread: 62 us vs 19)
    const int fl_ns = (fl + (NS > 0 ? NS : 1) - 1) / (NS > 0 ? NS : 1);
    int floor_blocks = rows < fl_ns ? rows : fl_ns;
    if (want < floor_blocks) want = floor_blocks;
    int cap = (nrep > 1 ? capv : 2048) / (NS > 0 ? NS : 1);
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    *chunk = (P + want - 1) / want;
    *nchunks = (P + *chunk - 1) / (*chunk);
    return 0;
}
static int stream_geometry(int P, int C, int V, int* PL, int* threads, int* chunk, int* nchunks, int NS) {
    int CV = C / V;
    if (CV > 256) return -1;
    const int nthr = 256;                        // threads per block
    *PL = nthr / CV;
    if (*PL < 1) *PL = 1;
    *threads = CV * (*PL);
    // 8 pixels per thread on big maps (two trips of four); on small maps fewer, so that ~1024 blocks exist
    int rows = (P + *PL - 1) / (*PL);
    const int ppt_s = 8;                         // pixels per thread (8 / floor 1024 since the prologues are LDS-shared; 32 / 256 before)
    int want = (rows + ppt_s - 1) / ppt_s;
    const int fl = 1024;                         // (measured: tools/bench_norm.py)
    // (the floor counts blocks of the whole launch: with per-sample statistics, NS > 1, every sample gets its share -- a
    // floor per SAMPLE cut the 128 x 128 group-norm layers into thousands of blocks of two pixels per thread: 62 us vs 19)
    const int fl_ns = (fl + (NS > 0 ? NS : 1) - 1) / (NS > 0 ? NS : 1);
    int floor_blocks = rows < fl_ns ? rows : fl_ns;
    if (want < floor_blocks) want = floor_blocks;
    int cap = 8192 / (NS > 0 ? NS : 1);
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    *chunk = (P + want - 1) / want;
    *nchunks = (P + *chunk - 1) / (*chunk);
    return 0;
}

// partial[T][2][C] -> sums[c][2]
__global__ void k_reduce_partials(const float* __restrict__ partial, int T, int C, float* __restrict__ sums) {
    const int i = blockIdx.x;                 // 0 .. 2C-1 : (which, c)
    const int which = i / C, c = i % C;
    // (four independent chains: the plain loop compiled to load / wait / add per tile, 16 serial round trips at 4 096 tiles)
    const float* p = partial + (size_t)which * C + c;
    const int st = blockDim.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int t = threadIdx.x;
    for (; t + 3 * st < T; t += 4 * st) {
        a0 += p[(size_t)t * 2 * C];
        a1 += p[(size_t)(t + st) * 2 * C];
        a2 += p[(size_t)(t + 2 * st) * 2 * C];
        a3 += p[(size_t)(t + 3 * st) * 2 * C];
    }
    for (; t < T; t += st) a0 += p[(size_t)t * 2 * C];
    float a = (a0 + a1) + (a2 + a3);
    __shared__ float sh[4];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) sums[c * 2 + which] = sh[0] + sh[1] + sh[2] + sh[3];
}

// per-sample form (group / instance norm): partial[ns * T + t][2][C] -> sums[ns][c][2], T pixel tiles per sample; a thread owns one
// (which, c) entry of one sample and walks its T tiles (coalesced across the block)
__global__ void k_reduce_partials_ns(const float* __restrict__ partial, int T, int C, float* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, ns = blockIdx.y;
    if (i >= 2 * C) return;
    const float* p = partial + (size_t)ns * T * 2 * C + i;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;             // four independent chains: the T loads of a thread overlap
    int t = 0;
    for (; t + 3 < T; t += 4) {
        a0 += p[(size_t)t * 2 * C];
        a1 += p[(size_t)(t + 1) * 2 * C];
        a2 += p[(size_t)(t + 2) * 2 * C];
        a3 += p[(size_t)(t + 3) * 2 * C];
    }
    for (; t < T; ++t) a0 += p[(size_t)t * 2 * C];
    const int which = i / C, c = i % C;
    sums[((size_t)ns * C + c) * 2 + which] = (a0 + a1) + (a2 + a3);
}

// Per-channel batch / instance-norm coefficients from the accumulated sums -- ONE definition with floating-point contraction off, used
// by the fused apply kernels AND by k_norm_finalize, so that the stand-alone finalisation and the fused one agree bit for bit (round 5:
// with the compiler free to contract `s2 * invP - d1 * d1` differently in the two kernels the variances differed in the last place --
// a bf16 flip per few thousand activations, and two Adam steps later 30 % of the ELBO of a deliberately ill-conditioned test network).
__device__ __forceinline__ void chan_coeffs(float s1, float s2, float pv, float invP, float eps, float* mu, float* var, float* rs) {
#pragma clang fp contract(off)
    const float d1 = s1 * invP;
    const float m2 = s2 * invP;
    const float dd = d1 * d1;
    *mu = pv + d1;
    *var = fmaxf(m2 - dd, 0.f);
    *rs = rsqrtf(*var + eps);
}
__device__ __forceinline__ void moving_update(float* mm, float* mv, float mu, float var, float m, float momentum) {
#pragma clang fp contract(off)       // TF1 fused-batch-norm moving update (unbiased variance): moving -= (moving - batch) * momentum
    const float ub = var * (m / fmaxf(m - 1.f, 1.f));
    const float dm = (*mm - mu) * momentum, dv = (*mv - ub) * momentum;
    *mm = *mm - dm;
    *mv = *mv - dv;
}
__device__ __forceinline__ void chan_scale_shift(float gamma, float beta, float mu, float rs, float* sc, float* sh) {
#pragma clang fp contract(off)
    const float scv = gamma * rs;
    const float t = mu * scv;
    *sc = scv;
    *sh = beta - t;
}
// per (ns, g): mean / rstd; per (ns, c): scale / shift; optional TF1 fused-batch-norm moving update
__global__ void k_norm_finalize(const float* __restrict__ sums, const float* __restrict__ pivot,
                                const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int NS, int P, int C, int G,
                                float* mean, float* rstd, float* scale, float* shift, float* moving_mean,
                                float* moving_var, float momentum) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NS * G) return;
    const int ns = idx / G, g = idx % G, cg = C / G;
    // per-channel mean / variance from the (pivot-shifted) sums, then the stable parallel-variance combination
    // over the channels of the group: var_g = mean_c[var_c + (mu_c - mu_g)^2]
    const float invP = 1.f / (float)P;
    if (cg == 1) {                            // batch / instance norm: the fused apply kernels' arithmetic, bit for bit (chan_coeffs)
        float mu1, var1, rs1, sc1, sh1;
        chan_coeffs(sums[((size_t)ns * C + g) * 2], sums[((size_t)ns * C + g) * 2 + 1], pivot ? pivot[(size_t)ns * C + g] : 0.f, invP, eps,
                    &mu1, &var1, &rs1);
        chan_scale_shift(gamma[g], beta[g], mu1, rs1, &sc1, &sh1);
        mean[idx] = mu1;
        rstd[idx] = rs1;
        scale[(size_t)ns * C + g] = sc1;
        shift[(size_t)ns * C + g] = sh1;
        if (momentum > 0.f && moving_mean) moving_update(&moving_mean[g], &moving_var[g], mu1, var1, (float)P, momentum);
        return;
    }
    float mu = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
        const float pv = pivot ? pivot[(size_t)ns * C + c] : 0.f;
        mu += pv + sums[((size_t)ns * C + c) * 2] * invP;
    }
    mu /= (float)cg;
    float var = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
        const float pv = pivot ? pivot[(size_t)ns * C + c] : 0.f;
        const float d1 = sums[((size_t)ns * C + c) * 2] * invP;
        float vc = sums[((size_t)ns * C + c) * 2 + 1] * invP - d1 * d1;
        vc = vc > 0.f ? vc : 0.f;
        const float dm = pv + d1 - mu;
        var += vc + dm * dm;
    }
    var /= (float)cg;
    const float m = (float)P * (float)cg;
    const float rs = rsqrtf(var + eps);
    mean[idx] = mu;
    rstd[idx] = rs;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
        const float sc = gamma[c] * rs;
        scale[(size_t)ns * C + c] = sc;
        shift[(size_t)ns * C + c] = beta[c] - mu * sc;
    }
    if (momentum > 0.f && moving_mean) {      // batch norm only (NS == 1, G == C): tf.contrib.layers.batch_norm
        const float unbiased = var * (m / fmaxf(m - 1.f, 1.f));
        moving_mean[g] -= (moving_mean[g] - mu) * momentum;
        moving_var[g] -= (moving_var[g] - unbiased) * momentum;
    }
}

__global__ void k_bn_infer_scale_shift(const float* gamma, const float* beta, const float* mm, const float* mv,
                                       float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(mv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
}

struct BnInferDesc {
    const float *gamma, *beta, *mm, *mv;
    float *scale, *shift;
    int C;
    float eps;
};
__global__ void k_bn_infer_scale_shift_multi(const BnInferDesc* __restrict__ descs) {
    const BnInferDesc d = descs[blockIdx.y];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.C; c += gridDim.x * blockDim.x) {
        const float sc = d.gamma[c] * rsqrtf(d.mv[c] + d.eps);
        d.scale[c] = sc;
        d.shift[c] = d.beta[c] - d.mm[c] * sc;
    }
}

// y = act(x*scale[ns][c] + shift[ns][c]).  A thread owns one channel vector (cv) and walks over pixels, so the
// per-channel coefficients live in registers; blockDim = CV * PL (as for the statistics kernels).
template <typename TI, typename TO, int V>
__global__ void k_affine_act(const TI* __restrict__ x, const float* __restrict__ scale,
                             const float* __restrict__ shift, TO* __restrict__ y, int P, int C, int PL, int chunk,
                             int act) {
    const int CV = C / V;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    if (pl >= PL) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        sc[j] = scale[(size_t)ns * C + cv * V + j];
        sh[j] = shift[(size_t)ns * C + cv * V + j];
    }
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {       // four pixels per trip, loads first (see k_norm_stats)
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) VecIO<TI, V>::load(x, ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[u][j] = act_fwd(fmaf(v[u][j], sc[j], sh[j]), act);
            VecIO<TO, V>::store(y, ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V, v[u]);
        }
    }
    for (; p < p1; p += PL) {
        const size_t off = ((size_t)ns * P + p) * C + (size_t)cv * V;
        float v[V];
        VecIO<TI, V>::load(x, off, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = act_fwd(fmaf(v[j], sc[j], sh[j]), act);
        VecIO<TO, V>::store(y, off, v);
    }
}

// ---- fused variants: the per-(ns, g) finalisation is recomputed by every thread for its own channels (a handful of
// loads), so the two tiny "finalize" launches per layer disappear; block (0, ns) publishes the statistics.
__device__ __forceinline__ void group_stats(const float* __restrict__ sums, const float* __restrict__ pivot, int ns,
                                            int C, int g, int cg, float invP, float eps, float* mu_out, float* var_out) {
    float mu = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c)
        mu += (pivot ? pivot[(size_t)ns * C + c] : 0.f) + sums[((size_t)ns * C + c) * 2] * invP;
    mu /= (float)cg;
    float var = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
        const float d1 = sums[((size_t)ns * C + c) * 2] * invP;
        float vc = sums[((size_t)ns * C + c) * 2 + 1] * invP - d1 * d1;
        vc = vc > 0.f ? vc : 0.f;
        const float dm = (pivot ? pivot[(size_t)ns * C + c] : 0.f) + d1 - mu;
        var += vc + dm * dm;
    }
    *mu_out = mu;
    *var_out = var / (float)cg;
}

// HN > 0: a 1x1 head with HN outputs reads a = act(norm(x)) and nothing else does (the likelihood's top layer feeding y_lvl0,
// likelihoods.py:220): its forward is computed here from the values just produced -- yh[p][o] = bh[o] + sum_c a[p][c] wh[c][o], the
// sum over the channel vectors of a pixel by lane shuffles (C / V a power of two <= 64) -- instead of by a pass of its own over a.
struct HeadFw {
    const float *w, *b;
    float* y;
    int ph, pw;          // > 0: x is in the packed pixel order of the phase-form convolution (csrc/upconv.hip: [B][ph][pw][(a, b)] pixels), y is
                         // written as the hi-res map [B][2 ph][2 pw] -- the depth-to-space permutation rides on the apply pass
};
// pixel index in the packed order [B][h][w][(a, b)] -> pixel index of the hi-res map [B][2h][2w]
__device__ __forceinline__ size_t packed_to_hi_pixel(size_t pix, int h, int w) {
    const unsigned q = (unsigned)(pix >> 2), ab = (unsigned)pix & 3u;
    const unsigned t = q / (unsigned)w, j = q - t * (unsigned)w, b = t / (unsigned)h, i = t - b * (unsigned)h;
    return ((size_t)b * 2 * h + 2 * i + (ab >> 1)) * 2 * w + 2 * j + (ab & 1u);
}
template <typename TI, typename TO, int V, int HN = 0>
__global__ void k_norm_apply_fused(const TI* __restrict__ x, const float* __restrict__ sums,
                                   const float* __restrict__ pivot, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, TO* __restrict__ y, float* mean_out,
                                   float* rstd_out, float* scale_out, float* shift_out, float* moving_mean,
                                   float* moving_var, float momentum, int P, int C, int G, int PL, int chunk, int act, int nrep,
                                   HeadFw hd) {
    const int CV = C / V, cg = C / G;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    // The statistics are finalised ONCE per block, cooperatively, into LDS: thread g derives (mean, rstd) of statistic g, thread c
    // the (scale, shift) of channel c.  (Every thread deriving the eight channels it streams -- 30-50 loads of the same few cache
    // lines per thread -- made the prologue 4-5 us per block and a launch with many blocks a hot spot in one L2 channel: these
    // kernels ran at 2.4-4.7 TB/s where a plain copy of the same tensor reaches 6-7.)
    extern __shared__ float lds_f[];
    float* gst = lds_f;                          // [G][2]  mean, rstd
    float* cof = lds_f + 2 * G;                  // [C][2]  scale, shift
    const float invP = 1.f / (float)P;
    const bool pub = blockIdx.x == 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float mu, var;
        if (cg == 1) {                                      // batch / instance norm: one channel per statistic
            float2 sq = *reinterpret_cast<const float2*>(sums + ((size_t)ns * C + g) * 2);
            for (int r = 1; r < nrep; ++r) {                    // (replicated accumulators: nrep is 1 on every shipped path)
                const float2 q = *reinterpret_cast<const float2*>(sums + (((size_t)r * gridDim.y + ns) * C + g) * 2);
                sq.x += q.x; sq.y += q.y;
            }
            float rs1;
            chan_coeffs(sq.x, sq.y, pivot ? pivot[(size_t)ns * C + g] : 0.f, invP, eps, &mu, &var, &rs1);
        } else {
            group_stats(sums, pivot, ns, C, g, cg, invP, eps, &mu, &var);
        }
        const float rs = rsqrtf(var + eps);
        gst[2 * g] = mu;
        gst[2 * g + 1] = rs;
        if (pub) {
            mean_out[ns * G + g] = mu;
            rstd_out[ns * G + g] = rs;
            if (momentum > 0.f && moving_mean)            // batch norm (G == C): TF1 fused-batch-norm moving update
                moving_update(&moving_mean[g], &moving_var[g], mu, var, (float)P * (float)cg, momentum);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg;
        float scv, shv;
        chan_scale_shift(gamma[c], beta[c], gst[2 * g], gst[2 * g + 1], &scv, &shv);
        cof[2 * c] = scv;
        cof[2 * c + 1] = shv;
        if (pub) {
            scale_out[(size_t)ns * C + c] = scv;
            shift_out[(size_t)ns * C + c] = shv;
        }
    }
    __syncthreads();
    if (pl >= PL) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        sc[j] = cof[2 * (cv * V + j)];
        sh[j] = cof[2 * (cv * V + j) + 1];
    }
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float hw[V][HN > 0 ? HN : 1];
    if constexpr (HN > 0) {
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int o = 0; o < HN; ++o) hw[j][o] = hd.w[(size_t)(cv * V + j) * HN + o];
    }
    auto head = [&](size_t pix, const float (&a)[V]) {     // (all CV lanes of the pixel are active together)
        if constexpr (HN > 0) {
            float part[HN];
#pragma unroll
            for (int o = 0; o < HN; ++o) part[o] = 0.f;
#pragma unroll
            for (int j = 0; j < V; j += 2) {                   // the head reads a as it is stored (bf16)
                const unsigned w2 = f2bf_pk(a[j], a[j + 1]);
                const float a0 = __uint_as_float(w2 << 16), a1 = __uint_as_float(w2 & 0xffff0000u);
#pragma unroll
                for (int o = 0; o < HN; ++o) part[o] = fmaf(a1, hw[j + 1][o], fmaf(a0, hw[j][o], part[o]));
            }
            for (int m = 1; m < CV; m <<= 1)
#pragma unroll
                for (int o = 0; o < HN; ++o) part[o] += __shfl_xor(part[o], m, 64);
            if (cv == 0)
#pragma unroll
                for (int o = 0; o < HN; ++o) hd.y[pix * HN + o] = part[o] + hd.b[o];
        }
    };
    auto yoff = [&](size_t pix) -> size_t { return (hd.pw > 0 ? packed_to_hi_pixel(pix, hd.ph, hd.pw) : pix) * C + (size_t)cv * V; };
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {       // four pixels per trip, loads first (see k_norm_stats)
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) VecIO<TI, V>::load(x, ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[u][j] = act_fwd(fmaf(v[u][j], sc[j], sh[j]), act);
            // (HN > 0, y == NULL: a is not written at all -- in a training plan its only other reader, the head's filter gradient,
            // re-forms it from x: 268 MB less to write for the likelihood's top layer)
            if (HN == 0 || y != nullptr) VecIO<TO, V>::store(y, yoff((size_t)ns * P + p + u * PL), v[u]);
            head((size_t)ns * P + p + u * PL, v[u]);
        }
    }
    for (; p < p1; p += PL) {
        const size_t off = ((size_t)ns * P + p) * C + (size_t)cv * V;
        float v[V];
        VecIO<TI, V>::load(x, off, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = act_fwd(fmaf(v[j], sc[j], sh[j]), act);
        if (HN == 0 || y != nullptr) VecIO<TO, V>::store(y, yoff((size_t)ns * P + p), v);
        head((size_t)ns * P + p, v);
    }
}

// The apply pass of a layer whose output also feeds a 2 x 2 average pool (round 5; posteriors.py:80-82, priors.py:76-78: averagepool2D
// of pre_z[i - 1] in front of every encoder level): a thread takes a 2 x 2 pixel quad -- four loads, four stores of a = act(norm(x)) and
// ONE store of their average -- so the pooled tensor costs no pass of its own over a (tfwrapper/layers.py:44-54, tf.nn.avg_pool 2x2
// stride 2; H and W even).  The average is taken of the values AS STORED (bf16), in k_avgpool_fwd's order: bit-identical to the two
// launches.  bf16 in / out, 16-byte channel vectors; statistics finalised once per block as in k_norm_apply_fused.
__global__ void k_norm_apply_pool(const bf16_t* __restrict__ x, const float* __restrict__ sums, const float* __restrict__ pivot,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                  bf16_t* __restrict__ ypool, float* mean_out, float* rstd_out, float* scale_out, float* shift_out,
                                  float* moving_mean, float* moving_var, float momentum, int P, int C, int G, int H, int W, int PL,
                                  int chunk, int act) {
    constexpr int V = 8;
    const int CV = C / V, cg = C / G;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float lds_f[];
    float* gst = lds_f;                          // [G][2]  mean, rstd
    float* cof = lds_f + 2 * G;                  // [C][2]  scale, shift
    const float invP = 1.f / (float)P;
    const bool pub = blockIdx.x == 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float mu, var;
        if (cg == 1) {
            const float2 sq = *reinterpret_cast<const float2*>(sums + ((size_t)ns * C + g) * 2);
            float rs1;
            chan_coeffs(sq.x, sq.y, pivot ? pivot[(size_t)ns * C + g] : 0.f, invP, eps, &mu, &var, &rs1);
        } else {
            group_stats(sums, pivot, ns, C, g, cg, invP, eps, &mu, &var);
        }
        const float rs = rsqrtf(var + eps);
        gst[2 * g] = mu;
        gst[2 * g + 1] = rs;
        if (pub) {
            mean_out[ns * G + g] = mu;
            rstd_out[ns * G + g] = rs;
            if (momentum > 0.f && moving_mean) moving_update(&moving_mean[g], &moving_var[g], mu, var, (float)P * (float)cg, momentum);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg;
        float scv, shv;
        chan_scale_shift(gamma[c], beta[c], gst[2 * g], gst[2 * g + 1], &scv, &shv);
        cof[2 * c] = scv;
        cof[2 * c + 1] = shv;
        if (pub) {
            scale_out[(size_t)ns * C + c] = scv;
            shift_out[(size_t)ns * C + c] = shv;
        }
    }
    __syncthreads();
    if (pl >= PL) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        sc[j] = cof[2 * (cv * V + j)];
        sh[j] = cof[2 * (cv * V + j) + 1];
    }
    const int Hh = H >> 1, Wh = W >> 1, Q = P >> 2;               // quads of this sample group (P = images x H x W)
    const int q0 = blockIdx.x * chunk, q1 = min(Q, q0 + chunk);
    for (int q = q0 + pl; q < q1; q += PL) {
        const int bl = q / (Hh * Wh), r = q - bl * (Hh * Wh);
        const int yq = r / Wh, xq = r - yq * Wh;
        const size_t p00 = (size_t)ns * P + ((size_t)bl * H + 2 * yq) * W + 2 * xq;
        const size_t off[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};      // k_avgpool_fwd's order: (0,0), (0,1), (1,0), (1,1)
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) VecIO<bf16_t, V>::load(x, off[u] * C + (size_t)cv * V, v[u]);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < V; j += 2) {
                const float a0 = act_fwd(fmaf(v[u][j], sc[j], sh[j]), act), a1 = act_fwd(fmaf(v[u][j + 1], sc[j + 1], sh[j + 1]), act);
                w[j >> 1] = f2bf_pk(a0, a1);
                acc[j] += __uint_as_float(w[j >> 1] << 16);                      // the values as stored
                acc[j + 1] += __uint_as_float(w[j >> 1] & 0xffff0000u);
            }
            *reinterpret_cast<uint4*>(y + off[u] * C + (size_t)cv * V) = make_uint4(w[0], w[1], w[2], w[3]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] *= 0.25f;
        const size_t po = (size_t)ns * Q + q;
        VecIO<bf16_t, V>::store(ypool, po * C + (size_t)cv * V, acc);
    }
}

// HN > 0 (backward of the layer whose only reader is a 1x1 head, see HeadFw): the upstream gradient dA = dyh wh^T is a rank-HN
// function of the head's tiny gradient -- it is formed here per element (rounded to the storage type, as the head's data-gradient
// launch would have stored it) instead of being written by that launch and read back by the two backward passes.
struct HeadBw {
    const float *dy, *w;
    int ph, pw, pc;      // pw > 0: x / dx are in the packed pixel order of the phase-form convolution, dA is the hi-res map [B][2 ph][2 pw][pc]
};
template <typename TD, int V, int HN>
__device__ __forceinline__ void load_da(const TD* __restrict__ dA, size_t off, const HeadBw& hb, size_t pix,
                                        const float (&hw)[V][HN > 0 ? HN : 1], float (&dv)[V]) {
    if constexpr (HN == 0) {
        if (hb.pw > 0) off = packed_to_hi_pixel(pix, hb.ph, hb.pw) * hb.pc + (off - pix * hb.pc);      // (space-to-depth on the fly)
        VecIO<TD, V>::load(dA, off, dv);
    } else {
        static_assert(HN == 0 || (V % 2 == 0 && std::is_same<TD, bf16_t>::value), "head-derived dA: bf16, even vector width");
        float d[HN];                                      // one 8- / 16-byte load per pixel
        if constexpr (HN == 2) {
            const float2 q = *reinterpret_cast<const float2*>(hb.dy + pix * 2);
            d[0] = q.x; d[1] = q.y;
        } else {
            const float4 q = *reinterpret_cast<const float4*>(hb.dy + pix * 4);
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        }
#pragma unroll
        for (int j = 0; j < V; j += 2) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int o = 0; o < HN; ++o) { a0 = fmaf(d[o], hw[j][o], a0); a1 = fmaf(d[o], hw[j + 1][o], a1); }
            const unsigned w2 = f2bf_pk(a0, a1);          // stored precision of dA (v_cvt_pk_bf16_f32: round to nearest even)
            dv[j] = __uint_as_float(w2 << 16);
            dv[j + 1] = __uint_as_float(w2 & 0xffff0000u);
        }
    }
}

template <typename TD, typename TX, typename TO, int V, int HN = 0>
__global__ void k_norm_bwd_apply_fused(const TD* __restrict__ dA, const TX* __restrict__ x,
                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       const float* __restrict__ gamma, const float* __restrict__ sums2,
                                       TO* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int P,
                                       int C, int G, int PL, int chunk, int act, int nrep,
                                       const float* __restrict__ fsums, const float* __restrict__ fpivot,
                                       float* __restrict__ dbias, HeadBw hb) {
    const int CV = C / V, cg = C / G;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    float hw[V][HN > 0 ? HN : 1];
    if constexpr (HN > 0) {
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int o = 0; o < HN; ++o) hw[j][o] = hb.w[(size_t)(cv * V + j) * HN + o];
    }
    // finalisation once per block, cooperatively, through LDS (see k_norm_apply_fused):
    //   st[c][2]  = sum over the accumulator replicas of {sum g, sum g * xhat}
    //   sg[g][2]  = S0, S1 = sum over the statistic's channels of gamma_c * st[c]
    //   cof[c][5] = a, b, c of dx = a * g + b + c * x, and the layer's (scale, shift) for the activation gradient
    extern __shared__ float st[];
    float* sg = st + 2 * C;
    float* cof = sg + 2 * G;
    {
        const size_t rstride = (size_t)gridDim.y * 2 * C;    // sums2[nrep][NS][C][2]
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float a = 0.f;
            for (int r = 0; r < nrep; ++r) a += sums2[r * rstride + (size_t)ns * 2 * C + i];
            st[i] = a;
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float S0 = 0.f, S1 = 0.f;
        for (int q = g * cg; q < (g + 1) * cg; ++q) {
            const float gm = gamma[q];
            S0 += gm * st[2 * q];
            S1 += gm * st[2 * q + 1];
        }
        sg[2 * g] = S0;
        sg[2 * g + 1] = S1;
    }
    __syncthreads();
    const float inv_m = 1.f / ((float)P * (float)cg);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg, sgi = ns * G + g;
        const float rs = rstd[sgi], mu = mean[sgi], S0 = sg[2 * g], S1 = sg[2 * g + 1];
        const float ca = rs * gamma[c];
        const float cc = -rs * rs * S1 * inv_m;
        cof[5 * c] = ca;
        cof[5 * c + 1] = -rs * S0 * inv_m - cc * mu;
        cof[5 * c + 2] = cc;
        cof[5 * c + 3] = scale[(size_t)ns * C + c];
        cof[5 * c + 4] = shift[(size_t)ns * C + c];
        if (blockIdx.x == 0) {
            const float t0 = st[2 * c], t1 = st[2 * c + 1];      // this channel's {sum g, sum g * xhat}
            atomicAdd(&dbeta[c], t0);
            atomicAdd(&dgamma[c], t1);
            if (dbias) {
                // gradient of the bias the producing convolution adds BEFORE the normalisation (group / instance norm keep it,
                // layers.py:126-132): sum_p dx[ns, p, c] = a * sum g + P * b + c * sum x, in closed form from the per-channel
                // sums of both passes -- no pass over dx.  (sum x - P * mean from the shifted forward sums: no cancellation.)
                const float dsum = fsums[((size_t)ns * C + c) * 2] + (float)P * ((fpivot ? fpivot[(size_t)ns * C + c] : 0.f) - mu);
                atomicAdd(&dbias[c], fmaf(ca, t0, fmaf(cc, dsum, -(float)P * rs * S0 * inv_m)));
            }
        }
    }
    __syncthreads();
    if (pl >= PL) return;
    float sc[V], sh[V], ca[V], cb[V], cc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const float* q = cof + 5 * (cv * V + j);
        ca[j] = q[0]; cb[j] = q[1]; cc[j] = q[2]; sc[j] = q[3]; sh[j] = q[4];
    }
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {       // four pixels per trip, loads first (see k_norm_stats)
        float xv[4][V], dv[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t off = ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V;
            VecIO<TX, V>::load(x, off, xv[u]);
            load_da<TD, V, HN>(dA, off, hb, (size_t)ns * P + p + u * PL, hw, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float gq = dv[u][j] * act_grad_pre(fmaf(xv[u][j], sc[j], sh[j]), act);
                o[j] = fmaf(ca[j], gq, fmaf(cc[j], xv[u][j], cb[j]));
            }
            VecIO<TO, V>::store(dx, ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V, o);
        }
    }
    for (; p < p1; p += PL) {
        const size_t off = ((size_t)ns * P + p) * C + (size_t)cv * V;
        float xv[V], dv[V], o[V];
        VecIO<TX, V>::load(x, off, xv);
        load_da<TD, V, HN>(dA, off, hb, (size_t)ns * P + p, hw, dv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gq = dv[j] * act_grad_pre(fmaf(xv[j], sc[j], sh[j]), act);
            o[j] = fmaf(ca[j], gq, fmaf(cc[j], xv[j], cb[j]));
        }
        VecIO<TO, V>::store(dx, off, o);
    }
}

// sums2[ns][c][2] += {sum g, sum g*xhat},  g = dA * act'(x*scale+shift), xhat = (x-mean)*rstd
template <typename TD, typename TX, int V, int HN = 0>
__global__ __launch_bounds__(256) void k_norm_bwd_reduce(const TD* __restrict__ dA, const TX* __restrict__ x,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                  float* __restrict__ sums2, int P, int C, int G, int PL, int chunk, int act, int nrep, HeadBw hb) {
    const int CV = C / V;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    float hw[V][HN > 0 ? HN : 1];
    if constexpr (HN > 0) {
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int o = 0; o < HN; ++o) hw[j][o] = hb.w[(size_t)(cv * V + j) * HN + o];
    }
    extern __shared__ float red[];
    // Same-address fp32 atomics retire at ~1 per 45 ns on MI355X (memory-side), so with a thousand blocks the adds into
    // one sums2 entry -- not the streaming -- set the kernel time: block b adds into replica b % nrep, the consumer
    // (k_norm_bwd_apply_fused) sums the replicas.
    sums2 += (size_t)(blockIdx.x % nrep) * gridDim.y * 2 * C;
    float s1[V], s2[V], sc[V], sh[V], mu[V], rs[V];
    const int cg = C / G;
    // per-channel constants through LDS: one load per channel and block instead of one per channel and thread (red is reused
    // for the block reduction below, after the barrier that ends the streaming loop)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        red[4 * c] = scale[(size_t)ns * C + c];
        red[4 * c + 1] = shift[(size_t)ns * C + c];
        red[4 * c + 2] = mean[(size_t)ns * G + c / cg];
        red[4 * c + 3] = rstd[(size_t)ns * G + c / cg];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        s1[j] = s2[j] = 0.f;
        sc[j] = red[4 * c]; sh[j] = red[4 * c + 1]; mu[j] = red[4 * c + 2]; rs[j] = red[4 * c + 3];
    }
    __syncthreads();                             // constants read: red is free for the partial sums
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    if (pl < PL) {
        int p = p0 + pl;
        for (; p + 3 * PL < p1; p += 4 * PL) {   // four pixels per trip, loads first (see k_norm_stats)
            float xv[4][V], dv[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t off = ((size_t)ns * P + p + u * PL) * C + (size_t)cv * V;
                VecIO<TX, V>::load(x, off, xv[u]);
                load_da<TD, V, HN>(dA, off, hb, (size_t)ns * P + p + u * PL, hw, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float g = dv[u][j] * act_grad_pre(xv[u][j] * sc[j] + sh[j], act);
                    s1[j] += g;
                    s2[j] += g * (xv[u][j] - mu[j]) * rs[j];
                }
        }
        for (; p < p1; p += PL) {
            float xv[V], dv[V];
            const size_t off = ((size_t)ns * P + p) * C + (size_t)cv * V;
            VecIO<TX, V>::load(x, off, xv);
            load_da<TD, V, HN>(dA, off, hb, (size_t)ns * P + p, hw, dv);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float g = dv[j] * act_grad_pre(xv[j] * sc[j] + sh[j], act);
                s1[j] += g;
                s2[j] += g * (xv[j] - mu[j]) * rs[j];
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            red[(pl * C + cv * V + j) * 2 + 0] = s1[j];
            red[(pl * C + cv * V + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float a = 0.f;
        for (int q = 0; q < PL; ++q) a += red[q * 2 * C + i];
        atomicAdd(&sums2[(size_t)ns * 2 * C + i], a);
    }
}

// ---- one-pass batch-norm backward (round 6) ------------------------------------------------------------------------------------
// k_norm_bwd_reduce + k_norm_bwd_apply_fused read dA and x twice (two launches, 4 + 6 B per element).  For tensors that fit the
// register file -- up to 256 blocks x 256 threads x NSLOT pixels x 8 channels -- ONE launch reads them once: every thread keeps its
// (dA, x) vectors packed in registers, the block adds its per-channel partial sums into sums2, a GRID BARRIER follows, and the
// gradient is formed from the registers (4 + 2 B per element, one launch).  Batch norm only (one statistic per channel), bf16.
// Grid barrier (cdna_hip_programming.md, Guideline 16): eight arrival shards (block % 8: one per XCD when blocks land round-robin --
// a speed heuristic, never a correctness assumption) and a top word counting finished shards; every word is an agent-scope atomic,
// zeroed before the launch by the caller (the plan's per-step memset); release fence + drained vmcnt before the arrival, one relaxed
// poll loop with s_sleep, one acquire fence after it.  RESIDENCY: the grid is at most 256 blocks (one per CU) of 256 threads with at
// most 256 registers and < 40 KB of LDS, so that TWO such launches -- the engine replays two lanes side by side -- are resident
// together and neither waits for a block the other one keeps off the chip; the spin is bounded and reports through bar[PHX_BAR_TIMEOUT].
#define PHX_BAR_STRIDE 32
#define PHX_BAR_TOP (8 * PHX_BAR_STRIDE)
#define PHX_BAR_TIMEOUT (9 * PHX_BAR_STRIDE)
#define PHX_BAR_WORDS (10 * PHX_BAR_STRIDE)
// PLAIN_STORES: the blocks exchange data written by plain stores (needs the L2 write-back of a release fence, ~1.7 us and more with
// dirty lines); false: everything exchanged went through agent-scope atomics, which are performed at the coherence point already.
template <bool PLAIN_STORES>
__device__ __forceinline__ void phx_grid_barrier(unsigned* __restrict__ bar, int nblocks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its stores / atomics have left
    __syncthreads();
    if (threadIdx.x == 0) {
        if (PLAIN_STORES) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait: restated)
        }
        unsigned want;
        if (nblocks <= 64) {                                  // few blocks: one level (one atomic round trip less)
            __hip_atomic_fetch_add(bar + PHX_BAR_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            want = (unsigned)nblocks;
        } else {
            const int shard = blockIdx.x & 7;
            const unsigned nshard = (unsigned)((nblocks - shard + 7) >> 3);
            const unsigned prev = __hip_atomic_fetch_add(bar + shard * PHX_BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1 == nshard) __hip_atomic_fetch_add(bar + PHX_BAR_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            want = 8u;
        }
        unsigned spins = 0;
        while (__hip_atomic_load(bar + PHX_BAR_TOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 21)) {                       // ~ seconds: never on a healthy launch; the result is then wrong, not hung
                __hip_atomic_store(bar + PHX_BAR_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int NSLOT, int ACT>
__global__ __launch_bounds__(256, 2) void k_bn_bwd_onepass(const bf16_t* __restrict__ dA, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ sums2,
                                                            unsigned* __restrict__ bar, bf16_t* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int P, int C, int PL,
                                                            int chunk, int nrep) {
    constexpr int act = ACT;          // (a run-time activation code makes every element evaluate all three derivatives)
    constexpr int V = 8;
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float red[];               // phase 1: 4 C constants, then PL x 2C partial sums; phase 2: 2C sums + 5C coefficients
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        red[4 * c] = scale[c];
        red[4 * c + 1] = shift[c];
        red[4 * c + 2] = mean[c];
        red[4 * c + 3] = rstd[c];
    }
    __syncthreads();
    float sc[V], sh[V], mu[V], rs[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        sc[j] = red[4 * c]; sh[j] = red[4 * c + 1]; mu[j] = red[4 * c + 2]; rs[j] = red[4 * c + 3];
    }
    __syncthreads();
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    uint4 xq[NSLOT], dq[NSLOT];
    if (pl < PL) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            const int p = p0 + pl + u * PL;
            const size_t off = (size_t)(p < p1 ? p : p1 - 1) * C + (size_t)cv * V;      // (clamped: always a valid address)
            xq[u] = *reinterpret_cast<const uint4*>(x + off);
            dq[u] = *reinterpret_cast<const uint4*>(dA + off);
            if (p >= p1) dq[u] = make_uint4(0u, 0u, 0u, 0u);                             // dA = 0: contributes nothing
        }
        float s1[V], s2[V];
#pragma unroll
        for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            const unsigned xw[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w}, dw[4] = {dq[u].x, dq[u].y, dq[u].z, dq[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x0 = __uint_as_float(xw[k] << 16), x1 = __uint_as_float(xw[k] & 0xffff0000u);
                const float d0 = __uint_as_float(dw[k] << 16), d1 = __uint_as_float(dw[k] & 0xffff0000u);
                const float g0 = d0 * act_grad_pre(x0 * sc[2 * k] + sh[2 * k], act);
                const float g1 = d1 * act_grad_pre(x1 * sc[2 * k + 1] + sh[2 * k + 1], act);
                s1[2 * k] += g0;
                s2[2 * k] += g0 * (x0 - mu[2 * k]) * rs[2 * k];
                s1[2 * k + 1] += g1;
                s2[2 * k + 1] += g1 * (x1 - mu[2 * k + 1]) * rs[2 * k + 1];
            }
            __builtin_amdgcn_sched_barrier(0);           // slot by slot: the unpacked values of one slot are dead before the next is opened
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            red[(pl * C + cv * V + j) * 2 + 0] = s1[j];
            red[(pl * C + cv * V + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    float* srep = sums2 + (size_t)(blockIdx.x % nrep) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float a = 0.f;
        for (int q = 0; q < PL; ++q) a += red[q * 2 * C + i];
        atomicAdd(&srep[i], a);
    }
    phx_grid_barrier<false>(bar, gridDim.x);      // (the only exchanged data are the atomically added sums)
    // (the packed registers are opaque from here on: otherwise the compiler keeps phase 1's unpacked x and g values alive across the
    // barrier -- 16 floats per slot instead of 8 packed registers -- and spills)
#pragma unroll
    for (int u = 0; u < NSLOT; ++u)
        asm volatile("" : "+v"(xq[u].x), "+v"(xq[u].y), "+v"(xq[u].z), "+v"(xq[u].w), "+v"(dq[u].x), "+v"(dq[u].y), "+v"(dq[u].z), "+v"(dq[u].w));
    // every block finalises for itself (k_norm_bwd_apply_fused's arithmetic, one statistic per channel: S0 = gamma t0, S1 = gamma t1)
    float* st = red;
    float* cof = red + 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < nrep; ++r) a += __hip_atomic_load(sums2 + (size_t)r * 2 * C + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st[i] = a;
    }
    __syncthreads();
    const float inv_m = 1.f / (float)P;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float gm = gamma[c], r_ = rstd[c], m_ = mean[c];
        const float t0 = st[2 * c], t1 = st[2 * c + 1];
        const float S0 = gm * t0, S1 = gm * t1;
        const float ca = r_ * gm;
        const float cc = -r_ * r_ * S1 * inv_m;
        cof[3 * c] = ca;
        cof[3 * c + 1] = -r_ * S0 * inv_m - cc * m_;
        cof[3 * c + 2] = cc;
        if (blockIdx.x == 0) {
            atomicAdd(&dbeta[c], t0);
            atomicAdd(&dgamma[c], t1);
        }
    }
    __syncthreads();
    if (pl >= PL) return;
    float ca[V], cb[V], cc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const float* q = cof + 3 * (cv * V + j);
        ca[j] = q[0]; cb[j] = q[1]; cc[j] = q[2];
    }
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
        const int p = p0 + pl + u * PL;
        if (p < p1) {
            const unsigned xw[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w}, dw[4] = {dq[u].x, dq[u].y, dq[u].z, dq[u].w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x0 = __uint_as_float(xw[k] << 16), x1 = __uint_as_float(xw[k] & 0xffff0000u);
                const float d0 = __uint_as_float(dw[k] << 16), d1 = __uint_as_float(dw[k] & 0xffff0000u);
                const float g0 = d0 * act_grad_pre(fmaf(x0, sc[2 * k], sh[2 * k]), act);
                const float g1 = d1 * act_grad_pre(fmaf(x1, sc[2 * k + 1], sh[2 * k + 1]), act);
                o[k] = f2bf_pk(fmaf(ca[2 * k], g0, fmaf(cc[2 * k], x0, cb[2 * k])), fmaf(ca[2 * k + 1], g1, fmaf(cc[2 * k + 1], x1, cb[2 * k + 1])));
            }
            *reinterpret_cast<uint4*>(dx + (size_t)p * C + (size_t)cv * V) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ void k_norm_bwd_finalize(const float* __restrict__ sums2, const float* __restrict__ gamma, float* S,
                                    float* dgamma, float* dbeta, int NS, int C, int G) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C / G;
    if (idx < NS * G) {
        const int ns = idx / G, g = idx % G;
        float a = 0.f, b = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            a += gamma[c] * sums2[((size_t)ns * C + c) * 2];
            b += gamma[c] * sums2[((size_t)ns * C + c) * 2 + 1];
        }
        S[idx * 2] = a;
        S[idx * 2 + 1] = b;
    }
    if (idx < C) {
        float a = 0.f, b = 0.f;
        for (int ns = 0; ns < NS; ++ns) {
            a += sums2[((size_t)ns * C + idx) * 2];
            b += sums2[((size_t)ns * C + idx) * 2 + 1];
        }
        dbeta[idx] += a;
        dgamma[idx] += b;
    }
}

// dx = rstd*(gamma*g - S0/m - xhat*S1/m) folded to dx = a*g + b + c*x with per-(ns, channel) coefficients in
// registers:  a = rstd*gamma,  c = -rstd^2*S1/m,  b = -rstd*S0/m - c*mean;  g = dA * act'(x*scale+shift).
template <typename TD, typename TX, typename TO, int V>
__global__ void k_norm_bwd_apply(const TD* __restrict__ dA, const TX* __restrict__ x,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ gamma, const float* __restrict__ S, TO* __restrict__ dx,
                                 int P, int C, int G, int PL, int chunk, int act) {
    const int CV = C / V, cg = C / G;
    const int ns = blockIdx.y;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    if (pl >= PL) return;
    const float inv_m = 1.f / ((float)P * (float)cg);
    float sc[V], sh[V], ca[V], cb[V], cc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        const int sg = ns * G + c / cg;
        const float rs = rstd[sg], mu = mean[sg];
        sc[j] = scale[(size_t)ns * C + c];
        sh[j] = shift[(size_t)ns * C + c];
        ca[j] = rs * gamma[c];
        cc[j] = -rs * rs * S[sg * 2 + 1] * inv_m;
        cb[j] = -rs * S[sg * 2] * inv_m - cc[j] * mu;
    }
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    for (int p = p0 + pl; p < p1; p += PL) {
        const size_t off = ((size_t)ns * P + p) * C + (size_t)cv * V;
        float xv[V], dv[V], o[V];
        VecIO<TX, V>::load(x, off, xv);
        VecIO<TD, V>::load(dA, off, dv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float g = dv[j] * act_grad_pre(fmaf(xv[j], sc[j], sh[j]), act);
            o[j] = fmaf(ca[j], g, fmaf(cc[j], xv[j], cb[j]));
        }
        VecIO<TO, V>::store(dx, off, o);
    }
}

template <typename TD, typename TY, typename TO>
__global__ void k_act_bwd(const TD* __restrict__ dy, const TY* __restrict__ y, TO* __restrict__ dpre, size_t n, int act) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        stf<TO>(dpre, i, ldf<TD>(dy, i) * act_grad_out(ldf<TY>(y, i), act));
}

// =================================================================================================
// tf.nn.avg_pool 2x2 stride 2 SAME
template <typename T, int V>
__global__ void k_avgpool_fwd(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2, CV = C / V;
    const size_t items = (size_t)B * OH * OW * CV;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        size_t r = it / CV;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        int cnt = 0;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                if (iy < H && ix < W) {
                    float v[V];
                    VecIO<T, V>::load(x, (((size_t)b * H + iy) * W + ix) * C + (size_t)cv * V, v);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] += v[j];
                    ++cnt;
                }
            }
        const float inv = 1.f / (float)cnt;
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] *= inv;
        VecIO<T, V>::store(y, it * V, acc);
    }
}

template <typename T, int V>
__global__ void k_avgpool_bwd(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int accumulate) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2, CV = C / V;
    const size_t items = (size_t)B * H * W * CV;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        size_t r = it / CV;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        const int oy = iy >> 1, ox = ix >> 1;
        const int cnt = ((2 * oy + 1 < H) ? 2 : 1) * ((2 * ox + 1 < W) ? 2 : 1);
        float v[V];
        VecIO<T, V>::load(dy, (((size_t)b * OH + oy) * OW + ox) * C + (size_t)cv * V, v);
        const float inv = 1.f / (float)cnt;
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] *= inv;
        if (accumulate) {                            // dx already holds another reader's gradient contribution
            float o[V];
            VecIO<T, V>::load(dx, it * V, o);
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] += o[j];
        }
        VecIO<T, V>::store(dx, it * V, v);
    }
}

// Block order of the two resize kernels.  Their items share rows with their vertical neighbours (the forward pass reads input row
// y0 + 1, the adjoint output rows 2 iy - 1 .. 2 iy + 1), and a row of items is several blocks long -- dispatched round-robin over the
// eight XCDs, neighbouring rows land in different L2s and every shared row is fetched from HBM once per L2 (measured, round 5:
// the adjoint fetched 1.7x, the forward pass 2.1x its input).  Banded: XCD k (blockIdx % 8) walks the k-th contiguous eighth of each
// grid sweep in order, so vertically adjacent items meet in ONE L2 a few blocks apart.  (Any grid that is not a multiple of 8 keeps
// the plain order; the item -> value map does not change, only who computes it.)
__device__ __forceinline__ size_t xcd_banded_block() {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    return (!(PHX_TILE_BANDS & 2) || (nb & 7u)) ? b : (b & 7u) * (nb >> 3) + (b >> 3);
}

// TF 1.12 ResizeBilinear x2, legacy coordinates: out[2k] = in[k], out[2k+1] = (in[k] + in[min(k+1,n-1)])/2
template <typename T, int V>
__global__ void k_bilinear_up2x_fwd(const T* __restrict__ x, T* __restrict__ y, int B, int h, int w, int C) {
    // One item = one channel vector of one INPUT pixel and the 2 x 2 output pixels it anchors (TF1 legacy resize, layers.py:336-345:
    // out[2k] = in[k], out[2k + 1] = (in[k] + in[min(k + 1, n - 1)]) / 2 per axis): four loads for four stores.  (One item per OUTPUT
    // vector re-loaded the four neighbours for each of them and ran at 3.1 TB/s on 64 x 64 x 64 x 192; a copy reaches 6.5.)
    const int OW = 2 * w, CV = C / V;
    const size_t items = (size_t)B * h * w * CV;
    for (size_t it = xcd_banded_block() * (size_t)blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        size_t r = it / CV;
        const int x0 = (int)(r % w); r /= w;
        const int y0 = (int)(r % h);
        const size_t base = (r / h) * h;                   // b * h
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        float a[V], bb[V], c[V], d[V], o00[V], o01[V], o10[V], o11[V];
        VecIO<T, V>::load(x, ((base + y0) * w + x0) * C + (size_t)cv * V, a);
        VecIO<T, V>::load(x, ((base + y0) * w + x1) * C + (size_t)cv * V, bb);
        VecIO<T, V>::load(x, ((base + y1) * w + x0) * C + (size_t)cv * V, c);
        VecIO<T, V>::load(x, ((base + y1) * w + x1) * C + (size_t)cv * V, d);
#pragma unroll
        for (int j = 0; j < V; ++j) {                      // (the expressions of the per-output form, fx / fy in {0, 0.5})
            const float top = a[j] + (bb[j] - a[j]) * 0.5f;
            const float bot = c[j] + (d[j] - c[j]) * 0.5f;
            o00[j] = a[j] + (c[j] - a[j]) * 0.f;
            o01[j] = top + (bot - top) * 0.f;
            o10[j] = a[j] + (c[j] - a[j]) * 0.5f;
            o11[j] = top + (bot - top) * 0.5f;
        }
        const size_t orow = ((base * 2 + 2 * y0) * OW + 2 * x0) * C + (size_t)cv * V;
        VecIO<T, V>::store(y, orow, o00);
        VecIO<T, V>::store(y, orow + C, o01);
        VecIO<T, V>::store(y, orow + (size_t)OW * C, o10);
        VecIO<T, V>::store(y, orow + (size_t)OW * C + C, o11);
    }
}

// adjoint (gather): 1-D taps of input k: (2k, 1), (2k+1, 1/2 [+1/2 if k == n-1]), (2k-1, 1/2 if k >= 1)
template <typename T, int V>
__global__ void k_bilinear_up2x_bwd(const T* __restrict__ dy, T* __restrict__ dx, int B, int h, int w, int C, int accumulate) {
    const int OH = 2 * h, OW = 2 * w, CV = C / V;
    const size_t items = (size_t)B * h * w * CV;
    for (size_t it = xcd_banded_block() * (size_t)blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(it % CV);
        size_t r = it / CV;
        const int ix = (int)(r % w); r /= w;
        const int iy = (int)(r % h);
        const int b = (int)(r / h);
        int ty[3], tx[3];
        float wy[3], wx[3];
        ty[0] = 2 * iy; wy[0] = 1.f;
        ty[1] = 2 * iy + 1; wy[1] = (iy == h - 1) ? 1.f : 0.5f;
        ty[2] = 2 * iy - 1; wy[2] = (iy >= 1) ? 0.5f : 0.f;
        tx[0] = 2 * ix; wx[0] = 1.f;
        tx[1] = 2 * ix + 1; wx[1] = (ix == w - 1) ? 1.f : 0.5f;
        tx[2] = 2 * ix - 1; wx[2] = (ix >= 1) ? 0.5f : 0.f;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        if (accumulate) VecIO<T, V>::load(dx, it * V, acc);
        for (int a = 0; a < 3; ++a) {
            if (wy[a] == 0.f) continue;
            for (int c = 0; c < 3; ++c) {
                if (wx[c] == 0.f) continue;
                float v[V];
                VecIO<T, V>::load(dy, (((size_t)b * OH + ty[a]) * OW + tx[c]) * C + (size_t)cv * V, v);
                const float ww = wy[a] * wx[c];
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += ww * v[j];
            }
        }
        VecIO<T, V>::store(dx, it * V, acc);
    }
}

// channel concat / split of two NHWC tensors
template <typename T>
__global__ void k_concat2(const T* __restrict__ a, int Ca, const T* __restrict__ b, int Cb, T* __restrict__ out, size_t npix) {
    const int C = Ca + Cb;
    const size_t n = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        out[i] = c < Ca ? a[p * Ca + c] : b[p * Cb + (c - Ca)];
    }
}
template <typename T>
__global__ void k_split2(const T* __restrict__ in, T* __restrict__ a, int Ca, T* __restrict__ b, int Cb, size_t npix) {
    const int C = Ca + Cb;
    const size_t n = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        if (c < Ca) { if (a) a[p * Ca + c] = in[i]; }
        else { if (b) b[p * Cb + (c - Ca)] = in[i]; }
    }
}
// 16-byte-chunk variants (Ca, Cb multiples of 8 elements for bf16 / 4 for f32): T16 = uint4
__global__ void k_concat2_v16(const uint4* __restrict__ a, int Ca16, const uint4* __restrict__ b, int Cb16,
                              uint4* __restrict__ out, size_t npix) {
    const int C = Ca16 + Cb16;
    const size_t n = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        out[i] = c < Ca16 ? a[p * Ca16 + c] : b[p * Cb16 + (c - Ca16)];
    }
}
__global__ void k_split2_v16(const uint4* __restrict__ in, uint4* __restrict__ a, int Ca16, uint4* __restrict__ b,
                             int Cb16, size_t npix) {
    const int C = Ca16 + Cb16;
    const size_t n = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        if (c < Ca16) { if (a) a[p * Ca16 + c] = in[i]; }
        else { if (b) b[p * Cb16 + (c - Ca16)] = in[i]; }
    }
}

template <typename T>
__global__ void k_add_inplace(T* __restrict__ dst, const T* __restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        stf<T>(dst, i, ldf<T>(dst, i) + ldf<T>(src, i));
}
// 8-element vectors (16-byte accesses), four per thread and trip with the loads first (the element-wise form above moved
// 2.4 TB/s); both pointers 16-byte aligned
template <typename T>
__global__ void k_add_inplace_v16(T* __restrict__ dst, const T* __restrict__ src, size_t nvec) {
    constexpr int VE = 8;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        float a[4][VE], b[4][VE];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            VecIO<T, VE>::load(dst, (i + u * stride) * VE, a[u]);
            VecIO<T, VE>::load(src, (i + u * stride) * VE, b[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < VE; ++j) a[u][j] += b[u][j];
            VecIO<T, VE>::store(dst, (i + u * stride) * VE, a[u]);
        }
    }
    for (; i < nvec; i += stride) {
        float a[VE], b[VE];
        VecIO<T, VE>::load(dst, i * VE, a);
        VecIO<T, VE>::load(src, i * VE, b);
#pragma unroll
        for (int j = 0; j < VE; ++j) a[j] += b[j];
        VecIO<T, VE>::store(dst, i * VE, a);
    }
}
template <typename T>
__global__ void k_channel_sum(const T* __restrict__ x, float* __restrict__ out, size_t npix, int C) {
    // block (c-chunk of 64 channels) x pixel slab; lanes along channels for coalescing
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int prow = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C)
        for (size_t p = blockIdx.x * 4 + prow; p < npix; p += (size_t)gridDim.x * 4) a += ldf<T>(x, p * C + c);
    __shared__ float sh[256];
    sh[threadIdx.x] = a;
    __syncthreads();
    if (prow == 0 && c < C) atomicAdd(&out[c], sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}
// the same sum, 16-byte loads: block = (C / 8 channel vectors) x PL pixel lanes, a chunk of pixels per block
template <typename T>
__global__ void k_channel_sum_v8(const T* __restrict__ x, float* __restrict__ out, size_t npix, int C, int PL, size_t chunk) {
    const int CV = C / 8;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float red[];   // [PL][C]
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    const size_t p0 = blockIdx.x * chunk, p1 = min(npix, p0 + chunk);
    if (pl < PL) {
        size_t p = p0 + pl;
        for (; p + 3 * (size_t)PL < p1; p += 4 * (size_t)PL) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) VecIO<T, 8>::load(x, (p + (size_t)u * PL) * C + (size_t)cv * 8, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += v[u][j];
        }
        for (; p < p1; p += PL) {
            float v[8];
            VecIO<T, 8>::load(x, p * C + (size_t)cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pl * C + cv * 8 + j] = a[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        float t = 0.f;
        for (int q = 0; q < PL; ++q) t += red[q * C + i];
        atomicAdd(&out[i], t);
    }
}
template <typename TI, typename TO>
__global__ void k_cast(const TI* __restrict__ src, TO* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        stf<TO>(dst, i, ldf<TI>(src, i));
}

template <typename TO>
__global__ void k_posterior_input(const float* __restrict__ x, const uint8_t* __restrict__ s, TO* __restrict__ out,
                                  size_t npix, int nlabels) {
    const int C = 1 + nlabels;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        stf<TO>(out, p * C, x[p]);
        const int lab = s[p];
        for (int c = 0; c < nlabels; ++c) stf<TO>(out, p * C + 1 + c, (c == lab ? 1.f : 0.f) - 0.5f);
    }
}

// global average pool [B][P][C] -> [B][C] (tiny tensors: one block per (b, c))
__global__ void k_gap_fwd(const float* __restrict__ x, float* __restrict__ y, int P, int C) {
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    float a = 0.f;
    for (int p = threadIdx.x; p < P; p += blockDim.x) a += x[((size_t)b * P + p) * C + c];
    a = wave_sum(a);
    if (threadIdx.x == 0) y[blockIdx.x] = a / (float)P;
}
__global__ void k_gap_bwd(const float* __restrict__ dy, float* __restrict__ dx, int P, int C, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t b = i / ((size_t)P * C);
        dx[i] = dy[b * C + c] / (float)P;
    }
}
template <typename TO>
__global__ void k_bcast_fwd(const float* __restrict__ z, TO* __restrict__ out, int P, int C, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t b = i / ((size_t)P * C);
        stf<TO>(out, i, z[b * C + c]);
    }
}
template <typename T>
__global__ void k_bcast_bwd(const T* __restrict__ dout, float* __restrict__ dz, int P, int C) {
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    float a = 0.f;
    for (int p = threadIdx.x; p < P; p += blockDim.x) a += ldf<T>(dout, ((size_t)b * P + p) * C + c);
    __shared__ float sh[4];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
        dz[blockIdx.x] = t;
    }
}

// =================================================================================================
extern "C" {

int phx_norm_stats(const void* x, int dt, float* sums, float* pivot, int NS, int P, int C, void* stream) {
    PHX_REQUIRE(NS > 0 && P > 0 && C > 0, PHX_E_SHAPE, "norm_stats: bad shape");
    PHX_DT_SWITCH(dt, T, PHX_VEC_SWITCH(C, V, {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(norm_geometry(P, C, V, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_stats: C too large");
        if (phx_deterministic()) { chunk = P; nchunks = 1; }       // one block per sample group: no cross-block atomics
        hipLaunchKernelGGL((k_norm_stats<T, V>), dim3(nchunks, NS), dim3(threads), (size_t)PL * C * 2 * sizeof(float),
                           (hipStream_t)stream, (const T*)x, sums, pivot, P, C, PL, chunk);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_norm_reduce_partials(const float* partial, int T, int C, float* sums, void* stream) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(2 * C), dim3(256), 0, (hipStream_t)stream, partial, T, C, sums);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_norm_reduce_partials_ns(const float* partial, int T, int NS, int C, float* sums, void* stream) {
    PHX_REQUIRE(partial && sums && T > 0 && NS > 0 && NS < 65536 && C > 0, PHX_E_INVAL, "norm_reduce_partials_ns: bad argument");
    hipLaunchKernelGGL(k_reduce_partials_ns, dim3((2 * C + 255) / 256, NS), dim3(256), 0, (hipStream_t)stream, partial, T, C, sums);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_norm_finalize(const float* sums, const float* pivot, const float* gamma, const float* beta, float eps, int NS,
                      int P, int C, int G,
                      float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                      float momentum, void* stream) {
    PHX_REQUIRE(G > 0 && C % G == 0, PHX_E_SHAPE, "norm_finalize: C % G != 0");
    PHX_REQUIRE(momentum == 0.f || (NS == 1 && G == C), PHX_E_INVAL, "moving update only for batch norm");
    const int n = NS * G;
    hipLaunchKernelGGL(k_norm_finalize, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, pivot, gamma, beta, eps,
                       NS, P, C, G, mean, rstd, scale, shift, moving_mean, moving_var, momentum);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_bn_infer_scale_shift(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                             float eps, int C, float* scale, float* shift, void* stream) {
    hipLaunchKernelGGL(k_bn_infer_scale_shift, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       moving_mean, moving_var, eps, C, scale, shift);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_bn_infer_scale_shift_multi(const void* descs_dev, int n, void* stream) {
    if (n <= 0) return PHX_OK;
    PHX_REQUIRE(descs_dev != nullptr, PHX_E_INVAL, "bn_infer_scale_shift_multi: null descriptor table");
    hipLaunchKernelGGL(k_bn_infer_scale_shift_multi, dim3(2, n), dim3(128), 0, (hipStream_t)stream, (const BnInferDesc*)descs_dev);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_affine_act(const void* x, int x_dt, const float* scale, const float* shift, void* y, int y_dt, int NS, int P,
                   int C, int act, void* stream) {
    PHX_DT_SWITCH(x_dt, TI, PHX_DT_SWITCH(y_dt, TO, PHX_VEC_SWITCH(C, V, {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(stream_geometry(P, C, V, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "affine_act: C too large");
        hipLaunchKernelGGL((k_affine_act<TI, TO, V>), dim3(nchunks, NS), dim3(threads), 0, (hipStream_t)stream,
                           (const TI*)x, scale, shift, (TO*)y, P, C, PL, chunk, act);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

static int norm_apply_impl(const void* x, int x_dt, const float* sums, int nrep, const float* pivot, const float* gamma,
                           const float* beta, float eps, void* y, int y_dt, float* mean, float* rstd, float* scale,
                           float* shift, float* moving_mean, float* moving_var, float momentum, int NS, int P, int C,
                           int G, int act, HeadFw hd, int hn, void* stream);
int phx_norm_head_supported(int C, int nout, int x_dt, int y_dt) {
    const int cv = C / 8;
    return (x_dt == PHX_BF16 && y_dt == PHX_BF16 && C % 8 == 0 && (nout == 2 || nout == 4) && cv >= 1 && cv <= 64 && (cv & (cv - 1)) == 0) ? 1 : 0;
}
int phx_norm_apply_fused_head(const void* x, int x_dt, const float* sums, const float* pivot, const float* gamma,
                              const float* beta, float eps, void* y, int y_dt, float* mean, float* rstd, float* scale,
                              float* shift, float* moving_mean, float* moving_var, float momentum, int NS, int P, int C,
                              int G, int act, const float* w_head, const float* b_head, int nout, float* y_head, void* stream) {
    PHX_REQUIRE(phx_norm_head_supported(C, nout, x_dt, y_dt) && w_head && b_head && y_head, PHX_E_SHAPE,
                "norm_apply_fused_head: bf16, C / 8 a power of two <= 64, nout in {2, 4}");
    return norm_apply_impl(x, x_dt, sums, 1, pivot, gamma, beta, eps, y, y_dt, mean, rstd, scale, shift, moving_mean, moving_var, momentum,
                           NS, P, C, G, act, HeadFw{w_head, b_head, y_head}, nout, stream);
}
static int norm_apply_impl(const void* x, int x_dt, const float* sums, int nrep, const float* pivot, const float* gamma,
                           const float* beta, float eps, void* y, int y_dt, float* mean, float* rstd, float* scale,
                           float* shift, float* moving_mean, float* moving_var, float momentum, int NS, int P, int C,
                           int G, int act, HeadFw hd, int hn, void* stream) {
    PHX_REQUIRE(G > 0 && C % G == 0, PHX_E_SHAPE, "norm_apply_fused: C % G != 0");
    PHX_REQUIRE(nrep == 1 || (nrep > 1 && G == C && pivot == nullptr), PHX_E_INVAL, "norm_apply_fused: replicas only for one channel per statistic, no pivot");
    if (hn > 0) {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(stream_geometry(P, C, 8, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_apply_fused_head: C too large");
#define NAH_LAUNCH(HNv)                                                                                                          \
        hipLaunchKernelGGL((k_norm_apply_fused<bf16_t, bf16_t, 8, HNv>), dim3(nchunks, NS), dim3(threads), (size_t)2 * (C + G) * sizeof(float), \
                           (hipStream_t)stream, (const bf16_t*)x, sums, pivot, gamma, beta, eps, (bf16_t*)y, mean, rstd, scale, shift,  \
                           moving_mean, moving_var, momentum, P, C, G, PL, chunk, act, nrep, hd)
        if (hn == 2) NAH_LAUNCH(2); else NAH_LAUNCH(4);
#undef NAH_LAUNCH
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    PHX_REQUIRE(momentum == 0.f || (NS == 1 && G == C), PHX_E_INVAL, "moving update only for batch norm");
    PHX_DT_SWITCH(x_dt, TI, PHX_DT_SWITCH(y_dt, TO, PHX_VEC_SWITCH(C, V, {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(stream_geometry(P, C, V, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_apply_fused: C too large");
        hipLaunchKernelGGL((k_norm_apply_fused<TI, TO, V>), dim3(nchunks, NS), dim3(threads), (size_t)2 * (C + G) * sizeof(float), (hipStream_t)stream,
                           (const TI*)x, sums, pivot, gamma, beta, eps, (TO*)y, mean, rstd, scale, shift, moving_mean,
                           moving_var, momentum, P, C, G, PL, chunk, act, nrep, hd);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
/* the apply pass + the 2 x 2 average pool of its output in one launch (bf16, C % 8 == 0, H and W even; P = images per statistic x H x W) */
int phx_norm_apply_pool_supported(int H, int W, int C) { return (H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && C / 8 <= 256) ? 1 : 0; }
int phx_norm_apply_pool(const void* x, const float* sums, const float* pivot, const float* gamma, const float* beta, float eps, void* y,
                        void* y_pool, float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                        float momentum, int NS, int P, int C, int G, int H, int W, int act, void* stream) {
    PHX_REQUIRE(phx_norm_apply_pool_supported(H, W, C) && G > 0 && C % G == 0 && P % (H * W) == 0, PHX_E_SHAPE,
                "norm_apply_pool: H, W even, C % 8 == 0, C % G == 0, P a multiple of H * W");
    PHX_REQUIRE(x && y && y_pool && sums, PHX_E_INVAL, "norm_apply_pool: null argument");
    PHX_REQUIRE(momentum == 0.f || (NS == 1 && G == C), PHX_E_INVAL, "moving update only for batch norm");
    int PL, threads, chunk, nchunks;
    PHX_REQUIRE(stream_geometry(P / 4, C, 8, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_apply_pool: C too large");
    hipLaunchKernelGGL(k_norm_apply_pool, dim3(nchunks, NS), dim3(threads), (size_t)2 * (C + G) * sizeof(float), (hipStream_t)stream,
                       (const bf16_t*)x, sums, pivot, gamma, beta, eps, (bf16_t*)y, (bf16_t*)y_pool, mean, rstd, scale, shift, moving_mean,
                       moving_var, momentum, P, C, G, H, W, PL, chunk, act);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_norm_apply_fused(const void* x, int x_dt, const float* sums, const float* pivot, const float* gamma,
                         const float* beta, float eps, void* y, int y_dt, float* mean, float* rstd, float* scale,
                         float* shift, float* moving_mean, float* moving_var, float momentum, int NS, int P, int C,
                         int G, int act, void* stream) {
    return norm_apply_impl(x, x_dt, sums, 1, pivot, gamma, beta, eps, y, y_dt, mean, rstd, scale, shift, moving_mean, moving_var, momentum,
                           NS, P, C, G, act, HeadFw{nullptr, nullptr, nullptr}, 0, stream);
}

/* the phase-form convolution's layer (csrc/upconv.hip): x in the packed pixel order, y written as the hi-res map */
int phx_norm_apply_fused_d2s(const void* x, int x_dt, const float* sums, const float* pivot, const float* gamma, const float* beta, float eps,
                             void* y, int y_dt, float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                             float momentum, int NS, int P, int C, int G, int act, int h, int w, void* stream) {
    PHX_REQUIRE(h > 0 && w > 0 && ((size_t)NS * P) % ((size_t)4 * h * w) == 0 && (NS == 1 || P == 4 * h * w), PHX_E_SHAPE,
                "norm_apply_fused_d2s: NS * P = images x 4 h w, a sample group = one image or all of them");
    return norm_apply_impl(x, x_dt, sums, 1, pivot, gamma, beta, eps, y, y_dt, mean, rstd, scale, shift, moving_mean, moving_var, momentum,
                           NS, P, C, G, act, HeadFw{nullptr, nullptr, nullptr, h, w}, 0, stream);
}

int phx_norm_bwd_apply_fused(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                             const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx,
                             int dx_dt, float* dgamma, float* dbeta, int NS, int P, int C, int G, int act, int nrep,
                             void* stream) {
    return phx_norm_bwd_apply_fused_bias(dA, da_dt, x, x_dt, scale, shift, mean, rstd, gamma, sums2, dx, dx_dt, dgamma, dbeta,
                                         nullptr, nullptr, nullptr, NS, P, C, G, act, nrep, stream);
}

static int norm_bwd_apply_impl(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                               const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx,
                               int dx_dt, float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot,
                               float* dbias, int NS, int P, int C, int G, int act, int nrep, int ph, int pw, void* stream);
int phx_norm_bwd_apply_fused_bias(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx,
                                  int dx_dt, float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot,
                                  float* dbias, int NS, int P, int C, int G, int act, int nrep, void* stream) {
    return norm_bwd_apply_impl(dA, da_dt, x, x_dt, scale, shift, mean, rstd, gamma, sums2, dx, dx_dt, dgamma, dbeta, fwd_sums, fwd_pivot, dbias,
                               NS, P, C, G, act, nrep, 0, 0, stream);
}
/* the phase-form convolution's layer: dA is the hi-res map, x and dx are in the packed pixel order (fwd_sums / fwd_pivot / dbias as
 * phx_norm_bwd_apply_fused_bias) */
int phx_norm_bwd_apply_fused_s2d(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                                 const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx, int dx_dt,
                                 float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot, float* dbias, int NS, int P,
                                 int C, int G, int act, int nrep, int h, int w, void* stream) {
    PHX_REQUIRE(h > 0 && w > 0 && ((size_t)NS * P) % ((size_t)4 * h * w) == 0 && (NS == 1 || P == 4 * h * w), PHX_E_SHAPE,
                "norm_bwd_apply_fused_s2d: NS * P = images x 4 h w, a sample group = one image or all of them");
    return norm_bwd_apply_impl(dA, da_dt, x, x_dt, scale, shift, mean, rstd, gamma, sums2, dx, dx_dt, dgamma, dbeta, fwd_sums, fwd_pivot, dbias,
                               NS, P, C, G, act, nrep, h, w, stream);
}
static int norm_bwd_apply_impl(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                               const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx,
                               int dx_dt, float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot,
                               float* dbias, int NS, int P, int C, int G, int act, int nrep, int ph, int pw, void* stream) {
    PHX_REQUIRE(da_dt == dx_dt, PHX_E_INVAL, "norm_bwd_apply_fused: dA and dx dtypes must match");
    PHX_REQUIRE(dbias == nullptr || fwd_sums != nullptr, PHX_E_INVAL, "norm_bwd_apply_fused_bias: dbias needs the forward sums");
    PHX_REQUIRE(nrep >= 1, PHX_E_INVAL, "norm_bwd_apply_fused: nrep >= 1");
    PHX_DT_SWITCH(da_dt, TD, PHX_DT_SWITCH(x_dt, TX, PHX_VEC_SWITCH(C, V, {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(stream_geometry(P, C, V, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_bwd_apply_fused: C too large");
        hipLaunchKernelGGL((k_norm_bwd_apply_fused<TD, TX, TD, V>), dim3(nchunks, NS), dim3(threads),
                           (size_t)(7 * C + 2 * G) * sizeof(float), (hipStream_t)stream, (const TD*)dA, (const TX*)x, scale, shift, mean, rstd, gamma, sums2,
                           (TD*)dx, dgamma, dbeta, P, C, G, PL, chunk, act, nrep, fwd_sums, fwd_pivot, dbias, HeadBw{nullptr, nullptr, ph, pw, C});
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
/* backward of a normalisation layer whose only reader is a 1x1 head (phx_norm_apply_fused_head): the upstream gradient is
 * dA = dy_head w_head^T (rounded to bf16), formed on the fly from dy_head [NS * P][nout] -- the head's data-gradient launch, its
 * [NS * P][C] output and the two reads of it disappear.  bf16 tensors, C % 8 == 0, nout in {2, 4}. */
int phx_norm_bwd_reduce_head(const float* dy_head, const float* w_head, int nout, const void* x, const float* scale, const float* shift,
                             const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act, int nrep,
                             void* stream) {
    PHX_REQUIRE(dy_head && w_head && (nout == 2 || nout == 4) && C % 8 == 0 && nrep >= 1, PHX_E_INVAL, "norm_bwd_reduce_head: bad arguments");
    int PL, threads, chunk, nchunks;
    PHX_REQUIRE(norm_geometry(P, C, 8, &PL, &threads, &chunk, &nchunks, NS, nrep) == 0, PHX_E_SHAPE, "norm_bwd_reduce_head: C too large");
    if (phx_deterministic() && nchunks > nrep) {
        chunk = (P + nrep - 1) / nrep;
        nchunks = (P + chunk - 1) / chunk;
    }
#define NRH_LAUNCH(HNv)                                                                                                           \
    hipLaunchKernelGGL((k_norm_bwd_reduce<bf16_t, bf16_t, 8, HNv>), dim3(nchunks, NS), dim3(threads),                              \
                       (size_t)(PL > 2 ? PL : 2) * C * 2 * sizeof(float), (hipStream_t)stream, (const bf16_t*)nullptr, (const bf16_t*)x, \
                       scale, shift, mean, rstd, sums2, P, C, G, PL, chunk, act, nrep, HeadBw{dy_head, w_head})
    if (nout == 2) NRH_LAUNCH(2); else NRH_LAUNCH(4);
#undef NRH_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_norm_bwd_apply_fused_head(const float* dy_head, const float* w_head, int nout, const void* x, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, const float* gamma, const float* sums2,
                                  void* dx, float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot, float* dbias,
                                  int NS, int P, int C, int G, int act, int nrep, void* stream) {
    PHX_REQUIRE(dy_head && w_head && (nout == 2 || nout == 4) && C % 8 == 0 && nrep >= 1, PHX_E_INVAL, "norm_bwd_apply_fused_head: bad arguments");
    PHX_REQUIRE(dbias == nullptr || fwd_sums != nullptr, PHX_E_INVAL, "norm_bwd_apply_fused_head: dbias needs the forward sums");
    int PL, threads, chunk, nchunks;
    PHX_REQUIRE(stream_geometry(P, C, 8, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_bwd_apply_fused_head: C too large");
#define NAH_LAUNCH(HNv)                                                                                                           \
    hipLaunchKernelGGL((k_norm_bwd_apply_fused<bf16_t, bf16_t, bf16_t, 8, HNv>), dim3(nchunks, NS), dim3(threads),                 \
                       (size_t)(7 * C + 2 * G) * sizeof(float), (hipStream_t)stream, (const bf16_t*)nullptr, (const bf16_t*)x, scale, shift, \
                       mean, rstd, gamma, sums2, (bf16_t*)dx, dgamma, dbeta, P, C, G, PL, chunk, act, nrep, fwd_sums, fwd_pivot, dbias, \
                       HeadBw{dy_head, w_head})
    if (nout == 2) NAH_LAUNCH(2); else NAH_LAUNCH(4);
#undef NAH_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

static int norm_bwd_reduce_impl(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                                const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act,
                                int nrep, int ph, int pw, void* stream);
int phx_norm_bwd_reduce(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                        const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act,
                        int nrep, void* stream) {
    return norm_bwd_reduce_impl(dA, da_dt, x, x_dt, scale, shift, mean, rstd, sums2, NS, P, C, G, act, nrep, 0, 0, stream);
}
int phx_norm_bwd_reduce_s2d(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                            const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act, int nrep, int h,
                            int w, void* stream) {
    PHX_REQUIRE(h > 0 && w > 0 && ((size_t)NS * P) % ((size_t)4 * h * w) == 0 && (NS == 1 || P == 4 * h * w), PHX_E_SHAPE,
                "norm_bwd_reduce_s2d: NS * P = images x 4 h w, a sample group = one image or all of them");
    return norm_bwd_reduce_impl(dA, da_dt, x, x_dt, scale, shift, mean, rstd, sums2, NS, P, C, G, act, nrep, h, w, stream);
}
static int norm_bwd_reduce_impl(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                                const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act,
                                int nrep, int ph, int pw, void* stream) {
    PHX_REQUIRE(nrep >= 1, PHX_E_INVAL, "norm_bwd_reduce: nrep >= 1");
    PHX_DT_SWITCH(da_dt, TD, PHX_DT_SWITCH(x_dt, TX, PHX_VEC_SWITCH(C, V, {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(norm_geometry(P, C, V, &PL, &threads, &chunk, &nchunks, NS, nrep) == 0, PHX_E_SHAPE, "norm_bwd_reduce: C too large");
        if (phx_deterministic() && nchunks > nrep) {               // block b owns replica b: one add per accumulator, the consumer
            chunk = (P + nrep - 1) / nrep;                         // sums the replicas in a fixed order
            nchunks = (P + chunk - 1) / chunk;
        }
        hipLaunchKernelGGL((k_norm_bwd_reduce<TD, TX, V>), dim3(nchunks, NS), dim3(threads),
                           (size_t)(PL > 2 ? PL : 2) * C * 2 * sizeof(float), (hipStream_t)stream, (const TD*)dA, (const TX*)x, scale,
                           shift, mean, rstd, sums2, P, C, G, PL, chunk, act, nrep, HeadBw{nullptr, nullptr, ph, pw, C});
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

// one-pass batch-norm backward: geometry (at most 256 blocks of <= 256 threads; slots = pixels per thread)
static bool bn_onepass_geometry(int P, int C, int* PL, int* threads, int* chunk, int* nblocks, int* nslot) {
    if (C % 8 != 0 || C / 8 > 256 || P < 1) return false;
    const int CV = C / 8;
    *PL = 256 / CV;
    *threads = CV * (*PL);
    int want = (P + (*PL) * 8 - 1) / ((*PL) * 8);            // ~8 pixels per thread ...
    if (want > 256) want = 256;                               // ... on at most one block per CU
    if (want < 1) want = 1;
    *chunk = (P + want - 1) / want;
    *nblocks = (P + *chunk - 1) / (*chunk);
    const int slots = (*chunk + *PL - 1) / (*PL);
    *nslot = slots <= 4 ? 4 : slots <= 8 ? 8 : slots <= 16 ? 16 : 0;
    return *nslot != 0 && (size_t)(*PL) * C * 2 * sizeof(float) <= 40 * 1024 && (size_t)5 * C * sizeof(float) <= 40 * 1024;
}
int phx_bn_bwd_onepass_supported(int P, int C, int act) {
    int PL, threads, chunk, nblocks, nslot;
    return (act == PHX_ACT_RELU && !phx_deterministic()
            && bn_onepass_geometry(P, C, &PL, &threads, &chunk, &nblocks, &nslot)) ? 1 : 0;
}
int phx_bn_bwd_onepass_barrier_words(void) { return PHX_BAR_WORDS; }
int phx_bn_bwd_onepass(const void* dA, const void* x, const float* scale, const float* shift, const float* mean, const float* rstd,
                       const float* gamma, float* sums2, unsigned* barrier, void* dx, float* dgamma, float* dbeta, int P, int C, int act,
                       int nrep, void* stream) {
    int PL, threads, chunk, nblocks, nslot;
    PHX_REQUIRE(dA && x && sums2 && barrier && dx && nrep >= 1, PHX_E_INVAL, "bn_bwd_onepass: null pointer");
    PHX_REQUIRE(bn_onepass_geometry(P, C, &PL, &threads, &chunk, &nblocks, &nslot), PHX_E_SHAPE,
                "bn_bwd_onepass: the tensor does not fit 256 blocks x 16 pixels per thread (phx_bn_bwd_onepass_supported)");
    size_t lds = (size_t)(PL > 2 ? PL : 2) * C * 2 * sizeof(float);
    if (lds < (size_t)5 * C * sizeof(float)) lds = (size_t)5 * C * sizeof(float);
    if (lds < (size_t)4 * C * sizeof(float)) lds = (size_t)4 * C * sizeof(float);
#define PHX_ONEPASS(NS_, ACT_)                                                                                                      \
    hipLaunchKernelGGL((k_bn_bwd_onepass<NS_, ACT_>), dim3(nblocks), dim3(threads), lds, (hipStream_t)stream, (const bf16_t*)dA,     \
                       (const bf16_t*)x, scale, shift, mean, rstd, gamma, sums2, barrier, (bf16_t*)dx, dgamma, dbeta, P, C, PL, chunk, nrep)
    PHX_REQUIRE(act == PHX_ACT_RELU, PHX_E_INVAL, "bn_bwd_onepass: relu layers only (every conv + batch norm of the zoo)");
    if (nslot == 4) PHX_ONEPASS(4, PHX_ACT_RELU); else if (nslot == 8) PHX_ONEPASS(8, PHX_ACT_RELU); else PHX_ONEPASS(16, PHX_ACT_RELU);
#undef PHX_ONEPASS
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_bn_small_supported(int P, int C, int dt) { return dt == PHX_BF16 && P >= 1 && P <= 4096 && C % 16 == 0; }

int phx_bn_small_fwd(const void* x, int x_dt, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                     float* scale, float* shift, float* moving_mean, float* moving_var, float momentum, int P, int C,
                     int act, void* stream) {
    PHX_REQUIRE(phx_bn_small_supported(P, C, PHX_BF16), PHX_E_SHAPE, "bn_small_fwd: needs bf16 output, P <= 4096, C % 16 == 0");
    PHX_REQUIRE(x_dt == PHX_BF16 || (x_dt == PHX_F32 && P <= 1024), PHX_E_INVAL, "bn_small_fwd: x is bf16, or fp32 with P <= 1024");
#define BNS_F(NITv, XFv)                                                                                              \
    hipLaunchKernelGGL((k_bn_small_fwd<NITv, XFv>), dim3(C / 16), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta, eps, \
                       (bf16_t*)y, mean, rstd, scale, shift, moving_mean, moving_var, momentum, P, C, act)
    if (x_dt == PHX_F32) { if (P <= 512) BNS_F(1, true); else BNS_F(2, true); }
    else if (P <= 512) BNS_F(1, false); else if (P <= 1024) BNS_F(2, false); else if (P <= 2048) BNS_F(4, false); else BNS_F(8, false);
#undef BNS_F
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_bn_small_bwd(const void* dA, const void* x, int x_dt, const float* scale, const float* shift, const float* mean,
                     const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, int P, int C, int act,
                     void* stream) {
    PHX_REQUIRE(phx_bn_small_supported(P, C, PHX_BF16), PHX_E_SHAPE, "bn_small_bwd: needs bf16 gradients, P <= 4096, C % 16 == 0");
    PHX_REQUIRE(x_dt == PHX_BF16 || (x_dt == PHX_F32 && P <= 1024), PHX_E_INVAL, "bn_small_bwd: x is bf16, or fp32 with P <= 1024");
#define BNS_B(NITv, XFv)                                                                                              \
    hipLaunchKernelGGL((k_bn_small_bwd<NITv, XFv>), dim3(C / 16), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)dA, \
                       x, scale, shift, mean, rstd, gamma, (bf16_t*)dx, dgamma, dbeta, P, C, act)
    if (x_dt == PHX_F32) { if (P <= 512) BNS_B(1, true); else BNS_B(2, true); }
    else if (P <= 512) BNS_B(1, false); else if (P <= 1024) BNS_B(2, false); else if (P <= 2048) BNS_B(4, false); else BNS_B(8, false);
#undef BNS_B
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

// the wide form (k_bn_wide_*): fp32 x given as nz split-K slices xs[z][P][C] (nz >= 1), 4 channels per block
int phx_bn_wide_supported(int P, int C) { return (P >= 1 && P <= 1024 && C % 4 == 0) ? 1 : 0; }
int phx_bn_wide_fwd(const float* xs, int nz, float* xsum, const float* gamma, const float* beta, float eps, void* y, float* mean,
                    float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, float momentum, int P, int C,
                    int act, void* stream) {
    PHX_REQUIRE(phx_bn_wide_supported(P, C) && nz >= 1, PHX_E_SHAPE, "bn_wide_fwd: needs P <= 1024, C % 4 == 0, nz >= 1");
    PHX_REQUIRE(xs && y && mean && rstd && scale && shift && (nz == 1 || xsum), PHX_E_INVAL, "bn_wide_fwd: null argument (xsum is required when nz > 1)");
    PHX_REQUIRE((((uintptr_t)xs | (uintptr_t)xsum | (uintptr_t)y) & 15) == 0, PHX_E_ALIGN, "bn_wide_fwd: 16-byte alignment");
#define BNW_F(NPTv)                                                                                                    \
    hipLaunchKernelGGL((k_bn_wide_fwd<NPTv>), dim3(C / 4), dim3(256), 0, (hipStream_t)stream, xs, nz, (size_t)P * C, xsum, gamma, \
                       beta, eps, (bf16_t*)y, mean, rstd, scale, shift, moving_mean, moving_var, momentum, P, C, act)
    if (P <= 256) BNW_F(1); else if (P <= 512) BNW_F(2); else BNW_F(4);
#undef BNW_F
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_bn_wide_bwd(const void* dA, const float* dA_slices, int nzd, const float* x, const float* scale, const float* shift,
                    const float* mean, const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, int P, int C,
                    int act, void* stream) {
    PHX_REQUIRE(phx_bn_wide_supported(P, C), PHX_E_SHAPE, "bn_wide_bwd: needs P <= 1024, C % 4 == 0");
    PHX_REQUIRE((dA != nullptr) != (dA_slices != nullptr) && (dA_slices == nullptr || nzd >= 1) && x && dx, PHX_E_INVAL,
                "bn_wide_bwd: exactly one of dA / dA_slices");
#define BNW_B(NPTv)                                                                                                    \
    hipLaunchKernelGGL((k_bn_wide_bwd<NPTv>), dim3(C / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dA, dA_slices, nzd, \
                       (size_t)P * C, x, scale, shift, mean, rstd, gamma, (bf16_t*)dx, dgamma, dbeta, P, C, act)
    if (P <= 256) BNW_B(1); else if (P <= 512) BNW_B(2); else BNW_B(4);
#undef BNW_B
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

// ---- group / instance norm, one launch per layer on maps of up to 256 pixels (k_norm_wave_*) ----
int phx_norm_small_supported(int NS, int P, int C, int G, int dt) {
    return dt == PHX_BF16 && NS >= 1 && P >= 1 && P <= 256 && C % 16 == 0 && (G == C || G * 16 == C);
}
// pixel lanes per sample (4, 16, 32), samples per block pass, block passes and sample blocks for the (C / 16, nsb) grid
static void norm_wave_plan(int NS, int P, int C, int* lps, int* passes, int* nsb) {
    *lps = P <= 4 ? 4 : P <= 16 ? 16 : 32;
    const int spb = 4 * (32 / *lps);                           // samples per block pass (4 waves)
    const int need = (NS + spb - 1) / spb;                     // block passes in all
    int want = 512 / (C / 16);                                 // sample blocks for ~512 blocks
    if (want < 1) want = 1;
    if (phx_deterministic()) want = 1;                         // one block per slice: a fixed order for dgamma / dbeta / dbias
    *nsb = need < want ? need : want;
    *passes = (need + *nsb - 1) / *nsb;
    *nsb = (need + *passes - 1) / *passes;
}
#define PHX_NW_SWITCH(P, ...)                                                                                        \
    do {                                                                                                             \
        if ((P) <= 4) { constexpr int LPSv = 4, NITv = 1; __VA_ARGS__; }                                             \
        else if ((P) <= 16) { constexpr int LPSv = 16, NITv = 1; __VA_ARGS__; }                                      \
        else if ((P) <= 32) { constexpr int LPSv = 32, NITv = 1; __VA_ARGS__; }                                      \
        else if ((P) <= 64) { constexpr int LPSv = 32, NITv = 2; __VA_ARGS__; }                                      \
        else if ((P) <= 128) { constexpr int LPSv = 32, NITv = 4; __VA_ARGS__; }                                     \
        else { constexpr int LPSv = 32, NITv = 8; __VA_ARGS__; }                                                     \
    } while (0)

int phx_norm_small_fwd(void* x, const float* ws, int nz, const float* bias, const float* gamma, const float* beta, float eps,
                       void* y, float* mean, float* rstd, float* scale, float* shift, int NS, int P, int C, int G, int act,
                       void* stream) {
    PHX_REQUIRE(phx_norm_small_supported(NS, P, C, G, PHX_BF16), PHX_E_SHAPE,
                "norm_small_fwd: needs bf16, P <= 256, C % 16 == 0, per-channel statistics or 16-channel groups");
    PHX_REQUIRE(x != nullptr && (ws == nullptr || nz >= 1), PHX_E_INVAL, "norm_small_fwd: x (and nz >= 1 with ws)");
    PHX_REQUIRE(ws != nullptr || bias == nullptr, PHX_E_INVAL, "norm_small_fwd: bias only with split-K slices");
    int lps, passes, nsb;
    norm_wave_plan(NS, P, C, &lps, &passes, &nsb);
    PHX_REQUIRE(passes < 65536 && nsb < 65536, PHX_E_SHAPE, "norm_small_fwd: too many samples");
    const dim3 grid(C / 16, nsb);
#define NW_F(SLv, Mv)                                                                                                           \
    PHX_NW_SWITCH(P, hipLaunchKernelGGL((k_norm_wave_fwd<LPSv, NITv, Mv, SLv>), grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ws, \
                                        nz, bias, gamma, beta, eps, (bf16_t*)y, mean, rstd, scale, shift, NS, P, C, passes, act))
    if (ws) { if (G == C) NW_F(true, 0); else NW_F(true, 1); }
    else { if (G == C) NW_F(false, 0); else NW_F(false, 1); }
#undef NW_F
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_norm_small_bwd(const void* dA, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, float* dbias, int NS, int P,
                       int C, int G, int act, void* stream) {
    PHX_REQUIRE(phx_norm_small_supported(NS, P, C, G, PHX_BF16), PHX_E_SHAPE,
                "norm_small_bwd: needs bf16, P <= 256, C % 16 == 0, per-channel statistics or 16-channel groups");
    int lps, passes, nsb;
    norm_wave_plan(NS, P, C, &lps, &passes, &nsb);
    PHX_REQUIRE(passes < 65536 && nsb < 65536, PHX_E_SHAPE, "norm_small_bwd: too many samples");
    const dim3 grid(C / 16, nsb);
#define NW_B(Mv)                                                                                                                \
    PHX_NW_SWITCH(P, hipLaunchKernelGGL((k_norm_wave_bwd<LPSv, NITv, Mv>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dA,   \
                                        (const bf16_t*)x, scale, shift, mean, rstd, gamma, (bf16_t*)dx, dgamma, dbeta, dbias, NS, P, C, passes, act))
    if (G == C) NW_B(0); else NW_B(1);                  // (per-channel statistics: the bias gradient is identically zero)
#undef NW_B
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_norm_bwd_finalize(const float* sums2, const float* gamma, float* S, float* dgamma, float* dbeta, int NS, int C,
                          int G, void* stream) {
    const int n = NS * G > C ? NS * G : C;
    hipLaunchKernelGGL(k_norm_bwd_finalize, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums2, gamma, S,
                       dgamma, dbeta, NS, C, G);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_norm_bwd_apply(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                       const float* mean, const float* rstd, const float* gamma, const float* S, void* dx, int dx_dt,
                       int NS, int P, int C, int G, int act, void* stream) {
    PHX_REQUIRE(da_dt == dx_dt, PHX_E_INVAL, "norm_bwd_apply: dA and dx dtypes must match");
    PHX_DT_SWITCH(da_dt, TD, PHX_DT_SWITCH(x_dt, TX, PHX_VEC_SWITCH(C, V, {
        int PL, threads, chunk, nchunks;
        PHX_REQUIRE(stream_geometry(P, C, V, &PL, &threads, &chunk, &nchunks, NS) == 0, PHX_E_SHAPE, "norm_bwd_apply: C too large");
        hipLaunchKernelGGL((k_norm_bwd_apply<TD, TX, TD, V>), dim3(nchunks, NS), dim3(threads), 0,
                           (hipStream_t)stream, (const TD*)dA, (const TX*)x, scale, shift, mean, rstd, gamma, S, (TD*)dx,
                           P, C, G, PL, chunk, act);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_act_bwd(const void* dy, int dy_dt, const void* y, int y_dt, void* dpre, int dpre_dt, size_t n, int act,
                void* stream) {
    PHX_REQUIRE(dy_dt == dpre_dt, PHX_E_INVAL, "act_bwd: dy and dpre dtypes must match");
    PHX_DT_SWITCH(dy_dt, TD, PHX_DT_SWITCH(y_dt, TY, {
        hipLaunchKernelGGL((k_act_bwd<TD, TY, TD>), dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const TD*)dy, (const TY*)y, (TD*)dpre, n, act);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

#define PHX_SPATIAL_LAUNCH(kern, items_expr, ...)                                                              \
    PHX_DT_SWITCH(dt, T, PHX_VEC_SWITCH(C, V, {                                                                \
        const size_t items = (items_expr) * (size_t)(C / V);                                                   \
        int grid_ = phx_grid_for(items, 256);                                                                  \
        if (grid_ > 8) grid_ = (grid_ + 7) & ~7;          /* whole XCD rounds (xcd_banded_block) */             \
        hipLaunchKernelGGL((kern<T, V>), dim3(grid_), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);         \
    }));                                                                                                       \
    PHX_CHECK_LAUNCH();                                                                                        \
    return PHX_OK;

int phx_avgpool2x2_fwd(const void* x, int dt, void* y, int B, int H, int W, int C, void* stream) {
    PHX_SPATIAL_LAUNCH(k_avgpool_fwd, (size_t)B * ((H + 1) / 2) * ((W + 1) / 2), (const T*)x, (T*)y, B, H, W, C)
}
int phx_avgpool2x2_bwd(const void* dy, int dt, void* dx, int B, int H, int W, int C, void* stream) {
    PHX_SPATIAL_LAUNCH(k_avgpool_bwd, (size_t)B * H * W, (const T*)dy, (T*)dx, B, H, W, C, 0)
}
int phx_avgpool2x2_bwd_acc(const void* dy, int dt, void* dx, int B, int H, int W, int C, void* stream) {
    PHX_SPATIAL_LAUNCH(k_avgpool_bwd, (size_t)B * H * W, (const T*)dy, (T*)dx, B, H, W, C, 1)
}
int phx_bilinear_up2x_fwd(const void* x, int dt, void* y, int B, int h, int w, int C, void* stream) {
    PHX_SPATIAL_LAUNCH(k_bilinear_up2x_fwd, (size_t)B * h * w, (const T*)x, (T*)y, B, h, w, C)
}
int phx_bilinear_up2x_bwd(const void* dy, int dt, void* dx, int B, int h, int w, int C, void* stream) {
    PHX_SPATIAL_LAUNCH(k_bilinear_up2x_bwd, (size_t)B * h * w, (const T*)dy, (T*)dx, B, h, w, C, 0)
}
int phx_bilinear_up2x_bwd_acc(const void* dy, int dt, void* dx, int B, int h, int w, int C, void* stream) {
    PHX_SPATIAL_LAUNCH(k_bilinear_up2x_bwd, (size_t)B * h * w, (const T*)dy, (T*)dx, B, h, w, C, 1)
}

int phx_concat2(const void* a, int Ca, const void* b, int Cb, void* out, size_t npix, int dt, void* stream) {
    const int es = dt == PHX_BF16 ? 2 : 4, per16 = 16 / es;
    if (Ca % per16 == 0 && Cb % per16 == 0) {
        const size_t n = npix * (size_t)((Ca + Cb) / per16);
        hipLaunchKernelGGL(k_concat2_v16, dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a,
                           Ca / per16, (const uint4*)b, Cb / per16, (uint4*)out, npix);
    } else {
        PHX_DT_SWITCH(dt, T, {
            hipLaunchKernelGGL((k_concat2<T>), dim3(phx_grid_for(npix * (Ca + Cb), 256)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)a, Ca, (const T*)b, Cb, (T*)out, npix);
        });
    }
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_split2(const void* in, void* a, int Ca, void* b, int Cb, size_t npix, int dt, void* stream) {
    const int es = dt == PHX_BF16 ? 2 : 4, per16 = 16 / es;
    if (Ca % per16 == 0 && Cb % per16 == 0) {
        const size_t n = npix * (size_t)((Ca + Cb) / per16);
        hipLaunchKernelGGL(k_split2_v16, dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in,
                           (uint4*)a, Ca / per16, (uint4*)b, Cb / per16, npix);
    } else {
        PHX_DT_SWITCH(dt, T, {
            hipLaunchKernelGGL((k_split2<T>), dim3(phx_grid_for(npix * (Ca + Cb), 256)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)in, (T*)a, Ca, (T*)b, Cb, npix);
        });
    }
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_add_inplace(void* dst, const void* src, size_t n, int dt, void* stream) {
    const size_t ve = 8;
    if (n >= 4096 && n % ve == 0 && ((uintptr_t)dst | (uintptr_t)src) % 16 == 0) {
        const size_t nvec = n / ve;
        size_t blocks = (nvec + 256 * 4 - 1) / (256 * 4);           // four vectors per thread
        if (blocks > 4096) blocks = 4096;
        PHX_DT_SWITCH(dt, T, {
            hipLaunchKernelGGL((k_add_inplace_v16<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (T*)dst,
                               (const T*)src, nvec);
        });
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_add_inplace<T>), dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (T*)dst,
                           (const T*)src, n);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_channel_sum_accumulate(const void* x, int dt, float* out, size_t npix, int C, void* stream) {
    if (C % 8 == 0 && C / 8 <= 256 && npix >= 64) {
        const int CV = C / 8, PL = 256 / CV, threads = CV * PL;
        size_t rows = (npix + PL - 1) / PL;
        size_t want = (rows + 31) / 32;                   // ~32 pixels per thread, at least 64 and at most 1024 blocks
        if (want < 64) want = rows < 64 ? rows : 64;
        if (want > 1024) want = 1024;
        if (phx_deterministic()) want = 1;
        const size_t chunk = (npix + want - 1) / want;
        const int nchunks = (int)((npix + chunk - 1) / chunk);
        PHX_DT_SWITCH(dt, T, {
            hipLaunchKernelGGL((k_channel_sum_v8<T>), dim3(nchunks), dim3(threads), (size_t)PL * C * sizeof(float), (hipStream_t)stream,
                               (const T*)x, out, npix, C, PL, chunk);
        });
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    int gx = (int)((npix + 63) / 64);
    if (gx > 256) gx = 256;
    if (gx < 1 || phx_deterministic()) gx = 1;
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_channel_sum<T>), dim3(gx, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const T*)x, out,
                           npix, C);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_cast(const void* src, int src_dt, void* dst, int dst_dt, size_t n, void* stream) {
    PHX_DT_SWITCH(src_dt, TI, PHX_DT_SWITCH(dst_dt, TO, {
        hipLaunchKernelGGL((k_cast<TI, TO>), dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const TI*)src, (TO*)dst, n);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_posterior_input(const float* x, const uint8_t* s, void* out, int out_dt, size_t npix, int nlabels, void* stream) {
    PHX_DT_SWITCH(out_dt, TO, {
        hipLaunchKernelGGL((k_posterior_input<TO>), dim3(phx_grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, x,
                           s, (TO*)out, npix, nlabels);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_global_avgpool_fwd(const float* x, float* y, int B, int P, int C, void* stream) {
    hipLaunchKernelGGL(k_gap_fwd, dim3(B * C), dim3(64), 0, (hipStream_t)stream, x, y, P, C);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_global_avgpool_bwd(const float* dy, float* dx, int B, int P, int C, void* stream) {
    const size_t n = (size_t)B * P * C;
    hipLaunchKernelGGL(k_gap_bwd, dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, P, C, n);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
// out[b * n + k][:] = x[b][:] for k < n: every image's feature map repeated for its n Monte-Carlo samples (16-byte pieces)
__global__ void k_repeat_rows16(const uint4* __restrict__ x, uint4* __restrict__ out, size_t per16, int n, size_t total16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total16; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / per16, e = i - row * per16;
        out[i] = x[(row / n) * per16 + e];
    }
}
int phx_repeat_batch(const void* x, void* out, int B, size_t bytes_per_sample, int n, void* stream) {
    PHX_REQUIRE(bytes_per_sample % 16 == 0 && n >= 1, PHX_E_SHAPE, "repeat_batch: sample size must be a multiple of 16 bytes");
    PHX_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 15) == 0, PHX_E_ALIGN, "repeat_batch: 16-byte alignment");
    const size_t per16 = bytes_per_sample / 16, total16 = per16 * B * n;
    hipLaunchKernelGGL(k_repeat_rows16, dim3(phx_grid_for(total16, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                       (uint4*)out, per16, n, total16);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_broadcast_pixels_fwd(const float* z, void* out, int out_dt, int B, int P, int C, void* stream) {
    const size_t n = (size_t)B * P * C;
    PHX_DT_SWITCH(out_dt, TO, {
        hipLaunchKernelGGL((k_bcast_fwd<TO>), dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, z, (TO*)out,
                           P, C, n);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_broadcast_pixels_bwd(const void* dout, int dt, float* dz, int B, int P, int C, void* stream) {
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_bcast_bwd<T>), dim3(B * C), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dz, P, C);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
