// 1x1 "head" convolutions with a handful of outputs: mu / sigma (zdim0 = 2 or 6 channels) and the per-level logits
// y_lvl (nlabels channels) -- posteriors.py:125-127, priors.py:117-119, likelihoods.py:220.  They are pure streaming
// passes over the feature map (C = 32..192 channels in, NOUT <= 8 out), so a thread owns one 8-channel vector
// (16 bytes of bf16) of a pixel, the NOUT x 8 filter slice lives in registers, and the kernels run at HBM speed.
#include <stdlib.h>
#include "phx_common.h"
#include "philox.h"

template <typename T, int V> struct HVec;
template <> struct HVec<float, 8> {
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
template <> struct HVec<bf16_t, 8> {
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
template <typename T> struct HVec<T, 1> {
    static __device__ __forceinline__ void load(const T* p, size_t i, float o[1]) { o[0] = ldf<T>(p, i); }
    static __device__ __forceinline__ void store(T* p, size_t i, const float o[1]) { stf<T>(p, i, o[0]); }
};

// y[p][o] = act(b[o] + sum_c x[p][c] * w[c][o]);  block = CV x PL threads, HU * PL pixels per iteration: a thread loads its
// 16-byte slice of HU pixels first (one load per iteration and two barriers around an LDS reduction made the kernel latency-bound:
// 1.8 TB/s), the cross-channel-vector sums of all HU * PL pixels go through LDS together
constexpr int HU = 4;
template <typename TX, int V, int NOUT>
__global__ void k_head1x1_fwd(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                              float* __restrict__ y, size_t npix, int C, int PL, int iters, int act) {
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float red[];                  // [HU * PL][NOUT][CV]
    float wr[V][NOUT];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) wr[j][o] = w[(size_t)(cv * V + j) * NOUT + o];
    for (int it = 0; it < iters; ++it) {
        const size_t pbase = ((size_t)blockIdx.x * iters + it) * (HU * PL);
        if (pl < PL) {
            float xv[HU][V];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const size_t p = pbase + u * PL + pl;
                if (p < npix) HVec<TX, V>::load(x, p * C + (size_t)cv * V, xv[u]);
                else
#pragma unroll
                    for (int j = 0; j < V; ++j) xv[u][j] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                float part[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) part[o] = 0.f;
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) part[o] = fmaf(xv[u][j], wr[j][o], part[o]);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) red[((u * PL + pl) * NOUT + o) * CV + cv] = part[o];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < HU * PL * NOUT; t += blockDim.x) {
            const int o = t % NOUT, q = t / NOUT;
            const size_t pp = pbase + q;
            if (pp < npix) {
                float a = bias ? bias[o] : 0.f;
                for (int k = 0; k < CV; ++k) a += red[(q * NOUT + o) * CV + k];
                y[pp * NOUT + o] = act_fwd(a, act);
            }
        }
        __syncthreads();
    }
}

// dx[p][c] = sum_o dy[p][o] * w[c][o]   (HU pixels per trip: the dy loads of a trip are issued together)
template <typename TO, int V, int NOUT>
__global__ void k_head1x1_dgrad(const float* __restrict__ dy, const float* __restrict__ w, TO* __restrict__ dx,
                                size_t npix, int C, int PL, int chunk) {
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    if (pl >= PL) return;
    float wr[V][NOUT];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) wr[j][o] = w[(size_t)(cv * V + j) * NOUT + o];
    const size_t p0 = (size_t)blockIdx.x * chunk;
    const size_t p1 = p0 + chunk < npix ? p0 + chunk : npix;
    auto one = [&](size_t p, const float d[NOUT]) {
        float o8[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a = 0.f;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) a = fmaf(d[o], wr[j][o], a);
            o8[j] = a;
        }
        HVec<TO, V>::store(dx, p * C + (size_t)cv * V, o8);
    };
    size_t p = p0 + pl;
    for (; p + (size_t)(HU - 1) * PL < p1; p += (size_t)HU * PL) {
        float d[HU][NOUT];
#pragma unroll
        for (int u = 0; u < HU; ++u)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) d[u][o] = dy[(p + (size_t)u * PL) * NOUT + o];
#pragma unroll
        for (int u = 0; u < HU; ++u) one(p + (size_t)u * PL, d[u]);
    }
    for (; p < p1; p += PL) {
        float d[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) d[o] = dy[p * NOUT + o];
        one(p, d);
    }
}


// ---- fused latent heads (posteriors.py:125-128, priors.py:117-120): the two 1x1 convolutions of a latent level read the same
// feature map, and the sample follows at once --   mu = x Wmu + bmu;  sigma = softplus(x Wsig + bsig);  z = mu + sigma * eps.
// One pass over x instead of two and one launch instead of three (two heads + phx_reparam_fwd); eps comes from the same Philox
// stream contract as phx_reparam_fwd (block = 4 consecutive per-sample elements, sample = global sample index).
template <typename TX, int V, int Z>
__global__ void k_latent_fwd(const TX* __restrict__ x, const float* __restrict__ wmu, const float* __restrict__ bmu,
                             const float* __restrict__ wsig, const float* __restrict__ bsig, float* __restrict__ mu,
                             float* __restrict__ sigma, float* __restrict__ z, size_t npix, int C, int PL, int iters, int hw,
                             unsigned long long seed, const int32_t* __restrict__ step_dev, int stream_id, int sample_offset) {
    constexpr int NOUT = 2 * Z;
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float red[];                  // [HU * PL][NOUT][CV]
    float wr[V][NOUT];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int o = 0; o < Z; ++o) {
            wr[j][o] = wmu[(size_t)(cv * V + j) * Z + o];
            wr[j][Z + o] = wsig[(size_t)(cv * V + j) * Z + o];
        }
    const unsigned step = z ? (unsigned)(*step_dev) : 0u;
    for (int it = 0; it < iters; ++it) {
        const size_t pbase = ((size_t)blockIdx.x * iters + it) * (HU * PL);
        if (pl < PL) {
            float xv[HU][V];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const size_t p = pbase + u * PL + pl;
                if (p < npix) HVec<TX, V>::load(x, p * C + (size_t)cv * V, xv[u]);
                else
#pragma unroll
                    for (int j = 0; j < V; ++j) xv[u][j] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                float part[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) part[o] = 0.f;
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) part[o] = fmaf(xv[u][j], wr[j][o], part[o]);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) red[((u * PL + pl) * NOUT + o) * CV + cv] = part[o];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < HU * PL * Z; t += blockDim.x) {
            const int zc = t % Z, q = t / Z;
            const size_t pp = pbase + q;
            if (pp < npix) {
                float am = bmu[zc], as = bsig[zc];
                for (int k = 0; k < CV; ++k) {
                    am += red[(q * NOUT + zc) * CV + k];
                    as += red[(q * NOUT + Z + zc) * CV + k];
                }
                as = act_fwd(as, PHX_ACT_SOFTPLUS);
                mu[pp * Z + zc] = am;
                sigma[pp * Z + zc] = as;
                if (z) {
                    const unsigned b = (unsigned)(pp / (size_t)hw), e = (unsigned)(pp % (size_t)hw) * Z + zc;
                    float n[4];
                    philox_normal4(e >> 2, b + (unsigned)sample_offset, (unsigned)stream_id, step, seed, n);
                    z[pp * Z + zc] = am + as * n[e & 3];
                }
            }
        }
        __syncthreads();
    }
}

// backward of the same three operators in one launch: with the upstream gradients dz (of the sample; may be NULL), dmu and dsigma
// (of the KL term; may be NULL)   g_mu = dz + dmu;   g_sig = (dz * eps + dsigma) * softplus'(pre) = (...) * (1 - exp(-sigma));
// dx = g_mu Wmu^T + g_sig Wsig^T  -- written once (two head data gradients, three in-place adds and phx_reparam_bwd before);
// g_mu / g_sig are kept for the deferred filter-gradient launch of the two heads.
template <typename TO, int V, int Z>
__global__ void k_latent_bwd(const float* __restrict__ dz, const float* __restrict__ dmu, const float* __restrict__ dsigma,
                             const float* __restrict__ sigma, const float* __restrict__ wmu, const float* __restrict__ wsig,
                             TO* __restrict__ dx, float* __restrict__ gmu, float* __restrict__ gsig, size_t npix, int C, int PL,
                             int chunk, int hw, unsigned long long seed, const int32_t* __restrict__ step_dev, int stream_id,
                             int sample_offset) {
    constexpr int NOUT = 2 * Z;
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    extern __shared__ float sg[];                   // [chunk][NOUT]
    const size_t p0 = (size_t)blockIdx.x * chunk;
    const size_t p1 = p0 + chunk < npix ? p0 + chunk : npix;
    const unsigned step = dz ? (unsigned)(*step_dev) : 0u;
    for (int t = threadIdx.x; t < (int)(p1 - p0) * Z; t += blockDim.x) {
        const int zc = t % Z, q = t / Z;
        const size_t pp = p0 + q, o = pp * Z + zc;
        float gm = dmu ? dmu[o] : 0.f, gs = dsigma ? dsigma[o] : 0.f;
        if (dz) {
            const unsigned b = (unsigned)(pp / (size_t)hw), e = (unsigned)(pp % (size_t)hw) * Z + zc;
            float n[4];
            philox_normal4(e >> 2, b + (unsigned)sample_offset, (unsigned)stream_id, step, seed, n);
            const float d = dz[o];
            gm += d;
            gs += d * n[e & 3];
        }
        gs *= act_grad_out(sigma[o], PHX_ACT_SOFTPLUS);
        gmu[o] = gm;
        gsig[o] = gs;
        sg[q * NOUT + zc] = gm;
        sg[q * NOUT + Z + zc] = gs;
    }
    __syncthreads();
    if (pl >= PL) return;
    float wr[V][NOUT];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int o = 0; o < Z; ++o) {
            wr[j][o] = wmu[(size_t)(cv * V + j) * Z + o];
            wr[j][Z + o] = wsig[(size_t)(cv * V + j) * Z + o];
        }
    for (size_t p = p0 + pl; p < p1; p += PL) {
        const float* d = sg + (p - p0) * NOUT;
        float o8[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a = 0.f;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) a = fmaf(d[o], wr[j][o], a);
            o8[j] = a;
        }
        HVec<TO, V>::store(dx, p * C + (size_t)cv * V, o8);
    }
}

// dw[c][o] += sum_p x[p][c] * dy[p][o];  db[o] += sum_p dy[p][o]
template <typename TX, int V, int NOUT>
__device__ __forceinline__ void head1x1_wgrad_body(const TX* __restrict__ x, const float* __restrict__ dy,
                                                   float* __restrict__ dw, float* __restrict__ db, size_t npix, int C, int PL,
                                                   int chunk, int bx, const float* __restrict__ xscale = nullptr,
                                                   const float* __restrict__ xshift = nullptr, int xact = 0) {
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    // xscale != NULL (round 5): x is the PRE-normalisation tensor y of the layer whose only reader is this head -- its
    // a = act(y * scale[c] + shift[c]) was never written (phx_norm_apply_fused_head with y == NULL) and is re-formed here, rounded to
    // bf16 as the stored tensor would have been (batch norm: one scale / shift per channel)
    float xsc[V], xsh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        xsc[j] = xscale ? xscale[cv * V + j] : 1.f;
        xsh[j] = xscale ? xshift[cv * V + j] : 0.f;
    }
    auto xform = [&](float (&v)[V]) {
        if (xscale != nullptr) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = act_fwd(fmaf(v[j], xsc[j], xsh[j]), xact);
            if constexpr (V % 2 == 0) {
#pragma unroll
                for (int j = 0; j < V; j += 2) {
                    const unsigned w2 = f2bf_pk(v[j], v[j + 1]);
                    v[j] = __uint_as_float(w2 << 16);
                    v[j + 1] = __uint_as_float(w2 & 0xffff0000u);
                }
            }
        }
    };
    extern __shared__ float red[];                  // [PL][C][NOUT]
    float acc[V][NOUT], accb[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        accb[o] = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j][o] = 0.f;
    }
    const size_t p0 = (size_t)bx * chunk;
    const size_t p1 = p0 + chunk < npix ? p0 + chunk : npix;
    if (pl < PL) {
        size_t p = p0 + pl;
        for (; p + 3 * (size_t)PL < p1; p += 4 * (size_t)PL) {    // four pixels per trip, loads first (latency-bound otherwise)
            float xv[4][V], d[4][NOUT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                HVec<TX, V>::load(x, (p + (size_t)u * PL) * C + (size_t)cv * V, xv[u]);
                xform(xv[u]);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) d[u][o] = dy[(p + (size_t)u * PL) * NOUT + o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) accb[o] += d[u][o];
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) acc[j][o] = fmaf(xv[u][j], d[u][o], acc[j][o]);
            }
        }
        for (; p < p1; p += PL) {
            float xv[V], d[NOUT];
            HVec<TX, V>::load(x, p * C + (size_t)cv * V, xv);
            xform(xv);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                d[o] = dy[p * NOUT + o];
                accb[o] += d[o];
            }
#pragma unroll
            for (int j = 0; j < V; ++j)
#pragma unroll
                for (int o = 0; o < NOUT; ++o) acc[j][o] = fmaf(xv[j], d[o], acc[j][o]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) red[((size_t)pl * C + cv * V + j) * NOUT + o] = acc[j][o];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * NOUT; i += blockDim.x) {
        float a = 0.f;
        for (int q = 0; q < PL; ++q) a += red[(size_t)q * C * NOUT + i];
        atomicAdd(&dw[i], a);
    }
    if (db) {
        __syncthreads();
        if (cv == 0 && pl < PL) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) red[pl * NOUT + o] = accb[o];
        }
        __syncthreads();
        if (threadIdx.x < NOUT) {
            float a = 0.f;
            for (int q = 0; q < PL; ++q) a += red[q * NOUT + threadIdx.x];
            atomicAdd(&db[threadIdx.x], a);
        }
    }
}
template <typename TX, int V, int NOUT>
__global__ void k_head1x1_wgrad(const TX* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                float* __restrict__ db, size_t npix, int C, int PL, int chunk) {
    head1x1_wgrad_body<TX, V, NOUT>(x, dy, dw, db, npix, C, PL, chunk, blockIdx.x);
}
// The head filter gradients are leaves of the backward graph (only Adam reads them): one launch for all heads of a plan,
// after the lanes have joined, instead of 23 latency-bound launches inside the posterior / prior / likelihood chains.
struct HeadWJob {
    const void* x; const float* dy; float* dw; float* db;
    unsigned long long npix;
    int C, PL, chunk, blk0;
    const float *xscale, *xshift;             // non-NULL: x is the pre-normalisation tensor, a = act(x * xscale[c] + xshift[c]) is re-formed on load
    int xact, pad_;
};
static_assert(sizeof(HeadWJob) == 80, "HeadWJob is packed by the host (include/phx.h, engine.py): 80 bytes");
template <typename TX, int NOUT>
__global__ __launch_bounds__(256) void k_head1x1_wgrad_multi(const HeadWJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                         // last job with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const HeadWJob j = jobs[lo];
    // (256-thread blocks; C = 192 uses 240 of them -- the others have pl >= PL and only take part in the barriers)
    head1x1_wgrad_body<TX, 8, NOUT>((const TX*)j.x, j.dy, j.dw, j.db, (size_t)j.npix, j.C, j.PL, j.chunk, (int)blockIdx.x - j.blk0,
                                    j.xscale, j.xshift, j.xact);
}

static int head_geo(int C, int V, int* PL, int* threads) {
    const int CV = C / V;
    if (CV > 256 || CV < 1) return -1;
    *PL = 256 / CV;
    *threads = CV * (*PL);
    return 0;
}

#define HEAD_NOUT_SWITCH(n, N, ...)                                         \
    do {                                                                    \
        if ((n) <= 2) { constexpr int N = 2; __VA_ARGS__; }                 \
        else if ((n) <= 4) { constexpr int N = 4; __VA_ARGS__; }            \
        else if ((n) <= 6) { constexpr int N = 6; __VA_ARGS__; }            \
        else { constexpr int N = 8; __VA_ARGS__; }                          \
    } while (0)
#define HEAD_VEC_SWITCH(C, V, ...)                                          \
    do {                                                                    \
        if ((C) % 8 == 0) { constexpr int V = 8; __VA_ARGS__; }             \
        else { constexpr int V = 1; __VA_ARGS__; }                          \
    } while (0)

extern "C" {

// w is the HWIO 1x1 filter [C][nout] fp32; nout must equal one of the instantiated widths (2, 4, 6, 8)
int phx_head1x1_fwd(const void* x, int x_dt, const float* w, const float* bias, float* y, size_t npix, int C, int nout,
                    int act, void* stream) {
    PHX_REQUIRE(nout == 2 || nout == 4 || nout == 6 || nout == 8, PHX_E_SHAPE, "head1x1: nout in {2,4,6,8}");
    PHX_DT_SWITCH(x_dt, TX, HEAD_VEC_SWITCH(C, V, HEAD_NOUT_SWITCH(nout, N, {
        int PL, threads;
        PHX_REQUIRE(head_geo(C, V, &PL, &threads) == 0, PHX_E_SHAPE, "head1x1: C too large");
        size_t groups = (npix + (size_t)HU * PL - 1) / ((size_t)HU * PL);
        int iters = (int)((groups + 2047) / 2048);
        if (iters < 1) iters = 1;
        const int grid = (int)((groups + iters - 1) / iters);
        hipLaunchKernelGGL((k_head1x1_fwd<TX, V, N>), dim3(grid), dim3(threads), (size_t)HU * PL * N * (C / V) * sizeof(float),
                           (hipStream_t)stream, (const TX*)x, w, bias, y, npix, C, PL, iters, act);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

#define LAT_Z_SWITCH(zd, Zv, ...)                                           \
    do {                                                                    \
        if ((zd) == 2) { constexpr int Zv = 2; __VA_ARGS__; }               \
        else if ((zd) == 4) { constexpr int Zv = 4; __VA_ARGS__; }          \
        else { constexpr int Zv = 6; __VA_ARGS__; }                         \
    } while (0)

int phx_latent_heads_fwd(const void* x, int x_dt, const float* w_mu, const float* b_mu, const float* w_sigma, const float* b_sigma,
                         float* mu, float* sigma, float* z, size_t npix, int C, int zdim, int pix_per_sample, uint64_t seed,
                         const int32_t* step_dev, int stream_id, int sample_offset, void* stream) {
    PHX_REQUIRE(zdim == 2 || zdim == 4 || zdim == 6, PHX_E_SHAPE, "latent_heads: zdim in {2,4,6}");
    PHX_REQUIRE(x && w_mu && b_mu && w_sigma && b_sigma && mu && sigma && pix_per_sample > 0 && npix % (size_t)pix_per_sample == 0 &&
                (z == nullptr || step_dev != nullptr), PHX_E_INVAL, "latent_heads_fwd: bad arguments");
    PHX_DT_SWITCH(x_dt, TX, HEAD_VEC_SWITCH(C, V, LAT_Z_SWITCH(zdim, ZZ, {
        int PL, threads;
        PHX_REQUIRE(head_geo(C, V, &PL, &threads) == 0, PHX_E_SHAPE, "latent_heads: C too large");
        size_t groups = (npix + (size_t)HU * PL - 1) / ((size_t)HU * PL);
        int iters = (int)((groups + 2047) / 2048);
        if (iters < 1) iters = 1;
        const int grid = (int)((groups + iters - 1) / iters);
        hipLaunchKernelGGL((k_latent_fwd<TX, V, ZZ>), dim3(grid), dim3(threads), (size_t)HU * PL * 2 * ZZ * (C / V) * sizeof(float),
                           (hipStream_t)stream, (const TX*)x, w_mu, b_mu, w_sigma, b_sigma, mu, sigma, z, npix, C, PL, iters,
                           pix_per_sample, (unsigned long long)seed, step_dev, stream_id, sample_offset);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_latent_heads_bwd(const float* dz, const float* dmu, const float* dsigma, const float* sigma, const float* w_mu,
                         const float* w_sigma, void* dx, int dx_dt, float* g_mu, float* g_sigma, size_t npix, int C, int zdim,
                         int pix_per_sample, uint64_t seed, const int32_t* step_dev, int stream_id, int sample_offset, void* stream) {
    PHX_REQUIRE(zdim == 2 || zdim == 4 || zdim == 6, PHX_E_SHAPE, "latent_heads: zdim in {2,4,6}");
    PHX_REQUIRE(sigma && w_mu && w_sigma && dx && g_mu && g_sigma && pix_per_sample > 0 && npix % (size_t)pix_per_sample == 0 &&
                (dz == nullptr || step_dev != nullptr), PHX_E_INVAL, "latent_heads_bwd: bad arguments");
    PHX_DT_SWITCH(dx_dt, TO, HEAD_VEC_SWITCH(C, V, LAT_Z_SWITCH(zdim, ZZ, {
        int PL, threads;
        PHX_REQUIRE(head_geo(C, V, &PL, &threads) == 0, PHX_E_SHAPE, "latent_heads: C too large");
        int chunk = PL * 8;
        size_t grid = (npix + chunk - 1) / chunk;
        if (grid > 4096) { chunk = (int)((npix + 4095) / 4096); grid = (npix + chunk - 1) / chunk; }
        hipLaunchKernelGGL((k_latent_bwd<TO, V, ZZ>), dim3((unsigned)grid), dim3(threads), (size_t)chunk * 2 * ZZ * sizeof(float),
                           (hipStream_t)stream, dz, dmu, dsigma, sigma, w_mu, w_sigma, (TO*)dx, g_mu, g_sigma, npix, C, PL, chunk,
                           pix_per_sample, (unsigned long long)seed, step_dev, stream_id, sample_offset);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_head1x1_dgrad(const float* dy, const float* w, void* dx, int dx_dt, size_t npix, int C, int nout, void* stream) {
    PHX_REQUIRE(nout == 2 || nout == 4 || nout == 6 || nout == 8, PHX_E_SHAPE, "head1x1: nout in {2,4,6,8}");
    PHX_DT_SWITCH(dx_dt, TO, HEAD_VEC_SWITCH(C, V, HEAD_NOUT_SWITCH(nout, N, {
        int PL, threads;
        PHX_REQUIRE(head_geo(C, V, &PL, &threads) == 0, PHX_E_SHAPE, "head1x1: C too large");
        int chunk = PL * 8;
        size_t grid = (npix + chunk - 1) / chunk;
        if (grid > 8192) { chunk = (int)((npix + 8191) / 8192); grid = (npix + chunk - 1) / chunk; }
        hipLaunchKernelGGL((k_head1x1_dgrad<TO, V, N>), dim3((unsigned)grid), dim3(threads), 0, (hipStream_t)stream, dy, w,
                           (TO*)dx, npix, C, PL, chunk);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

static void head_wgrad_geometry(size_t npix, int PL, int* chunk, int* grid) {
    int ch = PL * 32;
    size_t g = (npix + ch - 1) / ch;
    // every block ends in C * nout same-address atomics (~45 ns each, serialised): few, long blocks
    const int cap = 1024;
    if (g > (size_t)cap) { ch = (int)((npix + cap - 1) / cap); g = (npix + ch - 1) / ch; }
    if (phx_deterministic()) { ch = (int)npix; g = 1; }     // one block per head: fixed summation order
    *chunk = ch; *grid = (int)g;
}
/* plan4 = {PL, chunk, grid, dynamic LDS bytes} of the launch phx_head1x1_wgrad makes for C % 8 == 0 */
int phx_head1x1_wgrad_plan(size_t npix, int C, int nout, int* plan4) {
    PHX_REQUIRE(C % 8 == 0 && C / 8 <= 256, PHX_E_SHAPE, "head1x1_wgrad_plan: C % 8 == 0 required");
    int PL, threads;
    head_geo(C, 8, &PL, &threads);
    const int N = nout <= 2 ? 2 : nout <= 4 ? 4 : nout <= 6 ? 6 : 8;
    plan4[0] = PL;
    head_wgrad_geometry(npix, PL, &plan4[1], &plan4[2]);
    plan4[3] = (int)((size_t)PL * C * N * sizeof(float));
    return PHX_OK;
}
/* jobs_dev: device array of {const void* x; const float* dy; float* dw; float* db; uint64 npix; int C, PL, chunk, blk0;}
 * (PL, chunk from phx_head1x1_wgrad_plan, blk0 = running sum of the grids), all with the same x dtype and nout, C % 8 == 0 */
int phx_head1x1_wgrad_multi(const void* jobs_dev, int njobs, int total_blocks, int x_dt, int nout, size_t lds_bytes,
                            void* stream) {
    PHX_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, PHX_E_INVAL, "head1x1_wgrad_multi: empty job list");
    PHX_REQUIRE(nout == 2 || nout == 4 || nout == 6 || nout == 8, PHX_E_SHAPE, "head1x1: nout in {2,4,6,8}");
    PHX_DT_SWITCH(x_dt, TX, HEAD_NOUT_SWITCH(nout, N, {
        hipLaunchKernelGGL((k_head1x1_wgrad_multi<TX, N>), dim3((unsigned)total_blocks), dim3(256), lds_bytes,
                           (hipStream_t)stream, (const HeadWJob*)jobs_dev, njobs);
    }));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_head1x1_wgrad(const void* x, int x_dt, const float* dy, float* dw, float* db, size_t npix, int C, int nout,
                      void* stream) {
    PHX_REQUIRE(nout == 2 || nout == 4 || nout == 6 || nout == 8, PHX_E_SHAPE, "head1x1: nout in {2,4,6,8}");
    PHX_DT_SWITCH(x_dt, TX, HEAD_VEC_SWITCH(C, V, HEAD_NOUT_SWITCH(nout, N, {
        int PL, threads;
        PHX_REQUIRE(head_geo(C, V, &PL, &threads) == 0, PHX_E_SHAPE, "head1x1: C too large");
        int chunk, grid;
        head_wgrad_geometry(npix, PL, &chunk, &grid);
        size_t sh = (size_t)PL * C * N * sizeof(float);
        if (sh < (size_t)PL * N * sizeof(float)) sh = (size_t)PL * N * sizeof(float);
        hipLaunchKernelGGL((k_head1x1_wgrad<TX, V, N>), dim3((unsigned)grid), dim3(threads), sh, (hipStream_t)stream,
                           (const TX*)x, dy, dw, db, npix, C, PL, chunk);
    })));
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
