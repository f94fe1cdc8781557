// Reparameterisation sampler, residual multinoulli (cross-entropy) loss, hierarchical KL, TF1 Adam.
#include "philox.h"
#include "phx_common.h"

// ---- z = mu + sigma * eps (posteriors.py:108,128; priors.py:100,120) ------------------------------
// one thread per Philox block of 4 consecutive per-sample elements
__global__ void k_reparam_fwd(const float* __restrict__ mu, const float* __restrict__ sigma, float* __restrict__ z,
                              int B, int per_sample, unsigned long long seed, const int32_t* __restrict__ step_dev,
                              int stream_id, int sample_offset) {
    const int nblk = (per_sample + 3) / 4;
    const size_t total = (size_t)B * nblk;
    const unsigned step = (unsigned)(*step_dev);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / nblk), blk = (int)(i % nblk);
        float n[4];
        philox_normal4((unsigned)blk, (unsigned)(b + sample_offset), (unsigned)stream_id, step, seed, n);
        for (int j = 0; j < 4; ++j) {
            const int e = blk * 4 + j;
            if (e < per_sample) {
                const size_t o = (size_t)b * per_sample + e;
                const float m = mu ? mu[o] : 0.f, s = sigma ? sigma[o] : 1.f;
                z[o] = m + s * n[j];
            }
        }
    }
}
// dsigma = dz * eps   (dmu = dz needs no kernel)
__global__ void k_reparam_bwd(const float* __restrict__ dz, float* __restrict__ dsigma, int B, int per_sample,
                              unsigned long long seed, const int32_t* __restrict__ step_dev, int stream_id,
                              int sample_offset) {
    const int nblk = (per_sample + 3) / 4;
    const size_t total = (size_t)B * nblk;
    const unsigned step = (unsigned)(*step_dev);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / nblk), blk = (int)(i % nblk);
        float n[4];
        philox_normal4((unsigned)blk, (unsigned)(b + sample_offset), (unsigned)stream_id, step, seed, n);
        for (int j = 0; j < 4; ++j) {
            const int e = blk * 4 + j;
            if (e < per_sample) {
                const size_t o = (size_t)b * per_sample + e;
                dsigma[o] = dz[o] * n[j];
            }
        }
    }
}

// ---- residual multinoulli loss (phiseg_model.py:229-262) -------------------------------------------
#define CE_MAXL 8
#define CE_MAXC 8
struct CEArgs {
    const float* s[CE_MAXL];
    float* ds[CE_MAXL];
    int shift[CE_MAXL];
};

// block = 16x16 pixel tile of one image (4 waves, each an 8x8 sub-tile, lane = ly*8+lx) so that the
// f x f NEAREST_NEIGHBOR blocks of the coarse levels reduce with wave shuffles before one atomicAdd.
template <int C>
__global__ void k_residual_ce(CEArgs a, int L, const uint8_t* __restrict__ labels, int B, int H, int W, float gscale,
                              float inv_batch, float* __restrict__ loss_part, float* __restrict__ s_out,
                              float* __restrict__ sm_out, int det) {
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
    const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = ((tile % tiles_x) << 4) + ((wave & 1) << 3) + (lane & 7);
    const int py = ((tile / tiles_x) << 4) + ((wave >> 1) << 3) + (lane >> 3);
    const bool valid = px < W && py < H;
    const int lab = (valid && labels) ? labels[((size_t)b * H + py) * W + px] : 0;
    float A[C], G[CE_MAXL][C], ce[CE_MAXL];
#pragma unroll
    for (int c = 0; c < C; ++c) A[c] = 0.f;
#pragma unroll
    for (int l = CE_MAXL - 1; l >= 0; --l) {
        if (l >= L) continue;
        ce[l] = 0.f;
        const int sh = a.shift[l];
        const int hh = H >> sh, ww = W >> sh;
        if (valid) {
            const float* sp = a.s[l] + (((size_t)b * hh + (py >> sh)) * ww + (px >> sh)) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) A[c] += sp[c];
            float mx = A[0];
#pragma unroll
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, A[c]);
            float se = 0.f, e[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { e[c] = expf(A[c] - mx); se += e[c]; }
            const float lse = mx + logf(se);
            float al = A[0];
#pragma unroll
            for (int c = 1; c < C; ++c) al = (c == lab) ? A[c] : al;
            ce[l] = lse - al;
            const float inv = 1.f / se;
#pragma unroll
            for (int c = 0; c < C; ++c) G[l][c] = (e[c] * inv - (c == lab ? 1.f : 0.f)) * gscale;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) G[l][c] = 0.f;
        }
    }
    // A now holds sum_l s_l at this pixel
    if (valid && s_out) {
#pragma unroll
        for (int c = 0; c < C; ++c) s_out[(((size_t)b * H + py) * W + px) * C + c] = A[c];
    }
    if (valid && sm_out) {
        float mx = A[0];
#pragma unroll
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, A[c]);
        float se = 0.f, e[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { e[c] = expf(A[c] - mx); se += e[c]; }
#pragma unroll
        for (int c = 0; c < C; ++c) sm_out[(((size_t)b * H + py) * W + px) * C + c] = e[c] / se;
    }
    // losses: wave reduce -> one atomic per wave per level, spread over 64 slots
    if (loss_part) {
        __shared__ float wl[4][CE_MAXL];             // the four wave sums meet in LDS: one atomic per level and BLOCK
#pragma unroll
        for (int l = 0; l < CE_MAXL; ++l) {
            if (l >= L) continue;
            const float t = wave_sum(ce[l]);
            if (lane == 0) wl[wave][l] = t;
        }
        __syncthreads();
        if (threadIdx.x < L) {
            const int l = threadIdx.x;
            const float t = (wl[0][l] + wl[1][l]) + (wl[2][l] + wl[3][l]);
            if (det)      // deterministic mode: 32 slots of 2^-20 fixed point (integer adds commute), k_reduce_loss_parts converts
                atomicAdd(reinterpret_cast<unsigned long long*>(loss_part) + (blockIdx.x & 31) * CE_MAXL + l,
                          (unsigned long long)(long long)llrintf(t * inv_batch * 1048576.f));
            else
                atomicAdd(&loss_part[(blockIdx.x & 63) * CE_MAXL + l], t * inv_batch);
        }
    }
    // gradients: d/ds_k = sum_{l<=k} G_l (prefix over levels, finest first), then f x f block sum
    if (a.ds[0] || (L > 1 && a.ds[1])) {
        float run[C];
#pragma unroll
        for (int c = 0; c < C; ++c) run[c] = 0.f;
#pragma unroll
        for (int k = 0; k < CE_MAXL; ++k) {
            if (k >= L) continue;
#pragma unroll
            for (int c = 0; c < C; ++c) run[c] += G[k][c];
            if (!a.ds[k]) continue;
            const int sh = a.shift[k];
            const int hh = H >> sh, ww = W >> sh;
            const int rb = sh < 3 ? sh : 3;                     // bits reducible inside the 8x8 wave tile
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float v = run[c];
                for (int bit = 0; bit < rb; ++bit) {
                    v += __shfl_xor(v, 1 << bit, 64);           // x neighbours
                    v += __shfl_xor(v, 8 << bit, 64);           // y neighbours
                }
                const int m = (1 << rb) - 1;
                const bool leader = ((lane & 7) & m) == 0 && ((lane >> 3) & m) == 0;
                if (sh == 4) {
                    // the 16 x 16 block IS this work-group's tile: its four wave sums meet in LDS and are added in a fixed
                    // order (four atomics on one address otherwise: the only unordered sum of this kernel's gradients)
                    __shared__ float wsum[4][C];
                    if (lane == 0) wsum[wave][c] = v;
                    __syncthreads();
                    if (threadIdx.x == 0 && valid)
                        a.ds[k][(((size_t)b * hh + (py >> 4)) * ww + (px >> 4)) * C + c] = (wsum[0][c] + wsum[1][c]) + (wsum[2][c] + wsum[3][c]);
                    __syncthreads();
                } else if (leader && valid) {
                    // (sh <= 3: the f x f block lies inside this wave's 8 x 8 sub-tile and the shuffles above summed ALL of it -- one
                    // writer per element, a plain store; the atomics that used to stand here cost this kernel most of its 88 us)
                    a.ds[k][(((size_t)b * hh + (py >> sh)) * ww + (px >> sh)) * C + c] = v;
                }
            }
        }
    }
}

__global__ void k_reduce_loss_parts(const float* __restrict__ part, int L, float* __restrict__ losses, int det) {
    const int l = threadIdx.x;
    if (l >= L) return;
    if (det) {
        long long a = 0;
        for (int j = 0; j < 32; ++j) a += reinterpret_cast<const long long*>(part)[j * CE_MAXL + l];
        losses[l] = (float)((double)a * (1.0 / 1048576.0));
        return;
    }
    float a = 0.f;
    for (int j = 0; j < 64; ++j) a += part[j * CE_MAXL + l];
    losses[l] = a;
}

// ---- KL of diagonal Gaussians (phiseg_model.py:210-226) + Appendix C gradients ---------------------
__global__ void k_kl(const float* __restrict__ mu0, const float* __restrict__ s0, const float* __restrict__ mu1,
                     const float* __restrict__ s1, size_t n, float lw, float inv_batch, float gscale,
                     float* __restrict__ loss, float* dmu0, float* ds0, float* dmu1, float* ds1) {
    float acc = 0.f;
    const float c = gscale * lw * inv_batch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = s0[i] * s0[i], bq = s1[i] * s1[i] + 1e-10f, d = mu1[i] - mu0[i];
        acc += 0.5f * ((a + d * d) / bq + logf(bq) - logf(a + 1e-10f) - 1.f);
        if (dmu0) {
            dmu0[i] = -c * d / bq;
            dmu1[i] = c * d / bq;
            ds0[i] = c * s0[i] * (1.f / bq - 1.f / (a + 1e-10f));
            ds1[i] = c * s1[i] * (1.f / bq - (a + d * d) / (bq * bq));
        }
    }
    __shared__ float sh[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (sh[0] + sh[1] + sh[2] + sh[3]) * lw * inv_batch);
}

// all levels of the hierarchical KL term (phiseg_model.py:265-287) in ONE launch: block -> level by the running block counts
#define KL_MAXL 8
struct KLArgs {
    const float *mu0[KL_MAXL], *s0[KL_MAXL], *mu1[KL_MAXL], *s1[KL_MAXL];
    float *dmu0[KL_MAXL], *ds0[KL_MAXL], *dmu1[KL_MAXL], *ds1[KL_MAXL];
    float* loss[KL_MAXL];                      // zeroed by the caller (accumulated atomically)
    unsigned long long n[KL_MAXL];
    float lw[KL_MAXL];
    int blk0[KL_MAXL + 1];                     // first block of level l; blk0[L] = grid size
};
__global__ void k_kl_multi(KLArgs a, int L, float inv_batch, float gscale) {
    int l = 0;
#pragma unroll
    for (int q = 1; q < KL_MAXL; ++q)
        if (q < L && (int)blockIdx.x >= a.blk0[q]) l = q;
    const int bx = blockIdx.x - a.blk0[l], nb = a.blk0[l + 1] - a.blk0[l];
    const float *mu0 = a.mu0[l], *s0 = a.s0[l], *mu1 = a.mu1[l], *s1 = a.s1[l];
    float *dmu0 = a.dmu0[l], *ds0 = a.ds0[l], *dmu1 = a.dmu1[l], *ds1 = a.ds1[l];
    const size_t n = (size_t)a.n[l];
    const float lw = a.lw[l];
    float acc = 0.f;
    const float c = gscale * lw * inv_batch;
    for (size_t i = bx * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)nb * blockDim.x) {
        const float aa = s0[i] * s0[i], bq = s1[i] * s1[i] + 1e-10f, d = mu1[i] - mu0[i];
        acc += 0.5f * ((aa + d * d) / bq + logf(bq) - logf(aa + 1e-10f) - 1.f);
        if (dmu0) {
            dmu0[i] = -c * d / bq;
            dmu1[i] = c * d / bq;
            ds0[i] = c * s0[i] * (1.f / bq - 1.f / (aa + 1e-10f));
            ds1[i] = c * s1[i] * (1.f / bq - (aa + d * d) / (bq * bq));
        }
    }
    __shared__ float sh[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.loss[l], (sh[0] + sh[1] + sh[2] + sh[3]) * lw * inv_batch);
}

// ---- weight decay (phiseg_model.py:290-299): weight * sum over the 'weight_variables' collection of tf.nn.l2_loss(W) = sum w^2 / 2.
// The variables live in one flat arena; `mask` (1 for the elements of collection members, 0 elsewhere) selects them, so the
// term is one pass over the arena whatever the number of variables.  Two stages, fixed order: deterministic.
__global__ void k_l2_masked_partial(const float* __restrict__ p, const float* __restrict__ mask, size_t n, float* __restrict__ part) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += mask[i] * p[i] * p[i];
    __shared__ float sh[4];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void k_l2_masked_finish(const float* __restrict__ part, int nb, float scale, float* __restrict__ out) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) a += part[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) *out = 0.5f * scale * a;
}
// g += alpha * mask * p   (the gradient of alpha * l2 term)
__global__ void k_axpy_masked(float* __restrict__ g, const float* __restrict__ p, const float* __restrict__ mask, size_t n, float alpha) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        g[i] = fmaf(alpha * mask[i], p[i], g[i]);
}

// ---- Adam, TF 1.12 form ----------------------------------------------------------------------------
__global__ void k_adam_tf1(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                           float* __restrict__ v, size_t n4, size_t n, const float* __restrict__ lr_dev, float b1,
                           float b2, float eps, const int32_t* __restrict__ step_dev) {
    const float t = (float)(*step_dev + 1);
    const float lr_t = (*lr_dev) * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; const float* Gp = &gg.x; float* M = &mm.x; float* Vv = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            M[j] += (Gp[j] - M[j]) * (1.f - b1);
            Vv[j] += (Gp[j] * Gp[j] - Vv[j]) * (1.f - b2);
            P[j] -= lr_t * M[j] / (sqrtf(Vv[j]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    const size_t i = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        m[i] += (g[i] - m[i]) * (1.f - b1);
        v[i] += (g[i] * g[i] - v[i]) * (1.f - b2);
        p[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
    }
}

__global__ void k_step_increment(int32_t* s) { *s += 1; }
__global__ void k_stamp(unsigned long long* dst) { *dst = wall_clock64(); }      // constant 100 MHz counter
__global__ void k_sum_scalars(const float* in, int n, float* out) {
    float a = 0.f;
    for (int i = 0; i < n; ++i) a += in[i];
    *out = a;
}

struct WSArgs {
    const float* p[16];
    float w[16];
};
__global__ void k_weighted_sum(WSArgs a, int n, float* out) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += a.w[i] * (*a.p[i]);
    *out = acc;
}

extern "C" {

int phx_weighted_sum(const float* const* ptrs, const float* weights, int n, float* out, void* stream) {
    PHX_REQUIRE(n >= 1 && n <= 16, PHX_E_SHAPE, "weighted_sum: 1 <= n <= 16");
    WSArgs a;
    for (int i = 0; i < 16; ++i) { a.p[i] = i < n ? ptrs[i] : nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
    hipLaunchKernelGGL(k_weighted_sum, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, out);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_reparam_fwd(const float* mu, const float* sigma, float* z, int B, int per_sample, uint64_t seed,
                    const int32_t* step_dev, int stream_id, int sample_offset, void* stream) {
    const size_t total = (size_t)B * ((per_sample + 3) / 4);
    hipLaunchKernelGGL(k_reparam_fwd, dim3(phx_grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, mu, sigma, z, B,
                       per_sample, (unsigned long long)seed, step_dev, stream_id, sample_offset);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_reparam_bwd(const float* dz, float* dsigma, int B, int per_sample, uint64_t seed, const int32_t* step_dev,
                    int stream_id, int sample_offset, void* stream) {
    const size_t total = (size_t)B * ((per_sample + 3) / 4);
    hipLaunchKernelGGL(k_reparam_bwd, dim3(phx_grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dz, dsigma, B,
                       per_sample, (unsigned long long)seed, step_dev, stream_id, sample_offset);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_philox_normal(float* out, int B, int per_sample, uint64_t seed, const int32_t* step_dev, int stream_id,
                      int sample_offset, void* stream) {
    return phx_reparam_fwd(nullptr, nullptr, out, B, per_sample, seed, step_dev, stream_id, sample_offset, stream);
}

int phx_residual_ce(const float* const* s, float* const* ds, const int* shift, int L, const uint8_t* labels, int B,
                    int H, int W, int C, float weight, float inv_batch, float* losses, float* s_out, float* sm_out,
                    void* stream) {
    PHX_REQUIRE(L >= 1 && L <= CE_MAXL, PHX_E_SHAPE, "residual_ce: 1 <= L <= 8");
    PHX_REQUIRE(C >= 2 && C <= CE_MAXC, PHX_E_SHAPE, "residual_ce: 2 <= C <= 8");
    CEArgs a;
    for (int l = 0; l < CE_MAXL; ++l) {
        a.s[l] = l < L ? s[l] : nullptr;
        a.ds[l] = (l < L && ds) ? ds[l] : nullptr;
        a.shift[l] = l < L ? shift[l] : 0;
        if (l < L) PHX_REQUIRE((H >> a.shift[l]) << a.shift[l] == H && (W >> a.shift[l]) << a.shift[l] == W &&
                               a.shift[l] <= 4, PHX_E_SHAPE, "residual_ce: level size must divide the image, factor <= 16");
    }
    // scratch for the loss partials lives at losses[CE_MAXL .. CE_MAXL + 64*CE_MAXL): caller provides 8 + 512 floats
    float* part = losses ? losses + CE_MAXL : nullptr;
    if (part) PHX_CHECK_HIP(hipMemsetAsync(part, 0, 64 * CE_MAXL * sizeof(float), (hipStream_t)stream));
    const int tiles = ((W + 15) / 16) * ((H + 15) / 16);
    const float gscale = weight * inv_batch;
    const int det = phx_deterministic() ? 1 : 0;
#define CE_LAUNCH(CC)                                                                                            \
    hipLaunchKernelGGL((k_residual_ce<CC>), dim3(tiles* B), dim3(256), 0, (hipStream_t)stream, a, L, labels, B, H, W, \
                       gscale, inv_batch, part, s_out, sm_out, det)
    switch (C) {
        case 2: CE_LAUNCH(2); break;
        case 3: CE_LAUNCH(3); break;
        case 4: CE_LAUNCH(4); break;
        case 5: CE_LAUNCH(5); break;
        case 6: CE_LAUNCH(6); break;
        case 7: CE_LAUNCH(7); break;
        default: CE_LAUNCH(8); break;
    }
#undef CE_LAUNCH
    PHX_CHECK_LAUNCH();
    if (part) {
        hipLaunchKernelGGL(k_reduce_loss_parts, dim3(1), dim3(64), 0, (hipStream_t)stream, part, L, losses, det);
        PHX_CHECK_LAUNCH();
    }
    return PHX_OK;
}

int phx_kl_diag_gauss(const float* mu0, const float* s0, const float* mu1, const float* s1, size_t n, float level_w,
                      float inv_batch, float grad_scale, float* loss, float* dmu0, float* ds0, float* dmu1, float* ds1,
                      void* stream) {
    PHX_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(k_kl, dim3(phx_deterministic() ? 1 : phx_grid_for(n, 256, 64)), dim3(256), 0, (hipStream_t)stream, mu0, s0, mu1, s1, n,
                       level_w, inv_batch, grad_scale, loss, dmu0, ds0, dmu1, ds1);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

/* every level of the hierarchical KL term in one launch.  ptrs: 9 device pointers per level {mu0, s0, mu1, s1, dmu0, ds0, dmu1, ds1,
 * loss} (the four gradient pointers all NULL or all set), loss[l] += level_w[l] * inv_batch * sum KL -- the loss scalars must be ZERO
 * on entry (the engine keeps them in its per-step zero arena); one block per level in deterministic mode. */
int phx_kl_diag_gauss_multi(const void* const* ptrs, const size_t* n, const float* level_w, int L, float inv_batch, float grad_scale,
                            void* stream) {
    PHX_REQUIRE(L >= 1 && L <= KL_MAXL && ptrs && n && level_w, PHX_E_SHAPE, "kl_multi: 1 <= L <= 8");
    KLArgs a;
    int blk = 0;
    for (int l = 0; l < KL_MAXL; ++l) {
        const void* const* q = ptrs + 9 * (l < L ? l : 0);
        a.mu0[l] = (const float*)q[0]; a.s0[l] = (const float*)q[1]; a.mu1[l] = (const float*)q[2]; a.s1[l] = (const float*)q[3];
        a.dmu0[l] = (float*)q[4]; a.ds0[l] = (float*)q[5]; a.dmu1[l] = (float*)q[6]; a.ds1[l] = (float*)q[7];
        a.loss[l] = (float*)q[8];
        a.n[l] = l < L ? n[l] : 0;
        a.lw[l] = l < L ? level_w[l] : 0.f;
        a.blk0[l] = blk;
        if (l < L) {
            PHX_REQUIRE(q[0] && q[1] && q[2] && q[3] && q[8], PHX_E_INVAL, "kl_multi: null argument");
            blk += phx_deterministic() ? 1 : phx_grid_for(n[l], 256, 64);
        }
    }
    a.blk0[KL_MAXL] = blk;
    for (int l = L; l <= KL_MAXL; ++l) a.blk0[l] = blk;
    hipLaunchKernelGGL(k_kl_multi, dim3(blk), dim3(256), 0, (hipStream_t)stream, a, L, inv_batch, grad_scale);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

/* out = scale * sum_i mask[i] p[i]^2 / 2; work: 256 floats of scratch */
int phx_l2_masked(const float* p, const float* mask, size_t n, float scale, float* work256, float* out, void* stream) {
    PHX_REQUIRE(p && mask && work256 && out, PHX_E_INVAL, "l2_masked: null argument");
    hipLaunchKernelGGL(k_l2_masked_partial, dim3(256), dim3(256), 0, (hipStream_t)stream, p, mask, n, work256);
    hipLaunchKernelGGL(k_l2_masked_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)work256, 256, scale, out);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_axpy_masked(float* g, const float* p, const float* mask, size_t n, float alpha, void* stream) {
    PHX_REQUIRE(g && p && mask, PHX_E_INVAL, "axpy_masked: null argument");
    hipLaunchKernelGGL(k_axpy_masked, dim3(phx_grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, g, p, mask, n, alpha);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_adam_tf1(float* p, const float* g, float* m, float* v, size_t n, const float* lr_dev, float beta1, float beta2,
                 float eps, const int32_t* step_dev, void* stream) {
    PHX_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, PHX_E_ALIGN,
                "adam: arenas must be 16-byte aligned");
    const size_t n4 = n / 4;
    int grid = phx_grid_for(n4 > 0 ? n4 : 1, 256, 8192);
    hipLaunchKernelGGL(k_adam_tf1, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr_dev, beta1, beta2,
                       eps, step_dev);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_stamp(void* dst_u64, void* stream) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)dst_u64);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_step_increment(int32_t* step_dev, void* stream) {
    hipLaunchKernelGGL(k_step_increment, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_sum_scalars(const float* in, int n, float* out, void* stream) {
    hipLaunchKernelGGL(k_sum_scalars, dim3(1), dim3(1), 0, (hipStream_t)stream, in, n, out);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
