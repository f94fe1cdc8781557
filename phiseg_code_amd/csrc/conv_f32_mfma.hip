// fp32 3x3 SAME convolution (tf.nn.conv2d, tfwrapper/layers.py:122-135), its data gradient and its filter gradient on the fp32
// matrix instruction v_mfma_f32_32x32x2_f32: the arithmetic of the fp32 parity path -- the path that is pinned to the reference's
// per-level logits and ELBO at 1e-4 -- on the matrix cores instead of the vector ALU (conv_direct.hip).  The instruction is a
// k-ordered chain of fp32 fused multiply-adds (one rounding per product, no wider accumulator), i.e. the same arithmetic class as
// the direct kernel; only the summation order differs.
//
// Peak: 64 FLOP/clk/SIMD = 157 TFLOP/s, a sixteenth of the bf16 rate, so one 64-cycle instruction pays for ~2 KB of LDS reads:
// these kernels are bound by the matrix pipe alone and keep their staging simple (global -> registers -> LDS, next chunk's loads
// in flight under the running chunk's matrix instructions; two blocks per CU).
//
// Implicit GEMM, im2col-free.  Forward / data gradient (one kernel; the data gradient is the forward kernel on the flipped,
// transposed packed filter): M = 256 output pixels (tb images x th x tw), N = BN output channels, K walked in chunks of 8 channels x
// 9 taps; operand fragments are 16-byte LDS reads (lane (i, h) holds channels 4h .. 4h + 3 of row i: four instructions per read).
// Filter gradient: M = 32 input channels, N = 32 output channels per wave, K = pixels (pairs of horizontally adjacent pixels), nine
// accumulators (one per tap) per wave; blocks are persistent over a slice of the pixel tiles and leave partial filters in a
// workspace, summed in slice order by a second launch (always a fixed order: the same result in every run).
#include "phx_common.h"

namespace {

struct F32Geo {
    int tws, ths;                 // log2 of the tile width / height
    int tb;                       // images per tile
    int tiles_x, tiles_y, tiles_b;
    int pw, ph, npatch;           // patch width, height, tb * ph * pw
    int rows;                     // tb << (tws + ths)
    unsigned mpw, mpp;            // v / pw == (v * mpw) >> 20, v / (ph * pw) == (v * mpp) >> 20 for v < 2048
};

static unsigned magic20(int d) {
    unsigned m = ((1u << 20) + d - 1) / d;
    for (unsigned v = 0; v < 2048; ++v)
        if (((v * m) >> 20) != v / (unsigned)d) return 0;
    return m;
}

// forward / data gradient: 256-row tiles
static bool make_geo_fwd(int B, int H, int W, F32Geo* g) {
    int tw = 1, th = 1;
    g->tws = g->ths = 0;
    while (tw < W && tw < 16) { tw <<= 1; g->tws++; }
    while (th < H && th < 16) { th <<= 1; g->ths++; }
    g->tb = 256 / (tw * th);
    g->tiles_x = (W + tw - 1) / tw;
    g->tiles_y = (H + th - 1) / th;
    g->tiles_b = (B + g->tb - 1) / g->tb;
    g->pw = tw + 2; g->ph = th + 2;
    g->npatch = g->tb * g->ph * g->pw;
    g->rows = 256;
    g->mpw = magic20(g->pw); g->mpp = magic20(g->ph * g->pw);
    return g->mpw && g->mpp && g->npatch <= 1024;
}

// forward / data gradient on few-tile problems: 64-row tiles (8 x 8 pixels, or whole small maps of several images)
static bool make_geo_small(int B, int H, int W, F32Geo* g) {
    int tw = 1, th = 1;
    g->tws = g->ths = 0;
    while (tw < W && tw < 8) { tw <<= 1; g->tws++; }
    while (th < H && th < 8) { th <<= 1; g->ths++; }
    g->tb = 64 / (tw * th);
    g->tiles_x = (W + tw - 1) / tw;
    g->tiles_y = (H + th - 1) / th;
    g->tiles_b = (B + g->tb - 1) / g->tb;
    g->pw = tw + 2; g->ph = th + 2;
    g->npatch = g->tb * g->ph * g->pw;
    g->rows = 64;
    g->mpw = magic20(g->pw); g->mpp = magic20(g->ph * g->pw);
    return g->mpw && g->mpp && g->npatch <= 256;
}

// filter gradient: tiles of at most 128 rows and at most 180 patch pixels (two stages of a 64 x 64-channel block: 157.7 KB of LDS)
static bool make_geo_wgrad(int B, int H, int W, F32Geo* g) {
    int tw = 1, th = 1;
    g->tws = g->ths = 0;
    while (tw < W && tw < 16) { tw <<= 1; g->tws++; }
    while (th < H && th < 8) { th <<= 1; g->ths++; }
    g->pw = tw + 2; g->ph = th + 2;
    int tb = 128 / (tw * th);
    if (tb > 180 / (g->ph * g->pw)) tb = 180 / (g->ph * g->pw);
    if (tb < 1) tb = 1;
    if (tb > B) tb = B;
    g->tb = tb;
    g->tiles_x = (W + tw - 1) / tw;
    g->tiles_y = (H + th - 1) / th;
    g->tiles_b = (B + tb - 1) / tb;
    g->npatch = tb * g->ph * g->pw;
    g->rows = tb * tw * th;
    g->mpw = magic20(g->pw); g->mpp = magic20(g->ph * g->pw);
    return g->mpw && g->mpp && (g->rows & 1) == 0;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---- packed filters ---------------------------------------------------------------------------------------------------------
// wpk[K8 / 8][9][N][8]: element (tap t, row n, reduction channel k) at (((k / 8) * 9 + t) * N + n) * 8 + k % 8, K8 = K rounded up to
// a multiple of 8 (zero-filled).  forward: N = Cout, K = Cin, t = kh * 3 + kw; data gradient: N = Cin, K = Cout, t = (2 - kh) * 3 + (2 - kw).
struct PackF32Job {
    const float* w;
    float* wf;
    float* wd;
    int cin, cout;
};

__global__ __launch_bounds__(256) void k_pack_conv3x3_f32_multi(const PackF32Job* __restrict__ jobs) {
    const PackF32Job j = jobs[blockIdx.y];
    const int cin = j.cin, cout = j.cout;
    const int k8f = (cin + 7) & ~7, k8d = (cout + 7) & ~7;
    const size_t nf = (size_t)k8f * 9 * cout, nd = j.wd ? (size_t)k8d * 9 * cin : 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nf + nd; e += (size_t)gridDim.x * 256) {
        if (e < nf) {
            const int kk = (int)(e & 7);
            size_t r = e >> 3;
            const int n = (int)(r % cout); r /= cout;
            const int t = (int)(r % 9);
            const int k = (int)(r / 9) * 8 + kk;
            j.wf[e] = k < cin ? j.w[((size_t)t * cin + k) * cout + n] : 0.f;
        } else {
            const size_t d = e - nf;
            const int kk = (int)(d & 7);
            size_t r = d >> 3;
            const int n = (int)(r % cin); r /= cin;
            const int t = (int)(r % 9);
            const int k = (int)(r / 9) * 8 + kk;
            j.wd[d] = k < cout ? j.w[((size_t)(8 - t) * cin + n) * cout + k] : 0.f;
        }
    }
}

// ---- forward / data gradient -----------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256, 2) void k_conv3x3_f32_mfma(const float* __restrict__ x, const float* __restrict__ wpk,
                                                              const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                              int K, int N, int act, F32Geo g) {
    constexpr int NT = BN / 32;
    constexpr int NWP = (9 * BN * 2 + 255) / 256;        // 16-byte filter pieces per thread and chunk
    extern __shared__ float smem[];
    const int npatch = g.npatch;
    float* sp = smem;                        // [2][npatch][4]: plane h holds channels 4h .. 4h + 3 of the chunk
    float* sw = smem + 8 * npatch;           // [9][2][BN][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int tw = 1 << g.tws, th = 1 << g.ths, pw = g.pw, ph = g.ph;
    int t = blockIdx.x;
    const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
    const int b0 = t * g.tb;
    const int n0 = blockIdx.y * BN;

    int pbase[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = 64 * wv + 32 * mi + l31;
        const int lx = r & (tw - 1), ly = (r >> g.tws) & (th - 1), bi = r >> (g.tws + g.ths);
        pbase[mi] = (bi * ph + ly) * pw + lx;
    }
    // this thread's pieces of the patch: fixed over the chunks
    int goff[8];
    const int np2 = npatch * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j;
        goff[j] = -1;
        if (i < np2) {
            const unsigned pp = (unsigned)i >> 1;
            const int bi = (int)((pp * g.mpp) >> 20);
            const unsigned rem = pp - bi * ph * pw;
            const int py = (int)((rem * g.mpw) >> 20);
            const int px = (int)rem - py * pw;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1, gb = b0 + bi;
            if (gb < B && gy >= 0 && gy < H && gx >= 0 && gx < W) goff[j] = ((gb * H + gy) * W + gx) * K;
        }
    }
    const bool kvec = (K & 3) == 0;
    const int nkc = (K + 7) >> 3;
    f32x4 rp[8], rw[NWP];

    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < np2 && goff[j] >= 0) {
                const int c = kc * 8 + (i & 1) * 4;
                const float* p = x + (size_t)goff[j] + c;
                if (kvec) {
                    if (c < K) v = *(const f32x4*)p;
                } else {
                    if (c + 0 < K) v[0] = p[0];
                    if (c + 1 < K) v[1] = p[1];
                    if (c + 2 < K) v[2] = p[2];
                    if (c + 3 < K) v[3] = p[3];
                }
            }
            rp[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int i = tid + 256 * j;
            if (i < 9 * BN * 2) {
                const int t9 = i / (BN * 2), rem = i % (BN * 2);
                rw[j] = *(const f32x4*)(wpk + (((size_t)kc * 9 + t9) * N + n0 + (rem >> 1)) * 8 + (rem & 1) * 4);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j;
            if (i < np2) *(f32x4*)(sp + ((size_t)(i & 1) * npatch + (i >> 1)) * 4) = rp[j];
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int i = tid + 256 * j;
            if (i < 9 * BN * 2) {
                const int t9 = i / (BN * 2), rem = i % (BN * 2);
                *(f32x4*)(sw + ((t9 * 2 + (rem & 1)) * BN + (rem >> 1)) * 4) = rw[j];
            }
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc + 1 < nkc) load_chunk(kc + 1);            // in flight under this chunk's matrix instructions
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const int toff = (t9 / 3) * pw + (t9 % 3);
            f32x4 a[2], b[NT];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f32x4*)(sp + ((size_t)hh * npatch + pbase[mi] + toff) * 4);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) b[ni] = *(const f32x4*)(sw + ((t9 * 2 + hh) * BN + ni * 32 + l31) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma32(a[mi][j], b[ni][j], acc[mi][ni]);
        }
        __syncthreads();
        if (kc + 1 < nkc) {
            store_chunk();
            __syncthreads();
        }
    }
    // epilogue: lane holds column l31 of rows (e & 3) + 8 (e >> 2) + 4 hh of each 32 x 32 block
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int n = n0 + ni * 32 + l31;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = 64 * wv + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int ox = tx0 + (r & (tw - 1)), oy = ty0 + ((r >> g.tws) & (th - 1)), ob = b0 + (r >> (g.tws + g.ths));
                if (ox < W && oy < H && ob < B) y[((size_t)(ob * H + oy) * W + ox) * N + n] = act_fwd(acc[mi][ni][e] + bv, act);
            }
    }
}

// The same forward / data-gradient launch with LDS-DMA staging into two LDS stages and LOADER WAVES (as the filter gradient below):
// 512 threads, waves 0-3 compute (one per SIMD, 64 rows x BN channels each), waves 4-7 issue the next chunk's buffer_load ... lds.
// LDS images are lane-linear: patch [pixel][2 halves][4 channels], filter slab [tap][row][2 halves][4] -- exactly the order of the packed
// filter in memory.  Needs K % 4 == 0 (16-byte pieces); the register-staged kernel above takes the narrow inputs.
template <int BN>
__global__ __launch_bounds__(512, 2) void k_conv3x3_f32_mfma_dma(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                                  int K, int N, int act, F32Geo g) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int npatch = g.npatch;
    const int nxp = npatch * 2, nwp = 9 * BN * 2;                  // 16-byte pieces of the patch / of the filter slab
    const int nxi = (nxp + 63) >> 6, nwi = (nwp + 63) >> 6;
    const int stage_floats = (nxi + nwi) * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool loader = tid >= 256;
    const int wv = (tid >> 6) & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    const int tw = 1 << g.tws, th = 1 << g.ths, pw = g.pw, ph = g.ph;
    int t = blockIdx.x;
    const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
    const int b0 = t * g.tb;
    const int n0 = blockIdx.y * BN;
    const int nkc = (K + 7) >> 3;

    // One barrier per chunk, S_k = "chunk k has landed AND chunk k - 1 is consumed":
    //   loader waves : issue(0); for k: { wait for their own DMA; S_k; issue(k + 1) into the stage chunk k - 1 used }
    //   compute waves: for k: { S_k; matrix instructions of chunk k }
    if (loader) {
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * K * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, (int)((unsigned)nkc * 9 * N * 32u), 0x00020000);
        // this lane's pieces of the patch are the same pixels in every chunk: byte offset of (pixel, half) at channel 0, or out of range
        unsigned poff[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = (wv + 4 * q) * 64 + lane;
            const unsigned pp = (unsigned)e >> 1;
            const int bi = (int)((pp * g.mpp) >> 20);
            const unsigned rem = pp - bi * ph * pw;
            const int py = (int)((rem * g.mpw) >> 20);
            const int px = (int)rem - py * pw;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1, gb = b0 + bi;
            const bool ok = e < nxp && gb < B && gy >= 0 && gy < H && gx >= 0 && gx < W;
            poff[q] = ok ? (unsigned)(((gb * H + gy) * W + gx) * K + (e & 1) * 4) * 4u : 0xffffffffu;
        }
        auto issue = [&](int kc) {
            char* base = (char*)(smem + (size_t)(kc & 1) * stage_floats);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = wv + 4 * q;
                if (j < nxi) {
                    const int c = kc * 8 + (lane & 1) * 4;             // ((j * 64 + lane) & 1) == (lane & 1)
                    const unsigned vo = (poff[q] != 0xffffffffu && c < K) ? poff[q] + (unsigned)(kc * 32) : 0xffffffffu;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(base + j * 1024), 16, vo, 0, 0, 0);
                }
            }
            for (int j = wv; j < nwi; j += 4) {
                const int e = j * 64 + lane;                           // ((t9 * BN + n) * 2 + h)
                const int t9 = e / (BN * 2), r = e % (BN * 2);
                const unsigned vo = e < nwp ? (unsigned)(((kc * 9 + t9) * N + n0) * 8 + r * 4) * 4u : 0xffffffffu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(base + (nxi + j) * 1024), 16, vo, 0, 0, 0);
            }
        };
        issue(0);
        for (int kc = 0; kc < nkc; ++kc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kc + 1 < nkc) issue(kc + 1);
        }
        return;
    }

    int pbase[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = 64 * wv + 32 * mi + l31;
        const int lx = r & (tw - 1), ly = (r >> g.tws) & (th - 1), bi = r >> (g.tws + g.ths);
        pbase[mi] = (bi * ph + ly) * pw + lx;
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();                                               // S_kc: chunk kc has landed, chunk kc - 1 is consumed by every wave
        const float* sp = smem + (size_t)(kc & 1) * stage_floats;
        const float* sw = sp + (size_t)nxi * 256;
        // one wave per SIMD: the fragments of tap t + 1 are read under the sixteen matrix instructions of tap t (pinned: nothing else hides the LDS latency)
        f32x4 a[2][2], b[2][NT];
        auto frags = [&](int t9, f32x4 (&aa)[2], f32x4 (&bb)[NT]) {
            const int toff = (t9 / 3) * pw + (t9 % 3);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) aa[mi] = *(const f32x4*)(sp + ((size_t)(pbase[mi] + toff) * 2 + hh) * 4);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) bb[ni] = *(const f32x4*)(sw + ((size_t)(t9 * BN + ni * 32 + l31) * 2 + hh) * 4);
        };
        frags(0, a[0], b[0]);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            if (t9 + 1 < 9) frags(t9 + 1, a[(t9 + 1) & 1], b[(t9 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma32(a[t9 & 1][mi][j], b[t9 & 1][ni][j], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int n = n0 + ni * 32 + l31;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = 64 * wv + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int ox = tx0 + (r & (tw - 1)), oy = ty0 + ((r >> g.tws) & (th - 1)), ob = b0 + (r >> (g.tws + g.ths));
                if (ox < W && oy < H && ob < B) y[((size_t)(ob * H + oy) * W + ox) * N + n] = act_fwd(acc[mi][ni][e] + bv, act);
            }
    }
}

// ---- forward / data gradient, few-tile problems -------------------------------------------------------------------------------
// The 256-row kernel above leaves most of the chip idle on the H <= 16 levels (a 192 -> 192 layer on 8 x 8 maps at batch 64 is 48
// blocks of 221 K matrix cycles each).  Here a block owns 64 rows x 32 output channels and its four waves SPLIT THE REDUCTION: of
// every group of 32 reduction channels wave w takes channels [8w, 8w + 8); the four accumulator sets are summed through LDS in wave
// order at the end (a fixed order: bit-identical from run to run).  6x ... 8x the blocks, a quarter of the serial chain each.
__global__ __launch_bounds__(256, 2) void k_conv3x3_f32_mfma_small(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                                    int W, int K, int N, int act, F32Geo g) {
    extern __shared__ float smem[];
    const int npatch = g.npatch;
    float* sp = smem;                            // [4 waves][2 planes][npatch][4]
    float* sw = smem + 32 * npatch;              // [4 waves][9][2][32][4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int tw = 1 << g.tws, th = 1 << g.ths, pw = g.pw, ph = g.ph;
    int t = blockIdx.x;
    const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
    const int b0 = t * g.tb;
    const int n0 = blockIdx.y * 32;

    int pbase[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = 32 * mi + l31;
        const int lx = r & (tw - 1), ly = (r >> g.tws) & (th - 1), bi = r >> (g.tws + g.ths);
        pbase[mi] = (bi * ph + ly) * pw + lx;
    }
    int goff[8];
    const int np8 = npatch * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j;
        goff[j] = -1;
        if (i < np8) {
            const unsigned pp = (unsigned)i >> 3;
            const int bi = (int)((pp * g.mpp) >> 20);
            const unsigned rem = pp - bi * ph * pw;
            const int py = (int)((rem * g.mpw) >> 20);
            const int px = (int)rem - py * pw;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1, gb = b0 + bi;
            if (gb < B && gy >= 0 && gy < H && gx >= 0 && gx < W) goff[j] = ((gb * H + gy) * W + gx) * K;
        }
    }
    const bool kvec = (K & 3) == 0;
    const int nkc = (K + 7) >> 3, ngrp = (K + 31) >> 5;
    f32x4 rp[8], rw[9];

    auto load_group = [&](int kg) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < np8 && goff[j] >= 0) {
                const int c = kg * 32 + (i & 7) * 4;
                const float* p = x + (size_t)goff[j] + c;
                if (kvec) {
                    if (c < K) v = *(const f32x4*)p;
                } else {
                    if (c + 0 < K) v[0] = p[0];
                    if (c + 1 < K) v[1] = p[1];
                    if (c + 2 < K) v[2] = p[2];
                    if (c + 3 < K) v[3] = p[3];
                }
            }
            rp[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {            // 4 chunks x 9 taps x 32 rows x 2 halves = 2304 pieces
            const int i = tid + 256 * j;
            const int w4 = i / 576, r = i % 576, t9 = r >> 6, q = r & 63;
            const int kc = kg * 4 + w4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kc < nkc) v = *(const f32x4*)(wpk + (((size_t)kc * 9 + t9) * N + n0 + (q >> 1)) * 8 + (q & 1) * 4);
            rw[j] = v;
        }
    };
    auto store_group = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j;
            if (i < np8) {
                const int q = i & 7;
                *(f32x4*)(sp + ((size_t)((q >> 1) * 2 + (q & 1)) * npatch + (i >> 3)) * 4) = rp[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int i = tid + 256 * j;
            const int w4 = i / 576, r = i % 576, t9 = r >> 6, q = r & 63;
            *(f32x4*)(sw + (((w4 * 9 + t9) * 2 + (q & 1)) * 32 + (q >> 1)) * 4) = rw[j];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
    const float* spw = sp + (size_t)wv * 2 * npatch * 4;
    const float* sww = sw + (size_t)wv * 9 * 2 * 32 * 4;

    load_group(0);
    store_group();
    __syncthreads();
    for (int kg = 0; kg < ngrp; ++kg) {
        if (kg + 1 < ngrp) load_group(kg + 1);
        if (kg * 4 + wv < nkc) {
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int toff = (t9 / 3) * pw + (t9 % 3);
                f32x4 a[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f32x4*)(spw + ((size_t)hh * npatch + pbase[mi] + toff) * 4);
                const f32x4 b = *(const f32x4*)(sww + ((t9 * 2 + hh) * 32 + l31) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[mi] = mfma32(a[mi][j], b[j], acc[mi]);
            }
        }
        __syncthreads();
        if (kg + 1 < ngrp) {
            store_group();
            __syncthreads();
        }
    }
    // the four partial tiles -> LDS [wave][mi * 16 + e][lane], summed in wave order
    float* sr = smem;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) sr[(size_t)(wv * 32 + mi * 16 + e) * 64 + lane] = acc[mi][e];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, me = q >> 6, ln = q & 63;
        const float v = ((sr[(size_t)me * 64 + ln] + sr[(size_t)(32 + me) * 64 + ln]) + sr[(size_t)(64 + me) * 64 + ln]) + sr[(size_t)(96 + me) * 64 + ln];
        const int e = me & 15, mi = me >> 4;
        const int r = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
        const int n = n0 + (ln & 31);
        const int ox = tx0 + (r & (tw - 1)), oy = ty0 + ((r >> g.tws) & (th - 1)), ob = b0 + (r >> (g.tws + g.ths));
        if (ox < W && oy < H && ob < B) y[((size_t)(ob * H + oy) * W + ox) * N + n] = act_fwd(v + (bias ? bias[n] : 0.f), act);
    }
}

// ---- filter gradient --------------------------------------------------------------------------------------------------------
// Block = CIW x COW sub-blocks of 32 x 32 (ci x co), PS = 4 / (CIW COW) waves share a sub-block and split the tile's pixel pairs.
// ws[slice][9][Cin][Cout] (every element written exactly once), wsb[slice][Cout] column sums of dy (bias gradient).
template <int CIW, int COW>
__global__ __launch_bounds__(256, 2) void k_conv3x3_f32_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ ws, float* __restrict__ wsb, int B, int H, int W,
                                                               int Cin, int Cout, int ciblocks, int tps, int ntiles, F32Geo g) {
    constexpr int PS = 4 / (CIW * COW), CIB = 32 * CIW, COB = 32 * COW;
    extern __shared__ float smem[];
    const int npatch = g.npatch, rows = g.rows;
    float* sx = smem;                        // [npatch][CIB]
    float* sd = smem + (size_t)npatch * CIB; // [rows][COB]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ps = wv / (CIW * COW), cis = (wv % (CIW * COW)) % CIW, cos = (wv % (CIW * COW)) / CIW;
    const int ci0 = (blockIdx.y % ciblocks) * CIB, co0 = (blockIdx.y / ciblocks) * COB;
    const int tw = 1 << g.tws, th = 1 << g.ths, pw = g.pw, ph = g.ph;
    const int slice = blockIdx.x;
    const int t_lo = slice * tps, t_hi = min(ntiles, t_lo + tps);
    const bool cvec = (Cin & 3) == 0;

    f32x16 acc[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t9][e] = 0.f;
    float dbacc = 0.f;
    const bool do_bias = wsb != nullptr && (blockIdx.y % ciblocks) == 0;
    const int npairs = rows >> 1;
    const int j_lo = ps * npairs / PS, j_hi = (ps + 1) * npairs / PS;

    for (int tile = t_lo; tile < t_hi; ++tile) {
        int t = tile;
        const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
        const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
        const int b0 = t * g.tb;
        __syncthreads();
        for (int i = tid; i < npatch * (CIB / 4); i += 256) {
            const unsigned pp = (unsigned)i / (CIB / 4);
            const int c = ci0 + (i % (CIB / 4)) * 4;
            const int bi = (int)((pp * g.mpp) >> 20);
            const unsigned rem = pp - bi * ph * pw;
            const int py = (int)((rem * g.mpw) >> 20);
            const int px = (int)rem - py * pw;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1, gb = b0 + bi;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gb < B && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const float* p = x + ((size_t)(gb * H + gy) * W + gx) * Cin + c;
                if (cvec) {
                    if (c < Cin) v = *(const f32x4*)p;
                } else {
                    if (c + 0 < Cin) v[0] = p[0];
                    if (c + 1 < Cin) v[1] = p[1];
                    if (c + 2 < Cin) v[2] = p[2];
                    if (c + 3 < Cin) v[3] = p[3];
                }
            }
            *(f32x4*)(sx + (size_t)i * 4) = v;
        }
        for (int i = tid; i < rows * (COB / 4); i += 256) {
            const int r = i / (COB / 4);
            const int c = co0 + (i % (COB / 4)) * 4;
            const int ox = tx0 + (r & (tw - 1)), oy = ty0 + ((r >> g.tws) & (th - 1)), ob = b0 + (r >> (g.tws + g.ths));
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ox < W && oy < H && ob < B && c < Cout) v = *(const f32x4*)(dy + ((size_t)(ob * H + oy) * W + ox) * Cout + c);
            *(f32x4*)(sd + (size_t)i * 4) = v;
        }
        __syncthreads();
        for (int j = j_lo; j < j_hi; ++j) {
            const int r = 2 * j + hh;
            const int lx = r & (tw - 1), ly = (r >> g.tws) & (th - 1), bi = r >> (g.tws + g.ths);
            const float* pa = sx + (size_t)((bi * ph + ly) * pw + lx) * CIB + cis * 32 + l31;
            const float bv = sd[(size_t)r * COB + cos * 32 + l31];
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) acc[t9] = mfma32(pa[(size_t)((t9 / 3) * pw + (t9 % 3)) * CIB], bv, acc[t9]);
        }
        if (do_bias && tid < COB)
            for (int r = 0; r < rows; ++r) dbacc += sd[(size_t)r * COB + tid];
    }
    if (PS > 1) {
        // the PS waves of a sub-block hold partial sums over disjoint pixel pairs: added in wave order through LDS, tap by tap
        float* sr = smem;                    // [PS - 1][CIW * COW][16][64]
        const int sub = wv % (CIW * COW);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            __syncthreads();
            if (ps > 0)
#pragma unroll
                for (int e = 0; e < 16; ++e) sr[(size_t)(((ps - 1) * CIW * COW + sub) * 16 + e) * 64 + lane] = acc[t9][e];
            __syncthreads();
            if (ps == 0)
                for (int p = 0; p < PS - 1; ++p)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t9][e] += sr[(size_t)((p * CIW * COW + sub) * 16 + e) * 64 + lane];
        }
    }
    float* wo = ws + (size_t)slice * 9 * Cin * Cout;
    const int co = co0 + cos * 32 + l31;
    if (ps == 0) {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + cis * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (ci < Cin && co < Cout) wo[((size_t)t9 * Cin + ci) * Cout + co] = acc[t9][e];
            }
    }
    if (do_bias && tid < COB && co0 + tid < Cout) wsb[(size_t)slice * Cout + co0 + tid] = dbacc;
}

// The same filter gradient with the tiles staged by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass) into TWO
// LDS stages: the next tile's patch and dy rows land while the running tile's matrix instructions issue -- one block per CU, its four
// waves one per SIMD, the matrix pipe never waits for a load phase.  Lane-linear destination: lane l of wave-instruction j fills the
// 16-byte slot 64 j + l, i.e. piece (64 j + l) % Q of pixel (64 j + l) / Q with Q = channels / 4 pieces per pixel -- exactly the
// [pixel][channel] image the operand reads want; out-of-image pieces (and channels beyond Cin / Cout) get offset 0xffffffff, which
// the buffer's range check turns into zeros.  Needs Cin % 4 == 0 (16-byte pieces); the register-staged kernel above takes the rest.
template <int CIW, int COW>
__global__ __launch_bounds__(512, 2) void k_conv3x3_f32_wgrad_dma(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ ws, float* __restrict__ wsb, int B, int H, int W,
                                                                   int Cin, int Cout, int ciblocks, int tps, int ntiles, F32Geo g) {
    // 512 threads: waves 0-3 compute (one per SIMD), waves 4-7 only issue the DMA of the next tile (their address arithmetic -- ~40
    // integer instructions per 1 KB piece row -- fills the issue slots under the matrix instructions of the wave they share a SIMD with)
    constexpr int PS = 4 / (CIW * COW), CIB = 32 * CIW, COB = 32 * COW, QX = CIB / 4, QD = COB / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int npatch = g.npatch, rows = g.rows;
    const int nx = npatch * QX, nd = rows * QD;                     // 16-byte pieces of the patch / of the dy tile
    const int nxi = (nx + 63) >> 6, ndi = (nd + 63) >> 6;           // wave-instructions
    const int stage_floats = (nxi + ndi) * 256;                     // (each instruction fills 1 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const bool loader = tid >= 256;
    const int wv = (tid >> 6) & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ps = wv / (CIW * COW), cis = (wv % (CIW * COW)) % CIW, cos = (wv % (CIW * COW)) / CIW;
    const int ci0 = (blockIdx.y % ciblocks) * CIB, co0 = (blockIdx.y / ciblocks) * COB;
    const int tw = 1 << g.tws, th = 1 << g.ths, pw = g.pw, ph = g.ph;
    const int slice = blockIdx.x;
    const int t_lo = slice * tps, t_hi = min(ntiles, t_lo + tps);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * H * W * Cout * 4u), 0x00020000);

    auto issue = [&](int tile, int stage) {
        int t = tile;
        const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
        const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
        const int b0 = t * g.tb;
        char* base = (char*)(smem + (size_t)stage * stage_floats);
        for (int j = wv; j < nxi; j += 4) {
            const int e = j * 64 + lane;
            const unsigned pp = (unsigned)e / QX;
            const int c = ci0 + (e % QX) * 4;
            const int bi = (int)((pp * g.mpp) >> 20);
            const unsigned rem = pp - bi * ph * pw;
            const int py = (int)((rem * g.mpw) >> 20);
            const int px = (int)rem - py * pw;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1, gb = b0 + bi;
            const bool ok = e < nx && gb < B && gy >= 0 && gy < H && gx >= 0 && gx < W && c < Cin;
            const unsigned vo = ok ? (unsigned)(((gb * H + gy) * W + gx) * Cin + c) * 4u : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(base + j * 1024), 16, vo, 0, 0, 0);
        }
        for (int j = wv; j < ndi; j += 4) {
            const int e = j * 64 + lane;
            const int r = e / QD;
            const int c = co0 + (e % QD) * 4;
            const int ox = tx0 + (r & (tw - 1)), oy = ty0 + ((r >> g.tws) & (th - 1)), ob = b0 + (r >> (g.tws + g.ths));
            const bool ok = e < nd && ox < W && oy < H && ob < B && c < Cout;
            const unsigned vo = ok ? (unsigned)(((ob * H + oy) * W + ox) * Cout + c) * 4u : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lds_ptr_t)(base + (nxi + j) * 1024), 16, vo, 0, 0, 0);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t9][e] = 0.f;
    float dbacc = 0.f;
    const bool do_bias = wsb != nullptr && (blockIdx.y % ciblocks) == 0;
    const int npairs = rows >> 1;
    const int j_lo = ps * npairs / PS, j_hi = (ps + 1) * npairs / PS;

    if (loader && t_lo < t_hi) issue(t_lo, 0);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int stage = (tile - t_lo) & 1;
        if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the loader waves' pieces of the running tile have landed ...
        __syncthreads();                                                   // ... for everybody; the other stage's tile is consumed
        if (loader) {
            if (tile + 1 < t_hi) issue(tile + 1, stage ^ 1);
            continue;
        }
        const float* sx = smem + (size_t)stage * stage_floats;
        const float* sd = sx + (size_t)nxi * 256;
        auto operands = [&](int j, float (&a)[9], float& bv) {
            const int r = 2 * j + hh;
            const int lx = r & (tw - 1), ly = (r >> g.tws) & (th - 1), bi = r >> (g.tws + g.ths);
            const float* pa = sx + (size_t)((bi * ph + ly) * pw + lx) * CIB + cis * 32 + l31;
            bv = sd[(size_t)r * COB + cos * 32 + l31];
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) a[t9] = pa[(size_t)((t9 / 3) * pw + (t9 % 3)) * CIB];
        };
        float a0[9], b0v, a1[9], b1v;
        if (j_lo < j_hi) operands(j_lo, a0, b0v);
        for (int j = j_lo; j < j_hi; j += 2) {                       // operands of the next pair are read under this pair's instructions
            if (j + 1 < j_hi) operands(j + 1, a1, b1v);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) acc[t9] = mfma32(a0[t9], b0v, acc[t9]);
            if (j + 1 < j_hi) {
                if (j + 2 < j_hi) operands(j + 2, a0, b0v);
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) acc[t9] = mfma32(a1[t9], b1v, acc[t9]);
            }
        }
        if (do_bias && tid < COB)
            for (int r = 0; r < rows; ++r) dbacc += sd[(size_t)r * COB + tid];
    }
    if (PS > 1) {
        float* sr = smem;
        const int sub = wv % (CIW * COW);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            __syncthreads();
            if (!loader && ps > 0)
#pragma unroll
                for (int e = 0; e < 16; ++e) sr[(size_t)(((ps - 1) * CIW * COW + sub) * 16 + e) * 64 + lane] = acc[t9][e];
            __syncthreads();
            if (!loader && ps == 0)
                for (int p = 0; p < PS - 1; ++p)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t9][e] += sr[(size_t)((p * CIW * COW + sub) * 16 + e) * 64 + lane];
        }
    }
    if (loader) return;
    float* wo = ws + (size_t)slice * 9 * Cin * Cout;
    const int co = co0 + cos * 32 + l31;
    if (ps == 0) {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + cis * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (ci < Cin && co < Cout) wo[((size_t)t9 * Cin + ci) * Cout + co] = acc[t9][e];
            }
    }
    if (do_bias && tid < COB && co0 + tid < Cout) wsb[(size_t)slice * Cout + co0 + tid] = dbacc;
}

// dst[e] += sum over the slices: a block takes 64 consecutive elements, its four waves every fourth slice each; the four partial
// sums are added in wave order (a fixed order)
__global__ __launch_bounds__(256) void k_f32_wgrad_reduce(const float* __restrict__ ws, int nslice, size_t n, float* __restrict__ dst,
                                                          const float* __restrict__ wsb, int nb, float* __restrict__ db) {
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + el;
    const size_t nn = n + (db != nullptr ? (size_t)nb : 0);
    float s = 0.f;
    if (e < nn) {
        const float* src = e < n ? ws + e : wsb + (e - n);
        const size_t stride = e < n ? n : (size_t)nb;
        int k = q;
        float s1 = 0.f;
        for (; k + 4 < nslice; k += 8) {
            s += src[(size_t)k * stride];
            s1 += src[(size_t)(k + 4) * stride];
        }
        if (k < nslice) s += src[(size_t)k * stride];
        s += s1;
    }
    part[q][el] = s;
    __syncthreads();
    if (q == 0 && e < nn) {
        const float v = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
        if (e < n) dst[e] += v;
        else db[e - n] += v;
    }
}

struct WgradPlan {
    F32Geo g;
    int ciw, cow, ps, ciblocks, coblocks, ntiles, tps, nslice;
    bool dma;
    size_t lds, ws_floats, wsb_floats;
};

static bool plan_wgrad(int B, int H, int W, int Cin, int Cout, bool with_bias, WgradPlan* p) {
    if (!make_geo_wgrad(B, H, W, &p->g)) return false;
    p->ciw = Cin > 32 ? 2 : 1;
    p->cow = Cout > 32 ? 2 : 1;
    p->ps = 4 / (p->ciw * p->cow);
    p->ciblocks = (Cin + 32 * p->ciw - 1) / (32 * p->ciw);
    p->coblocks = (Cout + 32 * p->cow - 1) / (32 * p->cow);
    p->ntiles = p->g.tiles_x * p->g.tiles_y * p->g.tiles_b;
    // LDS-DMA form: 16-byte pieces (Cin % 4 == 0), 32-bit byte offsets, two stages within 160 KB
    const size_t stage = (size_t)(((p->g.npatch * 8 * p->ciw + 63) >> 6) + ((p->g.rows * 8 * p->cow + 63) >> 6)) * 1024;
    p->dma = Cin % 4 == 0 && (double)B * H * W * (Cin > Cout ? Cin : Cout) * 4.0 < 4294967296.0 && 2 * stage <= 160 * 1024;
    int want = (p->dma ? 512 : 1024) / (p->ciblocks * p->coblocks);     // two rounds of resident blocks (DMA form: one block per CU)
    if (want < 1) want = 1;
    if (want > 256) want = 256;                              // bound the workspace (256 partial filters)
    if (want > p->ntiles) want = p->ntiles;
    p->tps = (p->ntiles + want - 1) / want;
    p->nslice = (p->ntiles + p->tps - 1) / p->tps;
    p->lds = ((size_t)p->g.npatch * 32 * p->ciw + (size_t)p->g.rows * 32 * p->cow) * 4;
    if (p->dma) p->lds = 2 * stage;
    if (p->lds < (size_t)(p->ps - 1) * p->ciw * p->cow * 4096) p->lds = (size_t)(p->ps - 1) * p->ciw * p->cow * 4096;
    p->ws_floats = (size_t)p->nslice * 9 * Cin * Cout;
    p->wsb_floats = with_bias ? (size_t)p->nslice * Cout : 0;
    return p->lds <= 160 * 1024;
}

}  // namespace

extern "C" {

int phx_conv3x3_f32_mfma_supported(int B, int H, int W, int K, int N) {
    F32Geo g;
    return (B >= 1 && H >= 2 && W >= 2 && K >= 1 && N >= 32 && N % 32 == 0 && (size_t)B * H * W * (K > N ? K : N) < (1u << 31)
            && make_geo_fwd(B, H, W, &g)) ? 1 : 0;
}

size_t phx_conv3x3_f32_mfma_packed_floats(int K, int N) { return (size_t)((K + 7) & ~7) * 9 * N; }

int phx_pack_conv3x3_f32_multi(const void* descs_dev, int n, void* stream) {
    PHX_REQUIRE(descs_dev != nullptr && n >= 1, PHX_E_INVAL, "phx_pack_conv3x3_f32_multi: no jobs");
    k_pack_conv3x3_f32_multi<<<dim3(48, n), 256, 0, (hipStream_t)stream>>>((const PackF32Job*)descs_dev);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_conv3x3_f32_mfma(const float* x, const float* wpk, const float* bias, float* y, int B, int H, int W, int K, int N, int act,
                         void* stream) {
    PHX_REQUIRE(x && wpk && y, PHX_E_INVAL, "phx_conv3x3_f32_mfma: null pointer");
    PHX_REQUIRE(phx_conv3x3_f32_mfma_supported(B, H, W, K, N), PHX_E_SHAPE, "phx_conv3x3_f32_mfma: N must be a multiple of 32");
    F32Geo g;
    make_geo_fwd(B, H, W, &g);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    hipStream_t s = (hipStream_t)stream;
    F32Geo gs;
    if (ntiles * (N % 64 == 0 ? N / 64 : N / 32) < 512 && make_geo_small(B, H, W, &gs)) {
        // fewer than two 256-row blocks per CU: the 64-row tiles with the reduction split over the waves
        const size_t lds = ((size_t)32 * gs.npatch + 4 * 9 * 2 * 32 * 4) * 4;
        static bool attr = false;
        if (!attr) {
            PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_mfma_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        k_conv3x3_f32_mfma_small<<<dim3(gs.tiles_x * gs.tiles_y * gs.tiles_b, N / 32), 256, lds, s>>>(x, wpk, bias, y, B, H, W, K, N, act, gs);
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    // Which staging: the register-staged kernel keeps two blocks per CU (each covers the other's barriers: 0.70 of the fp32 peak on
    // 128 -> 128 @ 128 x 128), the LDS-DMA kernel one 512-thread block (0.65 there) -- but it is the faster one on the 32-channel
    // blocks (192 -> 32: 0.77 against 0.73, 32 -> 32: 0.63 against 0.57) and where 512 block slots leave a half-empty last round
    // (256 -> 192 @ 32 x 32, 768 blocks: 0.76 against 0.61).  Measured, tools/bench_f32_mfma.py.
    const long nblk = (long)ntiles * (N % 64 == 0 ? N / 64 : N / 32);
    const double eff_regs = 0.70 * (double)nblk / (double)(((nblk + 511) / 512) * 512);
    const double eff_dma = 0.65 * (double)nblk / (double)(((nblk + 255) / 256) * 256);
    if (K % 4 == 0 && (double)B * H * W * K * 4.0 < 4294967296.0 && g.npatch * 2 <= 8 * 4 * 64 && (N % 64 != 0 || eff_dma > eff_regs)) {
        // LDS-DMA staging into two stages, loader waves beside the computing waves (one 512-thread block per CU)
        const int bn = N % 64 == 0 ? 64 : 32;
        const size_t lds = (size_t)2 * (((g.npatch * 2 + 63) >> 6) + ((9 * bn * 2 + 63) >> 6)) * 1024;
        static bool attr64 = false, attr32 = false;
        if (bn == 64) {
            if (!attr64) {
                PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_mfma_dma<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr64 = true;
            }
            k_conv3x3_f32_mfma_dma<64><<<dim3(ntiles, N / 64), 512, lds, s>>>(x, wpk, bias, y, B, H, W, K, N, act, g);
        } else {
            if (!attr32) {
                PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_mfma_dma<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr32 = true;
            }
            k_conv3x3_f32_mfma_dma<32><<<dim3(ntiles, N / 32), 512, lds, s>>>(x, wpk, bias, y, B, H, W, K, N, act, g);
        }
        PHX_CHECK_LAUNCH();
        return PHX_OK;
    }
    if (N % 64 == 0) {
        const size_t lds = ((size_t)8 * g.npatch + 9 * 2 * 64 * 4) * 4;
        static bool attr = false;
        if (!attr) {
            PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_mfma<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        k_conv3x3_f32_mfma<64><<<dim3(ntiles, N / 64), 256, lds, s>>>(x, wpk, bias, y, B, H, W, K, N, act, g);
    } else {
        const size_t lds = ((size_t)8 * g.npatch + 9 * 2 * 32 * 4) * 4;
        static bool attr = false;
        if (!attr) {
            PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_mfma<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        k_conv3x3_f32_mfma<32><<<dim3(ntiles, N / 32), 256, lds, s>>>(x, wpk, bias, y, B, H, W, K, N, act, g);
    }
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_conv3x3_f32_mfma_wgrad_supported(int B, int H, int W, int Cin, int Cout) {
    WgradPlan p;
    return (B >= 1 && H >= 2 && W >= 2 && Cin >= 1 && Cout >= 32 && Cout % 32 == 0
            && (size_t)B * H * W * (Cin > Cout ? Cin : Cout) < (1u << 31) && plan_wgrad(B, H, W, Cin, Cout, true, &p)) ? 1 : 0;
}

size_t phx_conv3x3_f32_mfma_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int with_bias) {
    WgradPlan p;
    if (!plan_wgrad(B, H, W, Cin, Cout, with_bias != 0, &p)) return 0;
    return (p.ws_floats + p.wsb_floats) * 4;
}

int phx_conv3x3_f32_mfma_wgrad(const float* x, const float* dy, float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes,
                               int B, int H, int W, int Cin, int Cout, void* stream) {
    PHX_REQUIRE(x && dy && dw_hwio && workspace, PHX_E_INVAL, "phx_conv3x3_f32_mfma_wgrad: null pointer");
    PHX_REQUIRE(phx_conv3x3_f32_mfma_wgrad_supported(B, H, W, Cin, Cout), PHX_E_SHAPE, "phx_conv3x3_f32_mfma_wgrad: unsupported shape");
    WgradPlan p;
    plan_wgrad(B, H, W, Cin, Cout, dbias != nullptr, &p);
    PHX_REQUIRE(workspace_bytes >= (p.ws_floats + p.wsb_floats) * 4, PHX_E_INVAL, "phx_conv3x3_f32_mfma_wgrad: workspace too small");
    float* ws = (float*)workspace;
    float* wsb = dbias ? ws + p.ws_floats : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.nslice, p.ciblocks * p.coblocks);
#define PHX_F32_WGRAD(CIW, COW)                                                                                                      \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_wgrad<CIW, COW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              160 * 1024));                                                                          \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        k_conv3x3_f32_wgrad<CIW, COW><<<grid, 256, p.lds, s>>>(x, dy, ws, wsb, B, H, W, Cin, Cout, p.ciblocks, p.tps, p.ntiles, p.g); \
    } while (0)
#define PHX_F32_WGRAD_DMA(CIW, COW)                                                                                                  \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_f32_wgrad_dma<CIW, COW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              160 * 1024));                                                                          \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        k_conv3x3_f32_wgrad_dma<CIW, COW><<<grid, 512, p.lds, s>>>(x, dy, ws, wsb, B, H, W, Cin, Cout, p.ciblocks, p.tps, p.ntiles, p.g); \
    } while (0)
    if (p.dma) {
        if (p.ciw == 2 && p.cow == 2) PHX_F32_WGRAD_DMA(2, 2);
        else if (p.ciw == 1 && p.cow == 2) PHX_F32_WGRAD_DMA(1, 2);
        else if (p.ciw == 2 && p.cow == 1) PHX_F32_WGRAD_DMA(2, 1);
        else PHX_F32_WGRAD_DMA(1, 1);
    }
    else if (p.ciw == 2 && p.cow == 2) PHX_F32_WGRAD(2, 2);
    else if (p.ciw == 1 && p.cow == 2) PHX_F32_WGRAD(1, 2);
    else if (p.ciw == 2 && p.cow == 1) PHX_F32_WGRAD(2, 1);
    else PHX_F32_WGRAD(1, 1);
#undef PHX_F32_WGRAD_DMA
#undef PHX_F32_WGRAD
    PHX_CHECK_LAUNCH();
    const size_t n = (size_t)9 * Cin * Cout;
    const size_t tot = n + (dbias ? (size_t)Cout : 0);
    k_f32_wgrad_reduce<<<(unsigned)((tot + 63) / 64), 256, 0, s>>>(ws, p.nslice, n, dw_hwio, wsb, Cout, dbias);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
