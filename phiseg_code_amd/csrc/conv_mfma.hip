// bf16 MFMA 3x3 convolution for gfx950 (v_mfma_f32_32x32x16_bf16), NHWC, im2col-free implicit GEMM.
//   forward / data-gradient : k_conv3x3_mfma   (M = pixels, N = output channels, K = 9 * input channels)
//   filter-gradient         : k_conv3x3_wgrad  (M = Cin, N = Cout, K = pixels; LDS transpose reads)
// Replaces tf.nn.conv2d 3x3 SAME (tfwrapper/layers.py:123) and the two gradients TF derives for it.
//
// Tile geometry (all kernels): a block owns 256 output pixels = tb images x th rows x tw cols with
// th, tw = min(16, pow2ceil(H|W)), so 128x128 maps use 16x16 patches and the 2x2 .. 8x8 levels pack several
// images into one tile.  The (th+2) x (tw+2) zero-padded input patch is staged once per 32-channel chunk and
// re-used by all nine filter taps from LDS.
#include "conv_common.h"

// Kernel-selection policy of the forward / data-gradient launches: 1 = the measured policy below, 0 = never, 2 = whenever the shape
// is eligible.  In libphx.so it is a CONSTANT (1): the product library has no mutable policy, so a plan's tile counts and statistics
// rows cannot go stale under it.  Only the test build (libphx_dbg.so: the same sources with -DPHX_DEBUG_BUILD, include/phx_debug.h)
// can set it -- the kernel tests force every kernel family onto small shapes the CPU oracle finishes in seconds.
#ifdef PHX_DEBUG_BUILD
static int g_ws_policy = 1;                // large-map kernels on 16 x 32-pixel tiles (k_conv3x3_pp, k_conv3x3_c32)
static int g_tile_policy = 1;              // 16 x 32-pixel / 8-wave instantiations of k_conv3x3_mfma
#else
static constexpr int g_ws_policy = 1, g_tile_policy = 1;
#endif

// forward / data-gradient tiles.  16 x 32 tiles (512 pixels, 8-wave blocks, one per CU) halve the filter-slab bytes staged
// per FLOP but give up the overlap of two independent blocks per CU: measured, they win for 128-wide output-channel blocks
// and for the K >= 128 -> 32 layers, when the map has at least two such tiles per CU; everything else uses 256-pixel tiles.
static bool fwd_big_tiles(int B, int H, int W, int K, int N) {
    if (!g_tile_policy || H % 32 != 0 || W % 16 != 0) return false;
    if (g_tile_policy == 2) return true;                       // tests: whenever the map allows
    return (N % 128 == 0 || (N == 32 && K >= 128)) && (long)B * (H / 32) * (W / 16) >= 512;
}
// Large maps: 16 x 32-pixel tiles, LDS-DMA staged, when the map has at least 512 tiles x 64 (32)-channel blocks, i.e. a work item per
// CU for the pair kernel (the 128 x 128 and 64 x 64 levels at batch 64): measured 1.1-1.5x the 256-pixel kernels there, equal or
// behind on 32 x 32 maps.  32-channel blocks (N % 64 == 32): 13 % faster than the 256-pixel kernel at K = 32, 8 % at K = 64, 5 %
// SLOWER at K = 192 (the patch is staged once per 32 output channels).
static bool fwd_ws64(int B, int H, int W, int K, int N) {
    if (!g_ws_policy || H % 16 != 0 || W % 32 != 0 || N % 32 != 0 || K % 32 != 0) return false;
    if ((long)B * (H / 16) * (W / 32) >= 65536 || (double)B * H * W * (K > N ? K : N) >= 2147483648.0) return false;      // (tile index arithmetic of the large-map kernels)
    if (g_ws_policy >= 2) return true;
    if (N % 64 != 0 && K > 64) return false;
    return (long)B * (H / 16) * (W / 32) * (N / (N % 64 == 0 ? 64 : 32)) >= 512;
}
static MTile make_mtile_fwd(int B, int H, int W, int K, int N, bool allow_dma = true) {
    if (allow_dma && fwd_ws64(B, H, W, K, N)) {
        MTile g;
        g.tws = 5; g.ths = 4; g.tb = 1;
        g.tiles_x = W / 32; g.tiles_y = H / 16; g.tiles_b = B;
        mtile_magic(&g);
        return g;
    }
    if (!fwd_big_tiles(B, H, W, K, N)) return make_mtile(B, H, W);
    MTile g;
    g.tws = 4; g.ths = 5; g.tb = 1;
    g.tiles_x = W / 16; g.tiles_y = H / 32; g.tiles_b = B;
    mtile_magic(&g);
    return g;
}

// ---- filter packing ---------------------------------------------------------------------------------
// Packed bf16 filter for an (N rows) x (K reduction channels) convolution: [K / 32][9 taps][N][32], i.e. the 9 x N x 64 B
// slab a block stages per 32-channel chunk is CONTIGUOUS (full 128-byte lines; with [9][N][K] every row contributed a
// 64-byte half line and the staging path, which bounds these kernels, moved twice the requests).
__host__ __device__ __forceinline__ size_t pk_idx(int t, int n, int k, int N) {
    return ((((size_t)(k >> 5) * 9 + t) * N + n) << 5) + (k & 31);
}
// wpk_fwd(t, co, ci) = w[t][ci][co];  wpk_dgrad(8 - t, ci, co) = w[t][ci][co]   (element addressing: pk_idx)
__global__ void k_pack_conv3x3(const float* __restrict__ w, unsigned short* __restrict__ wf,
                               unsigned short* __restrict__ wd, int Cin, int Cout) {
    const size_t n = (size_t)9 * Cin * Cout;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const int ci = (int)((i / Cout) % Cin);
        const int t = (int)(i / ((size_t)Cout * Cin));
        const unsigned short v = f2bf(w[i]);
        if (wf) wf[pk_idx(t, co, ci, Cout)] = v;
        if (wd) wd[pk_idx(8 - t, ci, co, Cin)] = v;
    }
}

// Cin < 32 (image-input convolutions, Cin = 1 or 3): zero-pad the channel axis to 32 so the layer runs on the MFMA
// kernels.  wpk(t, co, ci_pad) = ci < Cin ? w[t][ci][co] : 0
// and (optional) wpk_dgrad(8 - t, ci_pad, co) = the same value   (element addressing: pk_idx)
__global__ void k_pack_conv3x3_pad(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                   unsigned short* __restrict__ wd, int Cin, int Cpad, int Cout) {
    const size_t n = (size_t)9 * Cout * Cpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cpad);
        const int co = (int)((i / Cpad) % Cout);
        const int t = (int)(i / ((size_t)Cpad * Cout));
        const unsigned short v = ci < Cin ? f2bf(w[((size_t)t * Cin + ci) * Cout + co]) : (unsigned short)0;
        wf[pk_idx(t, co, ci, Cout)] = v;
        if (wd) wd[pk_idx(8 - t, ci, co, Cpad)] = v;
    }
}
// all filters of a plan in ONE launch: descriptor table in device memory, blockIdx.y = filter
struct PackDesc {
    const float* w;
    unsigned short* wf;       // pk_idx(t, co, ci, Cout)
    unsigned short* wd;       // pk_idx(8 - t, ci, co, Cpad): taps flipped (nullable)
    int cin, cpad, cout, k1;  // k1 != 0: w is a 1x1 filter [Cin][Cout], packed as the centre tap of a 3x3 (other taps zero)
};
__global__ void k_pack_conv3x3_multi(const PackDesc* __restrict__ descs) {
    // 32 (ci) x 32 (co) tiles transposed through LDS so that the fp32 read, the [t][co][ci] write and the flipped
    // [8-t][ci][co] write are all coalesced (Cin_pad and Cout are multiples of 32 on the MFMA path).
    const PackDesc d = descs[blockIdx.y];
    __shared__ float tile[32][33];
    const int nci = d.cpad / 32, nco = d.cout / 32;
    const int ntile = 9 * nci * nco;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        const int cot = tl % nco, cit = (tl / nco) % nci, t = tl / (nco * nci);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = cit * 32 + ty * 4 + r, co = cot * 32 + tx;
            float v = 0.f;
            if (ci < d.cin) {
                if (!d.k1) v = d.w[((size_t)t * d.cin + ci) * d.cout + co];
                else if (t == 4) v = d.w[(size_t)ci * d.cout + co];
            }
            tile[ty * 4 + r][tx] = v;
            if (d.wd) d.wd[pk_idx(8 - t, ci, co, d.cpad)] = f2bf(v);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cot * 32 + ty * 4 + r, ci = cit * 32 + tx;
            d.wf[pk_idx(t, co, ci, d.cout)] = f2bf(tile[tx][ty * 4 + r]);
        }
    }
}
template <typename T>
__global__ void k_unpad_channels(const unsigned short* __restrict__ src, T* __restrict__ dst, int C, int Cpad, size_t npix) {
    const size_t n = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        stf<T>(dst, i, bf2f(src[p * Cpad + c]));
    }
}
template <typename T>
__global__ void k_pad_channels(const T* __restrict__ x, unsigned short* __restrict__ out, int C, int Cpad, size_t npix) {
    const size_t n = npix * Cpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cpad;
        const int c = (int)(i - p * Cpad);
        out[i] = c < C ? f2bf(ldf<T>(x, p * C + c)) : (unsigned short)0;
    }
}
// ntap = 9: all taps; ntap = 1: only the centre tap (a 1x1 filter run as the centre tap of a 3x3)
__global__ void k_unpad_rows_acc(const float* __restrict__ dwp, float* __restrict__ dw, int Cin, int Cpad, int Cout, int ntap) {
    const int n = ntap * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int co = i % Cout, ci = (i / Cout) % Cin, t = ntap == 1 ? 4 : i / (Cout * Cin);
        dw[i] += dwp[((size_t)t * Cpad + ci) * Cout + co];
    }
}

int phx_c32_set_trace(void* dev_buf);
int phx_wgrad_set_debug(void* trace_buf, void* blocklog_buf, int which);      // conv_wgrad.hip
bool phx_c32_enabled();                       // conv_c32.hip: the 32 -> 32-channel layers on large maps (filter in registers, persistent)
int phx_c32_launch(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial, int B, int H, int W,
                   const float* oscale, int stats_nrep, void* stream, const float* xscale = nullptr, const float* xshift = nullptr);
// anti-phase pair kernel for large maps (conv_pp.hip)
struct Dual;
bool phx_pp_shape_ok(int B, int H, int W, int K, int N);
int phx_pp_set_trace(void* dev_buf);
int phx_pp_set_grid(int blocks);
int phx_pp_launch(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial, int B, int H, int W,
                  int K, int N, const float* oscale, int stats_nrep, Dual du, int dbg, void* stream);

// ---- forward / dgrad ----------------------------------------------------------------------------------
// Options of the forward / data-gradient epilogue.
struct EpiOpts {
    int stats_atomic;         // stats_partial is the accumulator sums[N][2] itself, added to atomically (few pixel tiles)
    const float* oscale;      // per-output-channel scale of the bias / activation epilogue (inference-mode batch norm folded in)
    float* y_f32;             // fp32 output tensor [B * H * W][N] (phx_conv3x3_mfma_bf16_f32out): the split-K kernel's fp32 slices, summed
    bool keep_slices;         // ... or left in the workspace: no finishing pass (the consumer sums them: phx_bn_wide_fwd)
};

// Arguments of the one-launch conv + bias + group / instance norm + activation epilogue (FGN instantiations).
struct XForm {
    const float *gamma, *beta;    // [N]
    float eps;
    int G, act;
    unsigned short* a_out;    // [B][H][W][N] bf16
    float *mean_out, *rstd_out, *scale_out, *shift_out;     // [B][G], [B][N]
};
// Concat-free convolution (posteriors.py:87,120, priors.py:112, likelihoods.py:210 feed tf.concat([a, b], axis=3) to a 3x3 conv2D):
// forward / filter gradient read the two tensors in place -- reduction channels [0, K1) from x (pixel stride K1), [K1, K) from x2
// (pixel stride K - K1), K1 % 32 == 0 -- and the data gradient writes the two halves of d(concat) to two tensors: output channels
// [0, N1) to y (stride N1), [N1, N) to y2 (stride N - N1), N1 % 8 == 0.  x2 / y2 == NULL: the ordinary single-tensor launch.
struct Dual {
    const unsigned short* x2;
    unsigned short* y2;
    int K1, N1;
};

// NA = compile-time bound on the 16-byte input-patch pieces a thread stages per 32-channel chunk
// (ceil(npatch * 4 / 256): 6 for 16x16 tiles, 7 for 8x8x4, 9 for 4x4x16, 16 for 2x2x64).
// NW = waves per block (4: 256-pixel tiles, two blocks per CU; 8: 512-pixel 16 x 32 tiles, one block per CU -- the
// filter slab, 64 % of the staged bytes of a 256 x 64 tile, is then shared by twice the pixels: the kernel is bound
// by the L2 -> LDS path (~12 B/clk/CU), so bytes staged per FLOP set its speed).
// SPLITK (small maps: a handful of pixel tiles cannot fill 256 CUs and each block would walk all K / 32 chunks serially,
// ~2 us apiece): gridDim.z blocks share a tile, each takes a run of chunks and stores its fp32 accumulators to
// ws[z][pixel][N]; k_splitk_finish sums the slices, adds bias / activation and writes the bf16 tensor.
// DUAL (struct Dual; its own instantiations, so that the ordinary launches carry neither the second plan nor the selects): the
// second offset plan ga2 addresses du.x2, every prefetched chunk picks its tensor by a scalar compare; the epilogue stores by piece.
// FGN (maps of at most 16 x 16 pixels = whole samples per pixel tile; group norm with 16-channel groups or instance norm): the
// north_star's fused block -- conv2d + bias + group / instance norm + activation (tfwrapper/layers.py:123-135, normalisation.py:3-36)
// in ONE launch with no cross-block step at all: a block owns its samples' pixels for its BN channels, i.e. whole groups, so the
// two-pass statistics (sum, then sum of squared deviations, through a small LDS table) are block-local.  xf carries gamma / beta /
// eps / act / G / a_out and the published per-sample vectors; bias is the convolution's.
template <int BN, int NA, bool FAST16, bool BIASACT, int NW, bool SPLITK, bool DUAL = false, bool FGN = false>
// (NA = 16 -- the 4 x 4 x 16 / 2 x 2 x 64 tiles of the H <= 4 levels: at most a few dozen blocks per launch -- with the second plan of
// the DUAL instantiations is compiled for one block per CU: 96 staging registers + two plans + accumulators do not fit 256)
__global__ __launch_bounds__(NW * 64, (NW == 8 || (NA > 8 && DUAL)) ? 1 : 2) void k_conv3x3_mfma(const unsigned short* __restrict__ x,
                                                         const unsigned short* __restrict__ wpk,
                                                         unsigned short* __restrict__ y, const float* __restrict__ bias,
                                                         int act, float* __restrict__ stats_partial, int B, int H, int W,
                                                         int K, int N, MTile g, float* __restrict__ ws, EpiOpts bws, XForm xf, Dual du) {
    constexpr int NJ = BN / 32;
    constexpr int NT = NW * 64;                            // threads; the tile has NT pixels
    constexpr int NB = (9 * BN * 4 + NT - 1) / NT;        // filter-slab pieces per thread
    const int tw = FAST16 ? 16 : 1 << g.tws, th = FAST16 ? NT / 16 : 1 << g.ths;
    const int pw = tw + 2, ph = th + 2;
    const int npatch = FAST16 ? 18 * (NT / 16 + 2) : g.tb * ph * pw;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                    // [npatch][ROWB]
    // (the 512-pixel x 32-channel variant measured 12 % slower with the padded pitch -- it keeps the dense one)
    constexpr int P16 = (NW == 8 && BN == 32) ? 18 * ROWB : PITCH16;
    const int pitch = FAST16 ? P16 : g.rpitch;
    unsigned char* sB = smem + (FAST16 ? (NT / 16 + 2) * P16 : g.tb * g.ipitch);    // [9][BN][ROWB]

    // Linear block id -> (pixel tile, channel block).  Work-groups go round-robin over the 8 XCDs (id % 8), each with its
    // own L2: the N / BN channel blocks of a tile get ids 8 apart -- same XCD, dispatched back to back -- so the input
    // patch they all read comes from HBM once; the tiles of an XCD are a contiguous band of the launch (phx_band8), so the halos
    // of neighbouring tiles meet in that L2 too.  (Placement only affects speed, never results.)
    int tile_id, cob;
    {
        const int ncob = N / BN, ntl = g.tiles_x * g.tiles_y * g.tiles_b;
        const int id = blockIdx.x, full = (ntl >> 3) * 8 * ncob;
        if (id < full) {
            const int grp = id / (8 * ncob), r = id - grp * 8 * ncob;
            tile_id = phx_band8(grp * 8 + (r & 7), ntl);           // (XCD bands: phx_common.h)
            cob = r >> 3;
        } else {
            const int rem = ntl & 7, r = id - full;
            tile_id = (ntl & ~7) + r % rem;
            cob = r / rem;
        }
    }
    int t = tile_id;
    const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
    const int b0 = t * g.tb;
    const int n0 = cob * BN;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, khalf = lane >> 5;
    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wave * 64 + i * 32 + l31;
        const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
        aoff[i] = (FAST16 ? (lb * ph + ly) * pitch : lb * g.ipitch + ly * pitch) + lx * ROWB + khalf * 16;
    }
    const int boff = l31 * ROWB + khalf * 16;

    // staging plan: BYTE offsets (without the channel-chunk term) of this thread's 16-byte pieces for raw buffer loads;
    // 0xffffffff = outside the image / batch / slab -> the buffer range check returns zeros, so the prefetch is
    // branch-free and can be interleaved with the MFMAs of the running chunk.
    unsigned ga[NA];
    unsigned ga2[DUAL ? NA : 1];                 // DUAL: the same pieces in du.x2 (pixel stride K - K1)
    unsigned sa[FAST16 ? 1 : NA];                // small-map tiles: LDS byte offset of the piece's pixel (padded pitches, see mtile_magic)
    const int K1 = (DUAL && du.x2 != nullptr) ? du.K1 : K;      // channels (= pixel stride) of x
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int i = threadIdx.x + it * NT;
        const int q = i & 3, pp = i >> 2;
        ga[it] = 0xffffffffu;
        if constexpr (DUAL) ga2[it] = 0xffffffffu;
        if constexpr (!FAST16) sa[it] = 0;
        if (pp < npatch) {
            // 16-wide tiles: the patch is 18 wide -> compile-time divisors (runtime division costs ~40 instructions)
            int px, py, pb;
            if constexpr (FAST16) { px = pp % 18; py = pp / 18; pb = 0; }
            else patch_coords(g, pp, pw, ph, &px, &py, &pb);
            if constexpr (!FAST16) sa[it] = (unsigned)(pb * g.ipitch + py * g.rpitch + px * ROWB);
            const int gx = tx0 + px - 1, gy = ty0 + py - 1, gbi = b0 + pb;
            const bool in = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && gbi < B;
            unsigned off, off2 = 0;
            if constexpr (FAST16) {
                off = (unsigned)((((gbi * H + gy) * W + gx) * K1 + q * 8) * 2);
                if constexpr (DUAL) off2 = (unsigned)((((gbi * H + gy) * W + gx) * (K - K1) + q * 8) * 2);
            } else {                                           // small maps: B * H * W < 2^24 (checked by the launcher)
                const unsigned pix = __umul24(__umul24((unsigned)gbi, (unsigned)H) + gy, (unsigned)W) + gx;
                off = (__umul24(pix, (unsigned)K1) + q * 8) * 2;
                if constexpr (DUAL) off2 = (__umul24(pix, (unsigned)(K - K1)) + q * 8) * 2;
            }
            ga[it] = in ? off : 0xffffffffu;
            if constexpr (DUAL) ga2[it] = in ? off2 : 0xffffffffu;
        }
    }
    // filter-slab pieces: piece `it` of a thread lies it * 64 slab rows further on, i.e. a fixed byte stride -> one VGPR
    // offset plus a scalar stride (gbl: the last piece, only partly populated when 9 * BN * 4 is not a multiple of NT)
    unsigned gb0, gbl;
    {
        const int q = threadIdx.x & 3, row = threadIdx.x >> 2;
        const int tap = row / BN, n = row - tap * BN;
        gb0 = (unsigned)(((tap * N + n0 + n) * 32 + q * 8) * 2);          // chunk 0; chunk c adds c * 9 * N * 64 bytes
        gbl = (threadIdx.x + (NB - 1) * NT < 9 * BN * 4) ? gb0 : 0xffffffffu;
    }
    // a thread's consecutive slab pieces are NT / 4 slab rows = NT / 4 / BN taps apart (BN <= NT / 4)
    static_assert(NT / 4 % BN == 0, "slab piece stride must be whole taps");
    const int gbs = (NT / 4 / BN) * N * 64;
    const int gcs = 9 * N * 64;                   // bytes per 32-channel chunk of the packed filter
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * K1 * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? du.x2 : x), 0, DUAL ? (int)((unsigned)B * H * W * (K - K1) * 2u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, (int)(9u * N * K * 2u), 0x00020000);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    PHX_BLOCKLOG_BEGIN();
    PHX_TRACE(0);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 ra[NA], rb[NB];
    // piece `idx` (input-patch pieces first, then filter-slab pieces) of the chunk starting at channel c0
    auto prefetch_piece = [&](auto idxc, int c0) {
        constexpr int idx = decltype(idxc)::value;
        if constexpr (PHX_ABLATE & 1) return;
        if constexpr (idx < NA) {
            if constexpr (DUAL) {
                if (c0 >= K1) ra[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsx2, ga2[idx], (c0 - K1) * 2, 0);
                else ra[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ga[idx], c0 * 2, 0);
            } else ra[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ga[idx], c0 * 2, 0);
        }
        else if constexpr (idx < NA + NB)
            rb[idx - NA] = __builtin_amdgcn_raw_buffer_load_b128(rsw, idx - NA == NB - 1 ? gbl : gb0, (c0 >> 5) * gcs + (idx - NA) * gbs, 0);
    };
    // this block's run of 32-channel chunks [cbeg, cend)
    int cbeg = 0, cend = K;
    if constexpr (SPLITK) {
        const int per = (K / KC + gridDim.z - 1) / gridDim.z;
        cbeg = blockIdx.z * per * KC;
        cend = min(K, cbeg + per * KC);
    }
    {
        auto all = [&](auto self, auto idxc) {
            constexpr int idx = decltype(idxc)::value;
            if constexpr (idx < NA + NB) {
                prefetch_piece(idxc, cbeg);
                self(self, std::integral_constant<int, idx + 1>());
            }
        };
        all(all, std::integral_constant<int, 0>());
    }
    PHX_TRACE(1);
    // one 32-channel chunk: registers -> LDS, then 18 (tap, k-step) groups of 2 x NJ MFMAs; with PF the global loads of the
    // NEXT chunk are issued one or two per group, so the texture-address unit (64 B/clk: ~1 K cycles per chunk for the four
    // waves) works underneath the matrix pipe instead of in a phase of its own.
#ifndef PHX_PF_GROUPS   // the next chunk's global loads are spread over the first PHX_PF_GROUPS of the 18 (tap, k-step) groups (dev A/B: tools/build_variant.sh)
#define PHX_PF_GROUPS 18
#endif
    constexpr int IPG = (NA + NB + PHX_PF_GROUPS - 1) / PHX_PF_GROUPS;
    auto chunk = [&](int ccur, int cnext, auto pfc, bool tr) {   // ccur: first channel of the chunk in registers; cnext: of the chunk to prefetch
        constexpr bool PF = decltype(pfc)::value;
        __syncthreads();                         // every wave is done reading the previous chunk / epilogue tile from LDS
        if (tr) PHX_TRACE(2);
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int i = threadIdx.x + it * NT;
            const int pp = i >> 2;
            const int so = FAST16 ? (pp / 18) * (P16 - 18 * ROWB) + pp * ROWB : (int)sa[FAST16 ? 0 : it];
            if (!(PHX_ABLATE & 2) && pp < npatch) *reinterpret_cast<u32x4*>(sA + so + (i & 3) * 16) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int i = threadIdx.x + it * NT;
            if (!(PHX_ABLATE & 2) && i < 9 * BN * 4) *reinterpret_cast<u32x4*>(sB + (i >> 2) * ROWB + (i & 3) * 16) = rb[it];
        }
        __syncthreads();
        if (tr) PHX_TRACE(3);
        // software pipeline, pinned with sched_barrier: the operand reads of group gi+1 and this group's global loads are
        // issued before the MFMAs of group gi (fragment registers double-buffered by group parity)
        // (NJ = 4 keeps a single fragment set -- 128 accumulator registers leave no room for two; its 8 MFMAs per group
        // cover the LDS latency of the next group's reads, which are issued right behind them)
        // (64-channel blocks on the 4 x 4 x 16 / 2 x 2 x 64 tiles -- NA = 16 staging registers -- read ONE group ahead: with three
        // fragment sets the instantiation needs 265 registers and spilled 36 - 84 bytes per lane to scratch, the only scratch user
        // among the kernels of the headline step; the DUAL ones are compiled for one block per CU and have room)
        constexpr int FB = NJ <= 2 ? ((NA > 8 && NJ == 2 && !DUAL) ? PHX_FRAG_DEPTH : PHX_FRAG_DEPTH + 1) : 1;      // fragment sets: reads run FB - 1 groups ahead
        bf16x8 fa[FB][2], fb[FB][NJ];
        auto read_frags = [&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            constexpr int tap = gi / 2, ks = gi % 2, kh = tap / 3, kw = tap % 3;
            const int tapoff = kh * pitch + kw * ROWB;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[gi % FB][i] = *reinterpret_cast<const bf16x8*>(sA + aoff[i] + tapoff + ks * 32);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                fb[gi % FB][j] = *reinterpret_cast<const bf16x8*>(sB + (tap * BN + j * 32) * ROWB + boff + ks * 32);
        };
        read_frags(std::integral_constant<int, 0>());
        if constexpr (FB >= 3) read_frags(std::integral_constant<int, 1>());
        auto group = [&](auto self, auto gc) {
            constexpr int gi = decltype(gc)::value;
            if constexpr (gi < 18) {
                if constexpr (FB >= 2 && gi + FB - 1 < 18) read_frags(std::integral_constant<int, gi + FB - 1>());
                if constexpr (PF) {
                    auto pieces = [&](auto self2, auto pc) {
                        constexpr int pi = decltype(pc)::value;
                        if constexpr (pi < IPG) {
                            prefetch_piece(std::integral_constant<int, gi * IPG + pi>(), cnext);
                            self2(self2, std::integral_constant<int, pi + 1>());
                        }
                    };
                    pieces(pieces, std::integral_constant<int, 0>());
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if constexpr (PHX_ABLATE & 4) acc[i][j][0] += (float)fa[gi % FB][i][0] * (float)fb[gi % FB][j][0];   // (static index: keeps the operand reads alive)
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gi % FB][i], fb[gi % FB][j], acc[i][j], 0, 0, 0);
                if constexpr (FB == 1 && gi < 17) read_frags(std::integral_constant<int, gi + 1>());
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, gi + 1>());
            }
        };
        group(group, std::integral_constant<int, 0>());
        if (tr) PHX_TRACE(4);
    };
    const int odd = lane & 1;
    // Epilogue tile [pixel][BN] bf16 in LDS.  32 / 64-channel blocks: DENSE rows -- a lane group of the 16-byte read-back covers all 16
    // slots of a bank row, and the 32-bit writes of a lane pair's four rows (r, r + 1, r + 4, r + 5) hit every bank twice, the
    // minimum for 64 lanes; for 64-channel blocks that needs the 16-byte piece index XORed with 4 * (row & 1) (rows r and r + 1
    // would share their 16 banks otherwise).  The padded rows used before (80 / 144 bytes) were two- to three-way conflicted on
    // the read-back (23 % of these kernels' LDS cycles).  128-channel blocks keep the padded pitch.
    constexpr int OROW = BN == 128 ? BN * 2 + 16 : BN * 2;
    constexpr int OSWZ = BN == 64 ? 64 : 0;               // byte XOR (piece bit 2) applied on odd rows
    {
        const int cx0 = tx0, cy0 = ty0, cb0 = b0;
        const bool tr0 = true;
        for (int c0 = cbeg; c0 + KC < cend; c0 += KC) chunk(c0, c0 + KC, std::true_type(), c0 == KC);
        chunk(cend - KC, 0, std::false_type(), K == 2 * KC);
        if constexpr (FGN) {
            constexpr int OROWG = BN * 2, OSWZG = BN == 64 ? 64 : 0;      // the epilogue tile of the standard path (dense, swizzled)
            const int psh = g.tws + g.ths;                   // log2(pixels per sample)
            const int cg = N / xf.G, ngb = BN / cg;          // channels per group (16 or 1), groups in this channel block
            const float inv_n = 1.f / (float)((1 << psh) * cg);
            float* tbl = reinterpret_cast<float*>(smem + NT * OROWG);     // [tb][ngb][2]: {sum, sum of squared deviations}
            const int odd_g = lane & 1;
            __syncthreads();                                 // all MFMA operand reads are done
            for (int t2 = threadIdx.x; t2 < g.tb * ngb * 2; t2 += NT) tbl[t2] = 0.f;
            // y = bf16(conv + bias): what is stored, normalised and read back by the backward pass
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float bv = bias ? bias[n0 + j * 32 + l31] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        const unsigned w2 = f2bf_pk(acc[i][j][2 * rp] + bv, acc[i][j][2 * rp + 1] + bv);
                        acc[i][j][2 * rp] = __uint_as_float(w2 << 16);
                        acc[i][j][2 * rp + 1] = __uint_as_float(w2 & 0xffff0000u);
                    }
            }
            __syncthreads();
            // a segment = the four pixels (r & 3) of accumulator rows 4 rq .. 4 rq + 3: consecutive pixels, always inside one sample
            auto seg_img = [&](int i, int rq) { return (wave * 64 + i * 32 + 8 * rq + 4 * khalf) >> psh; };
            auto group_sum = [&](float v) {                  // over the 16 channels of a group (lanes l31 ^ 1, 2, 4, 8: same khalf)
                if (cg == 16) {
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                }
                return v;
            };
            const bool leader = cg == 1 || (l31 & 15) == 0;
            float mg[2][4][NJ], rg[2][4][NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const float sv = group_sum((acc[i][j][4 * rq] + acc[i][j][4 * rq + 1]) + (acc[i][j][4 * rq + 2] + acc[i][j][4 * rq + 3]));
                        if (leader) atomicAdd(&tbl[(seg_img(i, rq) * ngb + (j * 32 + l31) / cg) * 2], sv);
                    }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const float mu = tbl[(seg_img(i, rq) * ngb + (j * 32 + l31) / cg) * 2] * inv_n;
                        mg[i][rq][j] = mu;
                        float d2 = 0.f;
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) { const float dd = acc[i][j][4 * rq + k2] - mu; d2 = fmaf(dd, dd, d2); }
                        d2 = group_sum(d2);
                        if (leader) atomicAdd(&tbl[(seg_img(i, rq) * ngb + (j * 32 + l31) / cg) * 2 + 1], d2);
                    }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        rg[i][rq][j] = rsqrtf(tbl[(seg_img(i, rq) * ngb + (j * 32 + l31) / cg) * 2 + 1] * inv_n + xf.eps);
            // published per-sample vectors (backward pass): mean / rstd [NS][G], scale / shift [NS][N]
            for (int t2 = threadIdx.x; t2 < g.tb * BN; t2 += NT) {
                const int img = t2 / BN, ch = t2 - img * BN;
                if (b0 + img < B) {
                    const float mu = tbl[(img * ngb + ch / cg) * 2] * inv_n;
                    const float rs = rsqrtf(tbl[(img * ngb + ch / cg) * 2 + 1] * inv_n + xf.eps);
                    const float scv = xf.gamma[n0 + ch] * rs;
                    xf.scale_out[(size_t)(b0 + img) * N + n0 + ch] = scv;
                    xf.shift_out[(size_t)(b0 + img) * N + n0 + ch] = xf.beta[n0 + ch] - mu * scv;
                    if (ch % cg == 0) {
                        xf.mean_out[(size_t)(b0 + img) * xf.G + (n0 + ch) / cg] = mu;
                        xf.rstd_out[(size_t)(b0 + img) * xf.G + (n0 + ch) / cg] = rs;
                    }
                }
            }
            // y, then a = act((y - mean) * rstd * gamma + beta), each transposed through LDS and stored 16 bytes per lane (masked path)
            const unsigned pselg = odd_g ? 0x03020706u : 0x05040100u;
            for (int pass = 0; pass < 2; ++pass) {
                unsigned short* dst = pass == 0 ? y : xf.a_out;
                if (pass == 1) {                             // normalise the accumulators in place: uniform branches only
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const float gmv = xf.gamma[n0 + j * 32 + l31], bev = xf.beta[n0 + j * 32 + l31];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int rq = 0; rq < 4; ++rq) {
                                const float scv = rg[i][rq][j] * gmv, shv = bev - mg[i][rq][j] * scv;
#pragma unroll
                                for (int k2 = 0; k2 < 4; ++k2) acc[i][j][4 * rq + k2] = fmaf(acc[i][j][4 * rq + k2], scv, shv);
                            }
                    }
                    if (xf.act == PHX_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
                    } else if (xf.act != PHX_ACT_ID) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[i][j][r] = act_fwd(acc[i][j][r], xf.act);
                    }
                }
                __syncthreads();                             // the previous pass has read its tile (pass 0: the table is no longer needed ... it lives behind the tile)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int rp = 0; rp < 8; ++rp) {
                            const int r0 = 2 * rp;
                            const float v0 = acc[i][j][r0], v1 = acc[i][j][r0 + 1];
                            const unsigned w2 = f2bf_pk(v0, v1);
                            const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w2, 0xB1, 0xf, 0xf, true);
                            const int m0 = wave * 64 + i * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * khalf;
                            *reinterpret_cast<unsigned*>(smem + (m0 + odd_g) * OROWG + (((j * 32 + (l31 & ~1)) * 2) ^ (odd_g ? OSWZG : 0))) =
                                __builtin_amdgcn_perm(nb, w2, pselg);
                        }
                }
                __syncthreads();
                constexpr int PPPG = BN / 8;
#pragma unroll
                for (int it = 0; it < PPPG; ++it) {
                    const int i2 = threadIdx.x + it * NT;
                    const int m = i2 / PPPG, q = i2 % PPPG;
                    const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
                    const int ox = cx0 + lx, oy = cy0 + ly, ob = cb0 + lb;
                    if (ox < W && oy < H && ob < B)
                        *reinterpret_cast<uint4*>(dst + (((size_t)ob * H + oy) * W + ox) * N + n0 + q * 8) =
                            *reinterpret_cast<const uint4*>(smem + m * OROWG + ((q * 16) ^ ((m & 1) ? OSWZG : 0)));
                }
            }
            PHX_BLOCKLOG_END();
            return;
        }
        if constexpr (SPLITK) {
            // fp32 partial tile -> ws[z][pixel][N]; C layout: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            float* wz = ws + (size_t)blockIdx.z * B * H * W * N;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
                    const int ox = cx0 + lx, oy = cy0 + ly, ob = cb0 + lb;
                    if (ox < W && oy < H && ob < B) {
                        float* wr = wz + (((size_t)ob * H + oy) * W + ox) * N + n0 + l31;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) wr[j * 32] = acc[i][j][r];
                    }
                }
            PHX_BLOCKLOG_END();
            return;
        }

        // epilogue: C layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel).  The tile is transposed
        // through LDS ([pixel][BN] bf16, 16-byte padded rows) so that global stores are 16 bytes per lane, 128 contiguous
        // bytes per pixel, instead of 2-byte scattered stores.
        if (tr0) PHX_TRACE(5);
        // (statistics on packed fp32 pairs, the LDS word of a lane pair by one v_perm_b32: 8 instead of 17 VALU instructions per word)
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        f32x2_t s1v[NJ], s2v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { s1v[j] = f32x2_t{0.f, 0.f}; s2v[j] = f32x2_t{0.f, 0.f}; }
        const bool do_stats = stats_partial != nullptr;
        const unsigned psel = odd ? 0x03020706u : 0x05040100u;       // even lane: {own lo, neighbour lo}; odd: {neighbour hi, own hi}
        __syncthreads();                             // all MFMA operand reads of the last chunk are done
        if (tr0) PHX_TRACE(7);
        if constexpr (BIASACT) {                     // rare (no-norm layers): its own instantiation, so that the softplus
#pragma unroll                                       // code costs the common kernels neither registers nor issue slots
            for (int j = 0; j < NJ; ++j) {
                const float bv = bias ? bias[n0 + j * 32 + l31] : 0.f;
                const float sv = bws.oscale ? bws.oscale[n0 + j * 32 + l31] : 1.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(acc[i][j][r], sv, bv);
            }
            // the activation code is uniform per launch: ONE scalar branch here, not two per element (with act_fwd() inside the loops
            // every element carried the compare-and-branch pairs of the ReLU / softplus tests: a third of this epilogue's time)
            if (act == PHX_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
            } else if (act != PHX_ACT_ID) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = act_fwd(acc[i][j][r], act);
            }
        }
        // Lane pairs (channels n, n+1) trade one of two rows so that each lane writes ONE 32-bit word {ch n, ch n+1} per
        // row pair: half the LDS stores, no sub-dword writes.  Even lane keeps row 2rp, odd lane row 2rp+1.
        // pack two accumulator rows, add them to the statistics, hand back this lane's LDS word
        auto pack_pair = [&](int i, int j, int r0, float f0, float f1) -> unsigned {
            const unsigned w2 = f2bf_pk(acc[i][j][r0], acc[i][j][r0 + 1]);           // {lo = row r0, hi = row r0 + 1}
            if (do_stats) {
                const f32x2_t rv = {__uint_as_float(w2 << 16) * f0, __uint_as_float(w2 & 0xffff0000u) * f1};
                s1v[j] += rv;
                s2v[j] += rv * rv;
            }
            // neighbour lane's word through DPP quad_perm [1,0,3,2] (no LDS round trip)
            const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w2, 0xB1, 0xf, 0xf, true);
            return __builtin_amdgcn_perm(nb, w2, psel);
        };
        constexpr int PPP = BN / 8;                  // 16-byte pieces per pixel
        const bool full = (cx0 + tw) <= W && (cy0 + th) <= H && (cb0 + g.tb) <= B;
        if (FAST16 && full) {
            // interior 16x16 tile: every address is one per-thread base plus compile-time / scalar terms, no masks
            unsigned char* lw = smem + (wave * 64 + 4 * khalf + odd) * OROW + (l31 & ~1) * 2;      // (row parity = odd: swizzle below)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        *reinterpret_cast<unsigned*>(lw + (i * 32 + ((2 * rp) & 3) + 8 * ((2 * rp) >> 2)) * OROW + ((j * 64) ^ (odd ? OSWZ : 0))) =
                            pack_pair(i, j, 2 * rp, 1.f, 1.f);
            if (tr0) PHX_TRACE(8);
            __syncthreads();
            if (tr0) PHX_TRACE(9);
            const int mt = threadIdx.x / PPP, q = threadIdx.x % PPP;          // piece it: pixel mt + it * (NT / PPP)
            const unsigned char* lr = smem + mt * OROW + ((q * 16) ^ ((mt & 1) ? OSWZ : 0));      // (NT / PPP is even: row parity = mt & 1)
            // (dual destination: the piece's eight channels lie in y (row length N1) or in du.y2 (row length N - N1))
            unsigned short* ybase = y;
            int yld = N, ych = n0 + q * 8;
            if constexpr (DUAL)
                if (du.y2) {
                    if (ych < du.N1) yld = du.N1;
                    else { ybase = du.y2; yld = N - du.N1; ych -= du.N1; }
                }
            unsigned short* yp = ybase + (((size_t)cb0 * H + cy0 + (mt >> 4)) * W + cx0 + (mt & 15)) * yld + ych;
            const size_t ystep = (size_t)(NT / PPP / 16) * W * yld;
#pragma unroll
            for (int it = 0; it < PPP; ++it)
                if (!(PHX_ABLATE & 8) || cx0 < 0)
                    *reinterpret_cast<uint4*>(yp + it * ystep) = *reinterpret_cast<const uint4*>(lr + it * (NT / PPP) * OROW);
        } else {
            // edge tiles and the small-map tile shapes: per-row masks and addresses
            const int tid_o = threadIdx.x;
            const int wave_o = tid_o >> 6, l31_o = tid_o & 31, khalf_o = (tid_o >> 5) & 1, odd_o = tid_o & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    const int r0 = 2 * rp;
                    const int m0 = wave_o * 64 + i * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * khalf_o;   // row of r0; r0 + 1 is the next pixel
                    const int m1 = m0 + 1;
                    const int lx0 = m0 & (tw - 1), ly0 = (m0 >> g.tws) & (th - 1), lb0 = m0 >> (g.tws + g.ths);
                    const int lx1 = m1 & (tw - 1), ly1 = (m1 >> g.tws) & (th - 1), lb1 = m1 >> (g.tws + g.ths);
                    const float f0 = ((cx0 + lx0) < W && (cy0 + ly0) < H && (cb0 + lb0) < B) ? 1.f : 0.f;   // statistics mask
                    const float f1 = ((cx0 + lx1) < W && (cy0 + ly1) < H && (cb0 + lb1) < B) ? 1.f : 0.f;
                    const int mrow = odd_o ? m1 : m0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        *reinterpret_cast<unsigned*>(smem + mrow * OROW + (((j * 32 + (l31_o & ~1)) * 2) ^ (odd_o ? OSWZ : 0))) = pack_pair(i, j, r0, f0, f1);
                }
            if (tr0) PHX_TRACE(8);
            __syncthreads();
            if (tr0) PHX_TRACE(9);
#pragma unroll
            for (int it = 0; it < PPP; ++it) {       // NT pixels * PPP pieces / NT threads
                const int i = tid_o + it * NT;
                const int m = i / PPP, q = i % PPP;
                const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
                const int ox = cx0 + lx, oy = cy0 + ly, ob = cb0 + lb;
                if (ox < W && oy < H && ob < B) {
                    const uint4 v = *reinterpret_cast<const uint4*>(smem + m * OROW + ((q * 16) ^ ((m & 1) ? OSWZ : 0)));
                    unsigned short* ybase = y;
                    int yld = N, ych = n0 + q * 8;
                    if constexpr (DUAL)
                        if (du.y2) {
                            if (ych < du.N1) yld = du.N1;
                            else { ybase = du.y2; yld = N - du.N1; ych -= du.N1; }
                        }
                    *reinterpret_cast<uint4*>(ybase + (((size_t)ob * H + oy) * W + ox) * yld + ych) = v;
                }
            }
        }
        if (tr0) PHX_TRACE(6);
        if (stats_partial) {
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);      // [NW waves][2][BN]
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float t1 = s1v[j][0] + s1v[j][1], t2 = s2v[j][0] + s2v[j][1];
                const float a = t1 + __shfl_xor(t1, 32, 64);
                const float bq = t2 + __shfl_xor(t2, 32, 64);
                if (khalf == 0) {
                    red[(wave * 2 + 0) * BN + j * 32 + l31] = a;
                    red[(wave * 2 + 1) * BN + j * 32 + l31] = bq;
                }
            }
            __syncthreads();
            if (threadIdx.x < 2 * BN) {
                const int which = threadIdx.x / BN, n = threadIdx.x % BN;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += red[(w * 2 + which) * BN + n];
                if (bws.stats_atomic) atomicAdd(&stats_partial[((size_t)n0 + n) * 2 + which], v);
                else stats_partial[((size_t)tile_id * 2 + which) * N + n0 + n] = v;
            }
        }
    }
    PHX_BLOCKLOG_END();
}

// y[pix][n] = bf16(act(sum_z ws[z][pix][n] + bias[n])), four channels per thread
__global__ void k_splitk_finish(const float* __restrict__ ws, int nz, size_t total, int N, const float* __restrict__ bias,
                                int act, unsigned short* __restrict__ y, const float* __restrict__ oscale,
                                unsigned short* __restrict__ y2, int N1, float* __restrict__ yf) {
    for (size_t i4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i4 * 4 < total; i4 += (size_t)gridDim.x * blockDim.x) {
        const size_t i = i4 * 4;
        // (loads first, four slices at a time -- the plain loop compiled to load / wait / add per slice -- and the slices are added
        // in their old order: bit-identical sums)
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + i);
        const bool affine = bias != nullptr || oscale != nullptr;
        f32x4 sv = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
        if (affine) {
            const int n = (int)(i % N);
            if (oscale != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sv[q] = oscale[n + q];
            }
            if (bias != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[q] = bias[n + q];
            }
        }
        int z = 1;
        for (; z + 3 < nz; z += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ws + (size_t)z * total + i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(ws + (size_t)(z + 1) * total + i);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(ws + (size_t)(z + 2) * total + i);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(ws + (size_t)(z + 3) * total + i);
            a += v0; a += v1; a += v2; a += v3;
        }
        if (z + 1 < nz) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ws + (size_t)z * total + i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(ws + (size_t)(z + 1) * total + i);
            a += v0; a += v1;
            z += 2;
        }
        if (z < nz) a += *reinterpret_cast<const f32x4*>(ws + (size_t)z * total + i);
        if (affine) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = fmaf(a[q], sv[q], bv[q]);
        }
        if (act == PHX_ACT_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = a[q] > 0.f ? a[q] : 0.f;
        } else if (act != PHX_ACT_ID) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = act_fwd(a[q], act);
        }
        if (yf != nullptr) {                       // fp32 output (the small-map batch-norm layers normalise unrounded values)
            *reinterpret_cast<f32x4*>(yf + i) = a;
            continue;
        }
        uint2 o;
        o.x = f2bf_pk(a[0], a[1]);
        o.y = f2bf_pk(a[2], a[3]);
        if (y2 != nullptr) {                       // dual destination (see struct Dual): channels [0, N1) -> y, [N1, N) -> y2
            const size_t pix = i / N;
            const int n = (int)(i - pix * N);
            if (n < N1) *reinterpret_cast<uint2*>(y + pix * N1 + n) = o;
            else *reinterpret_cast<uint2*>(y2 + pix * (N - N1) + (n - N1)) = o;
        } else
        *reinterpret_cast<uint2*>(y + i) = o;
    }
}

extern "C" {

int phx_pack_conv3x3_bf16(const float* w_hwio, void* wpk_fwd, void* wpk_dgrad, int Cin, int Cout, void* stream) {
    const size_t n = (size_t)9 * Cin * Cout;
    hipLaunchKernelGGL(k_pack_conv3x3, dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w_hwio,
                       (unsigned short*)wpk_fwd, (unsigned short*)wpk_dgrad, Cin, Cout);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_pack_conv3x3_bf16_pad(const float* w_hwio, void* wpk_fwd, void* wpk_dgrad, int Cin, int Cin_pad, int Cout,
                              void* stream) {
    PHX_REQUIRE(Cin_pad % 32 == 0 && Cin <= Cin_pad, PHX_E_SHAPE, "pack_pad: Cin_pad % 32 == 0 and Cin <= Cin_pad");
    const size_t n = (size_t)9 * Cin_pad * Cout;
    hipLaunchKernelGGL(k_pack_conv3x3_pad, dim3(phx_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w_hwio,
                       (unsigned short*)wpk_fwd, (unsigned short*)wpk_dgrad, Cin, Cin_pad, Cout);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_pack_conv3x3_bf16_multi(const void* descs_dev, int n, void* stream) {
    if (n <= 0) return PHX_OK;
    hipLaunchKernelGGL(k_pack_conv3x3_multi, dim3(48, n), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs_dev);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_pad_channels_bf16(const void* x, int dt, int C, void* out, int Cpad, size_t npix, void* stream) {
    PHX_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((k_pad_channels<T>), dim3(phx_grid_for(npix * Cpad, 256, 8192)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)x, (unsigned short*)out, C, Cpad, npix);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_unpad_channels_bf16(const void* src, void* dst, int dst_dt, int C, int Cpad, size_t npix, void* stream) {
    PHX_DT_SWITCH(dst_dt, T, {
        hipLaunchKernelGGL((k_unpad_channels<T>), dim3(phx_grid_for(npix * C, 256, 8192)), dim3(256), 0,
                           (hipStream_t)stream, (const unsigned short*)src, (T*)dst, C, Cpad, npix);
    });
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_unpad_filter_grad_accumulate(const float* dw_pad, float* dw_hwio, int Cin, int Cin_pad, int Cout, void* stream) {
    hipLaunchKernelGGL(k_unpad_rows_acc, dim3(phx_grid_for((size_t)9 * Cin * Cout, 256)), dim3(256), 0,
                       (hipStream_t)stream, dw_pad, dw_hwio, Cin, Cin_pad, Cout, 9);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_unpad_filter_grad_center(const float* dw_pad, float* dw_1x1, int Cin, int Cin_pad, int Cout, void* stream) {
    hipLaunchKernelGGL(k_unpad_rows_acc, dim3(phx_grid_for((size_t)Cin * Cout, 256)), dim3(256), 0,
                       (hipStream_t)stream, dw_pad, dw_1x1, Cin, Cin_pad, Cout, 1);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

int phx_debug_set_trace(void* dev_buf) {
    if (int rc = phx_c32_set_trace(dev_buf)) return rc;
    if (int rc = phx_pp_set_trace(dev_buf)) return rc;
    if (int rc = phx_wgrad_set_debug(dev_buf, nullptr, 0)) return rc;
    unsigned long long* p = (unsigned long long*)dev_buf;
    PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phx_trace), &p, sizeof(p)));
    return PHX_OK;
}

#ifdef PHX_DEBUG_BUILD
int phx_debug_conv_policy(int large_maps, int big_tiles) {
    PHX_REQUIRE(large_maps >= 0 && large_maps <= 2 && big_tiles >= 0 && big_tiles <= 2, PHX_E_INVAL, "debug_conv_policy: 0 never, 1 policy, 2 force");
    g_ws_policy = large_maps;
    g_tile_policy = big_tiles;
    return PHX_OK;
}

int phx_debug_pair_kernel_grid(int blocks) {
    PHX_REQUIRE(blocks >= 0, PHX_E_INVAL, "debug_pair_kernel_grid: blocks >= 0 (0: one per CU)");
    return phx_pp_set_grid(blocks);
}
#endif

int phx_debug_set_blocklog(void* dev_buf) {
    if (int rc = phx_wgrad_set_debug(nullptr, dev_buf, 1)) return rc;
    unsigned long long* p = (unsigned long long*)dev_buf;
    PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phx_blocklog), &p, sizeof(p)));
    return PHX_OK;
}

static int q_tiles(int B, int H, int W, int K, int N) {
    MTile g = make_mtile_fwd(B, H, W, K, N);
    return g.tiles_x * g.tiles_y * g.tiles_b;
}

// ... of a concat-free launch (phx_conv3x3_mfma_bf16_dual with x2 / y2): large maps as above, everything else the 256-pixel tiles
// (the DUAL instantiations exist for those only)
static int q_tiles_dual(int B, int H, int W, int K, int N) {
    MTile g = fwd_ws64(B, H, W, K, N) ? make_mtile_fwd(B, H, W, K, N) : make_mtile(B, H, W);
    return g.tiles_x * g.tiles_y * g.tiles_b;
}

// split-K factor for the forward / data-gradient kernel: > 1 only for the 256-pixel-tile kernels on maps whose tiles x
// channel blocks leave most CUs idle (H <= 16 at batch 64); aims at ~64 blocks
static int fwd_ksplit(int B, int H, int W, int K, int N) {
    if (fwd_big_tiles(B, H, W, K, N)) return 1;
    MTile g = make_mtile(B, H, W);
    const int blocks = g.tiles_x * g.tiles_y * g.tiles_b * (N / (N % 64 == 0 ? 64 : 32));
    const int nck = K / KC;
    const int tgt = 64;                              // (re-measured under the two-lane schedule: 32-96 equal, 128+ 0.6 % slower)
    if (blocks * 2 > tgt || nck < 2) return 1;
    int ks = (tgt + blocks - 1) / blocks;
    if (ks > nck) ks = nck;
    const int per = (nck + ks - 1) / ks;
    return (nck + per - 1) / per;                    // no empty slices
}



static size_t q_ws_bytes(int B, int H, int W, int K, int N) {
    const int ks = fwd_ksplit(B, H, W, K, N);
    return ks > 1 ? (size_t)ks * B * H * W * N * sizeof(float) : 0;
}

static int conv3x3_mfma_impl(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial,
                             void* workspace, size_t workspace_bytes, int B, int H, int W, int K, int N, EpiOpts bws, Dual du,
                             void* stream);

// statistics added atomically into sums[N][2] (the layout phx_norm_apply_fused reads): for launches with few pixel tiles (the
// H <= 16 levels), where a same-address atomic per tile and channel (~45 ns each) is cheaper than a reduction launch or a pass of
// its own over y.  Generic 256-pixel-tile kernel only.
static bool fwd_stats_atomic_ok(int B, int H, int W, int K, int N) {
    if (phx_deterministic() || K % KC != 0 || N % 32 != 0) return false;
    if (fwd_ws64(B, H, W, K, N) || fwd_big_tiles(B, H, W, K, N)) return false;
    return q_tiles(B, H, W, K, N) <= 64;
}

// Concat-free forward / data gradient (struct Dual): every option of the entry points above in one call.
//   x2 != NULL: reduction channels [0, K1) are read from x (pixel stride K1), [K1, K) from x2 (stride K - K1), K1 % 32 == 0
//   y2 != NULL: output channels [0, N1) go to y (stride N1), [N1, N) to y2 (stride N - N1), N1 % 8 == 0; no statistics epilogue
//   stats_mode : 0 none, 1 per-tile partial rows stats[tile][2][N], 2 atomically into stats[N][2] (see ..._stats_atomic)
static int l_dual(const void* x, const void* x2, int K1, const void* wpk, void* y, void* y2, int N1, const float* bias,
                               const float* oscale, int act, float* stats, int stats_mode, void* workspace, size_t workspace_bytes,
                               int B, int H, int W, int K, int N, void* stream) {
    PHX_REQUIRE(x2 == nullptr || (K1 > 0 && K1 % 32 == 0 && ((uintptr_t)x2 & 15) == 0), PHX_E_INVAL, "conv3x3_mfma_dual: x2 (16-byte aligned), K1 % 32 == 0");
    PHX_REQUIRE(y2 == nullptr || (N1 > 0 && N1 % 8 == 0 && ((uintptr_t)y2 & 15) == 0), PHX_E_INVAL, "conv3x3_mfma_dual: y2 (16-byte aligned), N1 % 8 == 0");
    PHX_REQUIRE(stats_mode >= 0 && stats_mode <= 2 && (stats_mode == 0) == (stats == nullptr), PHX_E_INVAL, "conv3x3_mfma_dual: stats / stats_mode");
    PHX_REQUIRE(stats_mode != 2 || fwd_stats_atomic_ok(B, H, W, K, N), PHX_E_SHAPE, "conv3x3_mfma_dual: atomic statistics not supported for this shape");
    EpiOpts b{};
    b.stats_atomic = stats_mode == 2;
    b.oscale = oscale;
    Dual du{(const unsigned short*)x2, (unsigned short*)y2, x2 ? K1 : 0, y2 ? N1 : 0};
    return conv3x3_mfma_impl(x, wpk, y, bias, act, stats, workspace, workspace_bytes, B, H, W, K, N, b, du, stream);
}

// fp32 output on small maps: the plain convolution (no bias / activation / statistics) through the split-K instantiations of the
// 256-pixel kernels -- their fp32 accumulators go to the workspace slices and k_splitk_finish sums them into y_f32 without rounding
// (a single slice is written straight into y_f32).  For the one-launch batch norm of the 2 x 2 / 4 x 4 levels (phx_bn_small_fwd with
// x_dt = PHX_F32).  workspace: phx_conv3x3_mfma_ws_bytes (may be NULL / 0 when that is 0).  x2 / K1: concat-free input, as _dual.
static int q_f32out_ok(int B, int H, int W, int K, int N) {
    return (K % KC == 0 && N % 32 == 0 && !fwd_ws64(B, H, W, K, N) && !fwd_big_tiles(B, H, W, K, N) && (double)B * H * W < 16777216.0) ? 1 : 0;
}
static int l_f32out(const void* x, const void* x2, int K1, const void* wpk, float* y_f32, int sum_slices, void* workspace,
                                 size_t workspace_bytes, int B, int H, int W, int K, int N, void* stream) {
    PHX_REQUIRE(y_f32 != nullptr && ((uintptr_t)y_f32 & 15) == 0, PHX_E_INVAL, "conv3x3_mfma_f32out: y_f32 (16-byte aligned) is required");
    PHX_REQUIRE(x2 == nullptr || (K1 > 0 && K1 % 32 == 0 && ((uintptr_t)x2 & 15) == 0), PHX_E_INVAL, "conv3x3_mfma_f32out: x2 (16-byte aligned), K1 % 32 == 0");
    EpiOpts b{};
    b.y_f32 = y_f32;
    b.keep_slices = sum_slices == 0;
    Dual du{(const unsigned short*)x2, nullptr, x2 ? K1 : 0, 0};
    return conv3x3_mfma_impl(x, wpk, nullptr, nullptr, PHX_ACT_ID, nullptr, workspace, workspace_bytes, B, H, W, K, N, b, du, stream);
}

// conv2d on the PRE-normalisation tensor of the producing layer (round 5): y = conv3x3(relu(x * xscale[k] + xshift[k])) -- the 32 -> 32
// layers of the large maps (k_conv3x3_c32, XF: the HBM-bound 128 x 128 level, where the transform is free and the bytes are not).
static int q_xf_ok(int B, int H, int W, int K, int N) {
    return (K == 32 && N == 32 && fwd_ws64(B, H, W, K, N) && phx_c32_enabled()) ? 1 : 0;
}
static int l_xf(const void* x, const float* xscale, const float* xshift, const void* wpk, void* y, float* stats_partial,
                             int B, int H, int W, int K, int N, void* stream) {
    PHX_REQUIRE(q_xf_ok(B, H, W, K, N), PHX_E_SHAPE, "conv3x3_mfma_xf: shape not supported (see phx_conv3x3_xf_supported)");
    PHX_REQUIRE(x && xscale && xshift && wpk && y, PHX_E_INVAL, "conv3x3_mfma_xf: null argument");
    PHX_REQUIRE((((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)y) & 15) == 0, PHX_E_ALIGN, "conv3x3_mfma_xf: 16-byte alignment");
    return phx_c32_launch(x, wpk, y, nullptr, PHX_ACT_ID, stats_partial, B, H, W, nullptr, 0, stream, xscale, xshift);
}

// ---- conv + bias + group / instance norm + activation in one launch (FGN instantiations of k_conv3x3_mfma) ----------------------
// Maps that fit ONE pixel tile (H, W in {2, 4, 8, 16}): a block then holds whole samples and whole 16-channel groups.  -> 32 / 64
// (channels per block), 0: not supported.
static int fgn_plan(int B, int H, int W, int K, int N, int G) {
    if (K % KC != 0 || N % 32 != 0 || G < 1 || !(G == N || G * 16 == N)) return 0;
    if (H > 16 || W > 16 || (H & (H - 1)) || (W & (W - 1)) || H < 2 || W < 2) return 0;
    if ((double)B * H * W >= 16777216.0) return 0;
    MTile g = make_mtile(B, H, W);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    return (N % 64 == 0 && ntiles * (N / 64) > 256) ? 64 : 32;
}

static int l_fgn(const void* x, const void* wpk, void* y, void* a_out, const float* bias, const float* gamma,
                              const float* beta, float eps, int G, int act, float* mean_out, float* rstd_out, float* scale_out,
                              float* shift_out, int B, int H, int W, int K, int N, void* stream) {
    const int bn = fgn_plan(B, H, W, K, N, G);
    PHX_REQUIRE(bn != 0, PHX_E_SHAPE, "conv3x3_mfma_fgn: shape not supported (see phx_conv3x3_fgn_supported)");
    PHX_REQUIRE(x && wpk && y && a_out && gamma && beta && mean_out && rstd_out && scale_out && shift_out, PHX_E_INVAL,
                "conv3x3_mfma_fgn: null argument");
    PHX_REQUIRE((((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)y | (uintptr_t)a_out) & 15) == 0, PHX_E_ALIGN, "conv3x3_mfma_fgn: 16-byte alignment");
    MTile g = make_mtile(B, H, W);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    const int npatch = g.tb * ((1 << g.ths) + 2) * ((1 << g.tws) + 2);
    const int na = (npatch * 4 + 255) / 256;
    PHX_REQUIRE(na <= 16, PHX_E_SHAPE, "conv3x3_mfma_fgn: unexpected tile geometry");
    const bool fast16 = g.tws == 4 && g.ths == 4 && g.tb == 1;
    XForm xf{};
    xf.gamma = gamma; xf.beta = beta; xf.eps = eps; xf.act = act; xf.G = G;
    xf.a_out = (unsigned short*)a_out; xf.mean_out = mean_out; xf.rstd_out = rstd_out; xf.scale_out = scale_out; xf.shift_out = shift_out;
    const int cg = N / G;
#define FGN_LAUNCH(BNv, NAv, Fv)                                                                                        \
    do {                                                                                                                \
        auto kfn = k_conv3x3_mfma<BNv, NAv, Fv, false, 4, false, false, true>;                            \
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   \
        size_t sh = (Fv ? (size_t)(4 * 4 + 2) * PITCH16 : (size_t)g.tb * g.ipitch) + 9 * BNv * ROWB;                    \
        const size_t she = (size_t)256 * BNv * 2 + (size_t)g.tb * (BNv / cg) * 8;                                       \
        if (she > sh) sh = she;                                                                                         \
        hipLaunchKernelGGL(kfn, dim3(ntiles * (N / BNv)), dim3(256), sh, (hipStream_t)stream, (const unsigned short*)x, \
                           (const unsigned short*)wpk, (unsigned short*)y, bias, 0, nullptr, B, H, W, K, N, g, nullptr,  \
                           EpiOpts{}, xf, Dual{});                                                                     \
    } while (0)
    if (bn == 64) {
        if (fast16) FGN_LAUNCH(64, 8, true); else if (na <= 8) FGN_LAUNCH(64, 8, false); else FGN_LAUNCH(64, 16, false);
    } else {
        if (fast16) FGN_LAUNCH(32, 8, true); else if (na <= 8) FGN_LAUNCH(32, 8, false); else FGN_LAUNCH(32, 16, false);
    }
#undef FGN_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

static int conv3x3_mfma_impl(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial,
                             void* workspace, size_t workspace_bytes, int B, int H, int W, int K, int N, EpiOpts bws, Dual du,
                             void* stream) {
    PHX_REQUIRE(K % KC == 0 && N % 32 == 0, PHX_E_SHAPE, "conv3x3_mfma: K % 32 == 0 and N % 32 == 0 required");
    PHX_REQUIRE(du.x2 == nullptr || du.K1 < K, PHX_E_SHAPE, "conv3x3_mfma: dual input needs 0 < K1 < K");
    PHX_REQUIRE(du.y2 == nullptr || (du.N1 < N && (N - du.N1) % 8 == 0 && y != nullptr && stats_partial == nullptr),
                PHX_E_SHAPE, "conv3x3_mfma: dual output needs 0 < N1 < N, (N - N1) % 8 == 0, an output tensor and no statistics epilogue");
    PHX_REQUIRE((((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)y) & 15) == 0, PHX_E_ALIGN, "conv3x3_mfma: 16-byte alignment");
    int ksplit = 1;
    const bool f32out = bws.y_f32 != nullptr;        // fp32 output: always the split-K instantiation (one slice: straight into y_f32)
    if ((workspace && !stats_partial) || f32out) {
        ksplit = fwd_ksplit(B, H, W, K, N);
        PHX_REQUIRE(workspace_bytes >= (size_t)(ksplit > 1 ? ksplit : 0) * B * H * W * N * sizeof(float), PHX_E_INVAL,
                    "conv3x3_mfma: workspace too small");
    }
    if (f32out) {
        PHX_REQUIRE(y == nullptr && !bias && act == PHX_ACT_ID && !bws.oscale && !stats_partial && !du.y2, PHX_E_INVAL,
                    "conv3x3_mfma_f32out: plain convolution only (no bf16 output, bias, activation, statistics or dual output)");
        PHX_REQUIRE(!fwd_ws64(B, H, W, K, N) && !fwd_big_tiles(B, H, W, K, N), PHX_E_SHAPE,
                    "conv3x3_mfma_f32out: small maps only (the 256-pixel split-K kernels)");
        PHX_REQUIRE(ksplit == 1 || workspace != nullptr, PHX_E_INVAL, "conv3x3_mfma_f32out: this shape runs split-K and needs its workspace");
        if (ksplit == 1) workspace = bws.y_f32;
    }
    PHX_REQUIRE(y != nullptr || f32out || (ksplit > 1 && !bias && act == PHX_ACT_ID), PHX_E_INVAL,
                "conv3x3_mfma: y == NULL only for a split-K launch without bias / activation (slices left in the workspace)");
    const bool dual = du.x2 != nullptr || du.y2 != nullptr;
    if (fwd_ws64(B, H, W, K, N)) {
        // large maps (16 x 32-pixel tiles; phx_conv3x3_mfma_bf16_tiles counts those): the 32 -> 32-channel layers take the
        // filter-in-registers kernel (conv_c32.hip), everything else the anti-phase pair kernel (conv_pp.hip)
        PHX_REQUIRE(act == PHX_ACT_ID || act == PHX_ACT_RELU, PHX_E_INVAL, "conv3x3_mfma: large maps take an identity / ReLU epilogue");
        PHX_REQUIRE(!bws.stats_atomic, PHX_E_SHAPE, "conv3x3_mfma: atomic statistics are for launches with few pixel tiles");
        if (K == 32 && N == 32 && !dual && phx_c32_enabled())
            return phx_c32_launch(x, wpk, y, bias, act, stats_partial, B, H, W, bws.oscale, 0, stream);
        return phx_pp_launch(x, wpk, y, bias, act, stats_partial, B, H, W, K, N, bws.oscale, 0, du, 0, stream);
    }
    MTile g = dual ? make_mtile(B, H, W) : make_mtile_fwd(B, H, W, K, N, false);       // (DUAL instantiations exist for the 256-pixel tiles)
    const bool big = !dual && fwd_big_tiles(B, H, W, K, N);
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int npatch = g.tb * (th + 2) * (tw + 2);
    const int ntiles = g.tiles_x * g.tiles_y * g.tiles_b;
    PHX_REQUIRE((double)B * H * W * (K > N ? K : N) < 2147483648.0, PHX_E_SHAPE, "conv3x3_mfma: tensor exceeds 2^31 elements");
    static bool attr_set = false;
#define CM_ATTR1(BNv, NAv, Fv, Av, NWv, Sv)                                                                          \
    PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_mfma<BNv, NAv, Fv, Av, NWv, Sv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define CM_ATTR(BNv, NAv, Fv, NWv) CM_ATTR1(BNv, NAv, Fv, false, NWv, false); CM_ATTR1(BNv, NAv, Fv, true, NWv, false)
    if (!attr_set) {
        CM_ATTR(64, 8, false, 4); CM_ATTR(32, 8, false, 4); CM_ATTR(64, 16, false, 4); CM_ATTR(32, 16, false, 4);
        CM_ATTR(64, 8, true, 4); CM_ATTR(32, 8, true, 4);
        CM_ATTR(128, 5, true, 8); CM_ATTR(64, 5, true, 8); CM_ATTR(32, 5, true, 8);
        CM_ATTR1(64, 8, false, false, 4, true); CM_ATTR1(32, 8, false, false, 4, true); CM_ATTR1(64, 16, false, false, 4, true);
        CM_ATTR1(32, 16, false, false, 4, true); CM_ATTR1(64, 8, true, false, 4, true); CM_ATTR1(32, 8, true, false, 4, true);
        attr_set = true;
    }
#undef CM_ATTR
#undef CM_ATTR1
    const int na = (npatch * 4 + 255) / 256;
    PHX_REQUIRE(na <= 16, PHX_E_SHAPE, "conv3x3_mfma: unexpected tile geometry");
    const bool biasact = bias != nullptr || act != PHX_ACT_ID || bws.oscale != nullptr;
    // (the OROW-pitched epilogue tile also has to fit: NT * (2 BN + 16) bytes)
#define CM_LAUNCH1(BNv, NAv, Fv, Av, NWv, Sv)                                                                          \
    do {                                                                                                             \
        size_t sh = (Fv ? (size_t)(NWv * 4 + 2) * ((NWv == 8 && BNv == 32) ? 18 * ROWB : PITCH16)                    \
                        : (size_t)g.tb * g.ipitch) + 9 * BNv * ROWB;                                                 \
        const size_t she = (size_t)NWv * 64 * (BNv * 2 + 16);                                                        \
        if (she > sh) sh = she;                                                                                      \
        if (Sv)                                                                                                      \
            hipLaunchKernelGGL((k_conv3x3_mfma<BNv, NAv, Fv, false, NWv, Sv>), dim3(ntiles * (N / BNv), 1, ksplit),  \
                               dim3(NWv * 64), sh, (hipStream_t)stream, (const unsigned short*)x,                    \
                               (const unsigned short*)wpk, (unsigned short*)y, nullptr, 0, nullptr, B, H, W, K, N, g,\
                               (float*)workspace, EpiOpts{}, XForm{}, Dual{});                                        \
        else                                                                                                         \
            hipLaunchKernelGGL((k_conv3x3_mfma<BNv, NAv, Fv, Av, NWv, false>), dim3(ntiles * (N / BNv)), dim3(NWv * 64),\
                               sh, (hipStream_t)stream, (const unsigned short*)x, (const unsigned short*)wpk,        \
                               (unsigned short*)y, bias, act, stats_partial, B, H, W, K, N, g, nullptr, bws, XForm{}, Dual{}); \
    } while (0)
#define CM_LAUNCH(BNv, NAv, Fv, NWv)                                                                                 \
    do {                                                                                                             \
        if (NWv == 4 && (ksplit > 1 || f32out)) CM_LAUNCH1(BNv, NAv, Fv, false, NWv, (NWv == 4));                    \
        else if (biasact) CM_LAUNCH1(BNv, NAv, Fv, true, NWv, false);                                                \
        else CM_LAUNCH1(BNv, NAv, Fv, false, NWv, false);                                                            \
    } while (0)
    const bool fast16 = g.tws == 4 && g.ths == 4 && g.tb == 1;
    PHX_REQUIRE(fast16 || big || (double)B * H * W < 16777216.0, PHX_E_SHAPE, "conv3x3_mfma: small-map tiles need B*H*W < 2^24");
    // fewer 64-channel blocks than CUs: every block runs alone on its CU and the launch is one block's latency chain --
    // 32-channel blocks double the block count (two per CU) and halve each block's chain
    const bool narrow32 = N % 64 == 0 && !big && ksplit == 1 && ntiles * (N / 64) <= 256;
#define CM_DUAL1(BNv, NAv, Fv, Av, Sv)                                                                                \
    do {                                                                                                             \
        size_t sh = (Fv ? (size_t)(4 * 4 + 2) * PITCH16 : (size_t)g.tb * g.ipitch) + 9 * BNv * ROWB;                  \
        const size_t she = (size_t)256 * (BNv * 2 + 16);                                                             \
        if (she > sh) sh = she;                                                                                      \
        auto kfd = k_conv3x3_mfma<BNv, NAv, Fv, Av, 4, Sv, true>;                                                    \
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)kfd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL(kfd, dim3(ntiles * (N / BNv), 1, Sv ? ksplit : 1), dim3(256), sh, (hipStream_t)stream,     \
                           (const unsigned short*)x, (const unsigned short*)wpk, (unsigned short*)y, Sv ? nullptr : bias, \
                           Sv ? 0 : act, Sv ? nullptr : stats_partial, B, H, W, K, N, g, Sv ? (float*)workspace : nullptr, \
                           Sv ? EpiOpts{} : bws, XForm{}, du);                                                       \
    } while (0)
#define CM_DUAL(BNv, NAv, Fv)                                                                                        \
    do {                                                                                                             \
        if (ksplit > 1 || f32out) CM_DUAL1(BNv, NAv, Fv, false, true);                                               \
        else if (biasact) CM_DUAL1(BNv, NAv, Fv, true, false);                                                       \
        else CM_DUAL1(BNv, NAv, Fv, false, false);                                                                   \
    } while (0)
    if (dual) {                                      // concat-free input / output (struct Dual): the DUAL instantiations
        if (N % 64 == 0 && !narrow32) {
            if (fast16) CM_DUAL(64, 8, true); else if (na <= 8) CM_DUAL(64, 8, false); else CM_DUAL(64, 16, false);
        } else {
            if (fast16) CM_DUAL(32, 8, true); else if (na <= 8) CM_DUAL(32, 8, false); else CM_DUAL(32, 16, false);
        }
    } else if (big) {
        if (N % 128 == 0) CM_LAUNCH(128, 5, true, 8); else if (N % 64 == 0) CM_LAUNCH(64, 5, true, 8); else CM_LAUNCH(32, 5, true, 8);
    } else if (N % 64 == 0 && !narrow32) {
        if (fast16) CM_LAUNCH(64, 8, true, 4); else if (na <= 8) CM_LAUNCH(64, 8, false, 4); else CM_LAUNCH(64, 16, false, 4);
    } else {
        if (fast16) CM_LAUNCH(32, 8, true, 4); else if (na <= 8) CM_LAUNCH(32, 8, false, 4); else CM_LAUNCH(32, 16, false, 4);
    }
#undef CM_LAUNCH
#undef CM_LAUNCH1
#undef CM_DUAL
#undef CM_DUAL1
    PHX_CHECK_LAUNCH();
    if (ksplit > 1 && (y != nullptr || bws.y_f32 != nullptr) && !bws.keep_slices) {      // (else: the caller consumes the fp32 slices itself)
        const size_t total = (size_t)B * H * W * N;
        hipLaunchKernelGGL(k_splitk_finish, dim3(phx_grid_for(total / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)workspace, ksplit, total, N, bias, act, (unsigned short*)y, bws.oscale, du.y2, du.N1, bws.y_f32);
        PHX_CHECK_LAUNCH();
    }
    return PHX_OK;
}


// ---- the ONE launch entry and the ONE plan query of the bf16 forward / data-gradient family (include/phx.h) ------------------------
static_assert(sizeof(phx_conv3x3_desc) == 224, "phx_conv3x3_desc is packed by the host (phiseg_code_amd/runtime.py): 224 bytes");
int phx_conv3x3_desc_bytes(void) { return (int)sizeof(phx_conv3x3_desc); }

int phx_conv3x3_bf16_plan(int B, int H, int W, int K, int N, int G, phx_conv3x3_plan* out) {
    PHX_REQUIRE(out != nullptr, PHX_E_INVAL, "conv3x3_bf16_plan: out is required");
    memset(out, 0, sizeof(*out));
    if (B < 1 || H < 1 || W < 1 || K < 32 || N < 32 || K % 32 != 0 || N % 32 != 0) return PHX_OK;       // (everything 0: not an MFMA shape)
    out->tiles = q_tiles(B, H, W, K, N);
    out->tiles_dual = q_tiles_dual(B, H, W, K, N);
    out->ksplit = fwd_ksplit(B, H, W, K, N);
    out->ws_bytes = q_ws_bytes(B, H, W, K, N);
    out->stats_atomic_ok = fwd_stats_atomic_ok(B, H, W, K, N) ? 1 : 0;
    out->f32out_ok = q_f32out_ok(B, H, W, K, N);
    out->xf_ok = q_xf_ok(B, H, W, K, N);
    out->fgn_block = G > 0 ? fgn_plan(B, H, W, K, N, G) : 0;
    return PHX_OK;
}

int phx_conv3x3_bf16(const phx_conv3x3_desc* d, void* stream) {
    PHX_REQUIRE(d != nullptr, PHX_E_INVAL, "conv3x3_bf16: descriptor is required");
    if (d->gn_groups > 0) {
        PHX_REQUIRE(d->x2 == nullptr && d->y2 == nullptr && d->xscale == nullptr && d->y_f32 == nullptr && d->oscale == nullptr
                    && d->stats_mode == PHX_CONV_STATS_NONE, PHX_E_INVAL, "conv3x3_bf16: the fused group-norm epilogue takes the plain input / output only");
        return l_fgn(d->x, d->wpk, d->y, d->a_out, d->bias, d->gamma, d->beta, d->gn_eps, d->gn_groups, d->act, d->mean_out, d->rstd_out,
                     d->scale_out, d->shift_out, d->B, d->H, d->W, d->K, d->N, stream);
    }
    if (d->xscale != nullptr || d->xshift != nullptr) {
        PHX_REQUIRE(d->x2 == nullptr && d->y2 == nullptr && d->y_f32 == nullptr && d->bias == nullptr && d->oscale == nullptr && d->act == PHX_ACT_ID
                    && d->stats_mode != PHX_CONV_STATS_ATOMIC, PHX_E_INVAL, "conv3x3_bf16: the transforming loader takes the plain epilogue only");
        return l_xf(d->x, d->xscale, d->xshift, d->wpk, d->y, d->stats_mode == PHX_CONV_STATS_PARTIAL ? d->stats : nullptr, d->B, d->H, d->W,
                    d->K, d->N, stream);
    }
    if (d->y_f32 != nullptr) {
        PHX_REQUIRE(d->y == nullptr && d->y2 == nullptr && d->bias == nullptr && d->oscale == nullptr && d->act == PHX_ACT_ID
                    && d->stats_mode == PHX_CONV_STATS_NONE, PHX_E_INVAL, "conv3x3_bf16: the fp32 output takes no epilogue");
        PHX_REQUIRE(q_f32out_ok(d->B, d->H, d->W, d->K, d->N), PHX_E_SHAPE, "conv3x3_bf16: fp32 output not supported for this shape (plan.f32out_ok)");
        return l_f32out(d->x, d->x2, d->K1, d->wpk, d->y_f32, d->sum_slices, d->workspace, d->workspace_bytes, d->B, d->H, d->W, d->K, d->N, stream);
    }
    return l_dual(d->x, d->x2, d->K1, d->wpk, d->y, d->y2, d->N1, d->bias, d->oscale, d->act, d->stats, d->stats_mode, d->workspace,
                  d->workspace_bytes, d->B, d->H, d->W, d->K, d->N, stream);
}

}  // extern "C"
