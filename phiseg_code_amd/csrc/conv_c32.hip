// bf16 MFMA 3x3 convolution for gfx950, forward / data-gradient of the 32 -> 32-channel layers on large maps (the 128 x 128 level of
// PHiSeg's posterior / prior / likelihood: tfwrapper/layers.py:123 with num_filters = 32, and the data gradient TF derives for it).
//
// These layers sit on the HBM side of the ridge (144 FLOP/B of tensor traffic: 134 MB per launch at batch 64 against 19 GFLOP), and
// in the general LDS-DMA kernel (k_conv3x3_fwd_dma128<BN = 32>) they are instruction-bound: every one of the 4 096 blocks stages the
// 18 KiB filter slab again (75 MB of L2 -> LDS traffic per launch, as much as the input tensor), reads three filter fragments per
// twelve MFMAs from LDS and pays prologue + epilogue once per 16 x 32-pixel tile (39 us per launch = 3.4 TB/s).
// Here the WHOLE filter lives in registers (9 taps x 2 k-steps x one 32 x 16 fragment = 72 VGPRs per lane, loaded once per block),
// blocks are persistent over a strided list of tiles, and the patch of tile t + 1 goes global -> LDS by DMA (buffer_load ... lds)
// into the second stage while tile t is multiplied and stored: per tile a wave issues 10 DMA instructions, 36 ds_read_b128 and 72
// MFMAs; nothing but the 39 KiB patch crosses LDS.  The finished tile is transposed through the stage it was computed from
// (dense 64-byte rows: conflict-free for the 32-bit writes and for the ds_read_b128 lane groups) and stored 16 bytes per lane.
// Two blocks per CU (2 x 80 KiB of LDS): one block's stores and DMA latency run under the other's MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "phx_common.h"

namespace {

constexpr int C32_AI = 39;                   // 1 KiB DMA instructions per patch: 612 pixels (18 x 34) x 64 B
constexpr int C32_NPL = 10;                  // ... per wave (wave w issues w, w + 4, ...)
constexpr int C32_STAGE = 40 * 1024;         // stage pitch (the 32 KiB output tile reuses the stage it was computed from)
constexpr int C32_PROW = 34 * 64;            // bytes per patch row
constexpr int C32_LDS = 2 * C32_STAGE;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS reads behind the compiler's back: while a buffer_load ... lds is in flight hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front
// of the first ds_read that follows an asm barrier -- which would drain the next tile's patch before this tile's stores.  The
// epilogue therefore reads through inline asm and waits on lgkmcnt itself.
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

__device__ unsigned long long* g_c32_trace = nullptr;      // dev: cycle stamps of block 0, wave 0 (phx_debug_set_trace)
#define C32_TRACE(slot)                                                                          \
    do {                                                                                         \
        if (g_c32_trace && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 512)                   \
            g_c32_trace[slot] = __builtin_readcyclecounter();                                     \
    } while (0)

// XF (round 5; phx_conv3x3_mfma_bf16_xf): x is the PRE-normalisation tensor of the layer in front and this kernel re-forms that layer's
// a = relu(x * xscale[k] + xshift[k]) itself (conv2d -> batch_norm -> relu -> conv2d, tfwrapper/layers.py:123-135: the producer's apply
// pass and its activation tensor are never made).  The patch still arrives by LDS-DMA; when it has landed every thread rewrites ten
// 16-byte pieces of it IN PLACE (ds_read_b128 -> unpack, fma, max, pack -> ds_write_b128; a thread keeps one source piece = eight
// channels, 16 coefficient registers) and out-of-image pieces are forced back to zero -- the padding is of a, not of x.  This kernel
// is HBM-bound (matrix pipe ~18 % busy): the extra VALU / LDS work runs under the other block's loads, and the bytes it removes are
// the ones that bound the level (the lesson of the pair-kernel attempt, DESIGN.md section 5: it pays only on the HBM side of the ridge).
template <bool BIASACT, bool XF = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_c32(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wpk,
                                                        unsigned short* __restrict__ y, const float* __restrict__ bias, int act,
                                                        float* __restrict__ stats_partial, int B, int H, int W, int tiles_x,
                                                        int tiles_y, int ntiles, const float* __restrict__ oscale, int stats_nrep, unsigned mgx, unsigned mgy,
                                                        const float* __restrict__ xscale = nullptr, const float* __restrict__ xshift = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, khalf = lane >> 5;

    // the filter: B fragment of (tap, k-step) = 16 bytes of packed row [tap][n = l31], channels 16 ks + 8 khalf .. + 7
    bf16x8 fb[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            fb[tap][ks] = *reinterpret_cast<const bf16x8*>(wpk + ((tap * 32 + l31) * 32 + (ks * 2 + khalf) * 8));

    // DMA plan of a patch (tile independent part): lane `lane` of instruction j = wave + 4 n fills LDS slot e = 64 j + lane with
    // source piece `piece` of patch pixel (px, py); the 16-byte slot index is XORed with bits 2-3 of the patch column on the SOURCE
    // side (the LDS destination of buffer_load ... lds is lane-linear), so every ds_read_b128 lane group covers a whole bank row
    // Per piece only two values depend on the tile: the base offset (scalar) and whether the piece lies outside the image, which it
    // can only do in patch column 0 / 33 or patch row 0 / 17 of a tile on the matching image edge: rel[n] = byte offset relative to
    // patch pixel (0, 0), edge[n] = {column 0, column 33, row 0, row 17} bits (15: no such piece) -- 4 VALU instructions per DMA.
    unsigned rel[C32_NPL], edge[C32_NPL];
#pragma unroll
    for (int n = 0; n < C32_NPL; ++n) {
        const int j = wave + 4 * n;
        const int e = j * 64 + lane, pp = e >> 2, slot = e & 3;
        const int py = pp / 34, px = pp - py * 34;
        rel[n] = (unsigned)((py * W + px) * 64 + ((slot ^ ((px >> 2) & 3)) << 4));
        edge[n] = (j < C32_AI && pp < 612) ? (unsigned)((px == 0) | ((px == 33) << 1) | ((py == 0) << 2) | ((py == 17) << 3)) : 15u;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * 64u), 0x00020000);
    // tile index -> (x0, y0, image) by multiply-high (mgx = ceil(2^32 / tiles_x), mgy = ceil(2^32 / tiles_y): exact for t < 2^16)
    auto decode = [&](int t, int* tx0, int* ty0, int* b0) {
        const int q1 = tiles_x == 1 ? t : (int)__umulhi((unsigned)t, mgx);         // (the magic number of 1 does not fit 32 bits)
        *tx0 = (t - q1 * tiles_x) << 5;
        const int q2 = tiles_y == 1 ? q1 : (int)__umulhi((unsigned)q1, mgy);
        *ty0 = (q1 - q2 * tiles_y) << 4;
        *b0 = q2;
    };
    unsigned pbase = 0, pout = 0;                      // patch being fetched: base offset, image-edge bits
    int pstage = 0;
    auto patch_setup = [&](int t, int stage) {
        int tx0, ty0, b0;
        decode(t, &tx0, &ty0, &b0);
        pbase = (unsigned)(((b0 * H + ty0 - 1) * W + tx0 - 1) * 64);                       // (may wrap below zero: added modulo 2^32)
        pout = (tx0 == 0 ? 1u : 0u) | (tx0 + 32 >= W ? 2u : 0u) | (ty0 == 0 ? 4u : 0u) | (ty0 + 16 >= H ? 8u : 0u);
        pstage = stage;
    };
    auto issue_piece = [&](auto nc) {                  // DMA instruction n of this wave for the patch set up last
        constexpr int n = decltype(nc)::value;
        const int j = wave + 4 * n;
        if (j < C32_AI) {
            const bool bad = edge[n] == 15u || (edge[n] & pout) != 0u;
            const unsigned vo = bad ? 0xffffffffu : pbase + rel[n];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + pstage * C32_STAGE + j * 1024), 16, (int)vo, 0, 0, 0);
        }
    };
    auto issue_range = [&](auto self, auto lo, auto hi) {
        constexpr int l = decltype(lo)::value, h = decltype(hi)::value;
        if constexpr (l < h && l < C32_NPL) {
            issue_piece(lo);
            self(self, std::integral_constant<int, l + 1>(), hi);
        }
    };
    // A fragment of (tile row r, tap row kh), tap column kw, k-step ks = patch row 4 wave + r + kh, pixel l31 + kw
    unsigned aK[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aK[kw][ks] = (unsigned)((wave * 4 * 34 + l31 + kw) * 64 + (((ks * 2 + khalf) ^ (((l31 + kw) >> 2) & 3)) << 4));
    float bv = 0.f, sv = 1.f;
    if constexpr (BIASACT) {
        bv = bias ? bias[l31] : 0.f;
        sv = oscale ? oscale[l31] : 1.f;
    }
    const int odd = lane & 1;

    int slot = blockIdx.x, t = phx_band8(slot, ntiles);       // (slot -> tile: conv_common.h, XCD bands)
    if (slot < ntiles) {
        patch_setup(t, 0);
        issue_range(issue_range, std::integral_constant<int, 0>(), std::integral_constant<int, C32_NPL>());
    }
    for (int it = 0; slot < ntiles; slot += gridDim.x, t = phx_band8(slot, ntiles), ++it) {
        const int cur = it & 1;
        const unsigned char* st = smem + cur * C32_STAGE;
        // this tile's patch (issued a whole tile ago) has landed -- vmcnt retires in order and only the previous tile's 8 (wave 0: 9)
        // stores are younger than its DMA instructions, so they may stay in flight -- and the other stage is free: every wave has
        // issued the stores that read it.  (Raw barriers throughout: __syncthreads() would drain the DMA in flight with vmcnt(0).)
        C32_TRACE(it * 8 + 0);
        // (first tile of the block: nothing but the DMA instructions is outstanding, so "all but 8" would not cover them)
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        C32_TRACE(it * 8 + 1);
        if constexpr (XF) {
            // in-place transform of the landed patch: thread -> (source piece sp = tid & 3, patch pixels (tid >> 2) + 64 n)
            int tx0c, ty0c, b0c;
            decode(t, &tx0c, &ty0c, &b0c);
            const bool e_l = tx0c == 0, e_r = tx0c + 32 >= W, e_t = ty0c == 0, e_b = ty0c + 16 >= H;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));                  // (opaque: nothing of this pass is hoisted out of the tile loop and kept live across the MFMAs)
            const int sp = tid & 3;
            unsigned char* const stw = smem + cur * C32_STAGE;
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            typedef __attribute__((ext_vector_type(4))) float f32x4_t;
            // (the sixteen coefficients are re-loaded per tile -- L1 hits -- rather than held: the kernel has no registers to spare)
            const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(xscale + sp * 8), c1 = *reinterpret_cast<const f32x4_t*>(xscale + sp * 8 + 4);
            const f32x4_t d0 = *reinterpret_cast<const f32x4_t*>(xshift + sp * 8), d1 = *reinterpret_cast<const f32x4_t*>(xshift + sp * 8 + 4);
            const float xsc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            const float xsh[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                const int pp = (tid >> 2) + 64 * n;
                if (pp < 612) {
                    const int py = (int)(((unsigned)pp * 1928u) >> 16), px = pp - py * 34;      // pp / 34, pp % 34
                    const bool outside = (px == 0 && e_l) || (px == 33 && e_r) || (py == 0 && e_t) || (py == 17 && e_b);
                    u32x4_t* const q = reinterpret_cast<u32x4_t*>(stw + pp * 64 + ((sp ^ ((px >> 2) & 3)) << 4));
                    const u32x4_t r = *q;
                    u32x4_t o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float a0 = fmaxf(fmaf(__uint_as_float(r[k] << 16), xsc[2 * k], xsh[2 * k]), 0.f);
                        const float a1 = fmaxf(fmaf(__uint_as_float(r[k] & 0xffff0000u), xsc[2 * k + 1], xsh[2 * k + 1]), 0.f);
                        o[k] = outside ? 0u : f2bf_pk(a0, a1);
                    }
                    *q = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // the next tile's patch: two DMA instructions behind the MFMAs of each of the first five groups (issued back to back they
        // take ~200 cycles apiece out of this wave's instruction stream; the matrix pipe drains its queue meanwhile)
        const bool more = slot + (int)gridDim.x < ntiles;
        if (more) patch_setup(phx_band8(slot + (int)gridDim.x, ntiles), cur ^ 1);
        C32_TRACE(it * 8 + 2);

        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // six (k-step, tap column) groups: the six patch rows of a group are read once and feed twelve MFMAs (4 tile rows x 3 tap
        // rows); the next group's reads are pinned ahead of this group's MFMAs
        bf16x8 fa[2][6];
        auto read_a = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int ks = g / 3, kw = g % 3;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr)
                fa[g & 1][rr] = *reinterpret_cast<const bf16x8*>(st + aK[kw][ks] + rr * C32_PROW);
        };
        read_a(std::integral_constant<int, 0>());
        auto groups = [&](auto self, auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 6) {
                constexpr int ks = g / 3, kw = g % 3;
                if constexpr (g + 1 < 6) read_a(std::integral_constant<int, g + 1>());
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][i + kh], fb[kh * 3 + kw][ks], acc[i], 0, 0, 0);
                if (more) issue_range(issue_range, std::integral_constant<int, 2 * g>(), std::integral_constant<int, 2 * g + 2>());
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, g + 1>());
            }
        };
        groups(groups, std::integral_constant<int, 0>());
        C32_TRACE(it * 8 + 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading the patch
        C32_TRACE(it * 8 + 4);

        // epilogue: C layout col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 khalf (pixel of the tile row).  Lane pairs
        // (channels n, n + 1) trade rows by DPP so that a lane writes one 32-bit word {ch n, ch n + 1}; [pixel][32 ch] bf16, 64-byte rows
        // (v_perm_b32 builds the word from this lane's and the neighbour's packed pairs in one instruction; the statistics run on
        // packed fp32 pairs: the pack loop is the VALU-bound part of this kernel -- 17 -> 8 instructions per word)
        if constexpr (BIASACT) {                     // (the activation code is uniform per launch: one scalar branch, not two per element)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = fmaf(acc[i][r], sv, bv);
            if (act == PHX_ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(acc[i][r], 0.f);
            } else if (act != PHX_ACT_ID) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = act_fwd(acc[i][r], act);
            }
        }
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        f32x2_t s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        const unsigned psel = odd ? 0x03020706u : 0x05040100u;       // even lane: {own lo, neighbour lo}; odd: {neighbour hi, own hi}
        unsigned char* lwp = smem + cur * C32_STAGE + (wave * 128 + 4 * khalf + odd) * 64 + (l31 & ~1) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const int r0 = 2 * rp;
                float v0 = acc[i][r0], v1 = acc[i][r0 + 1];
                const unsigned w2 = f2bf_pk(v0, v1);
                if (stats_partial) {
                    const f32x2_t rv = {__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xffff0000u)};
                    s1v += rv;
                    s2v += rv * rv;
                }
                const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w2, 0xB1, 0xf, 0xf, true);
                *reinterpret_cast<unsigned*>(lwp + (i * 32 + (r0 & 3) + 8 * (r0 >> 2)) * 64) = __builtin_amdgcn_perm(nb, w2, psel);
            }
        C32_TRACE(it * 8 + 5);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            int tx0, ty0, b0;
            decode(t, &tx0, &ty0, &b0);
            const int mt = threadIdx.x >> 2, q = threadIdx.x & 3;         // 64 pixels (two tile rows) x four 16-byte pieces per step
            const unsigned char* lr = smem + cur * C32_STAGE + mt * 64 + q * 16;
            unsigned short* yp = y + (((size_t)b0 * H + ty0 + (mt >> 5)) * W + tx0 + (mt & 31)) * 32 + q * 8;
            const size_t ystep = (size_t)2 * W * 32;
            const unsigned la = lds_offset(lr);
            uint4 v[8];
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\t"
                         "ds_read_b128 %3, %8 offset:12288\n\tds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %8 offset:20480\n\t"
                         "ds_read_b128 %6, %8 offset:24576\n\tds_read_b128 %7, %8 offset:28672\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(la)
                         : "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<uint4*>(yp + k * ystep) = v[k];
        }
        C32_TRACE(it * 8 + 6);
        if (stats_partial) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the tile has been read: its stage becomes reduction scratch
            float* red = reinterpret_cast<float*>(smem + cur * C32_STAGE);      // [4 waves][2][32]
            const float s1 = s1v[0] + s1v[1], s2 = s2v[0] + s2v[1];
            const float a = s1 + __shfl_xor(s1, 32, 64);
            const float bq = s2 + __shfl_xor(s2, 32, 64);
            if (khalf == 0) {
                red[(wave * 2 + 0) * 32 + l31] = a;
                red[(wave * 2 + 1) * 32 + l31] = bq;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x < 64) {
                const int which = threadIdx.x >> 5, n = threadIdx.x & 31;
                float r0, r1, r2, r3;
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\t"
                             "ds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                             : "v"(lds_offset(red + which * 32 + n))
                             : "memory");
                const float v = (r0 + r1) + (r2 + r3);
                // stats_nrep > 0: stats_partial is the replicated accumulator sums[rep][32][2] (tile t -> replica t % stats_nrep, atomics)
                if (stats_nrep > 0) atomicAdd(&stats_partial[((size_t)(t % stats_nrep) * 32 + n) * 2 + which], v);
                else stats_partial[((size_t)t * 2 + which) * 32 + n] = v;
            }
        }
    }
}

}  // namespace

int phx_c32_set_trace(void* dev_buf) {
    PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_c32_trace), &dev_buf, sizeof(void*)));
    return PHX_OK;
}
bool phx_c32_enabled() { return true; }

// same contract as the pair-kernel launch inside conv3x3_mfma_impl (K = N = 32, H % 16 == 0, W % 32 == 0): statistics as
// per-tile partial rows [tile][2][32] (stats_nrep == 0) or added to stats_nrep accumulator replicas
int phx_c32_launch(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial, int B, int H, int W,
                   const float* oscale, int stats_nrep, void* stream, const float* xscale, const float* xshift) {
    const int tiles_x = W / 32, tiles_y = H / 16, ntiles = B * tiles_x * tiles_y;
    const int cap = 512;                          // persistent grid: two blocks per CU
    const int grid = ntiles < cap ? ntiles : cap;
    const bool ba = bias != nullptr || act != PHX_ACT_ID || oscale != nullptr;
    PHX_REQUIRE(ntiles < 65536, PHX_E_SHAPE, "conv3x3_c32: more than 65535 tiles");
    const unsigned mgx = (unsigned)((0x100000000ull + tiles_x - 1) / tiles_x), mgy = (unsigned)((0x100000000ull + tiles_y - 1) / tiles_y);
    if (xscale != nullptr) {
        PHX_REQUIRE(xshift != nullptr && !ba, PHX_E_INVAL, "conv3x3_c32: the input transform takes a plain epilogue");
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_c32<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((k_conv3x3_c32<false, true>), dim3(grid), dim3(256), C32_LDS, (hipStream_t)stream, (const unsigned short*)x,
                           (const unsigned short*)wpk, (unsigned short*)y, bias, act, stats_partial, B, H, W, tiles_x, tiles_y, ntiles,
                           oscale, stats_nrep, mgx, mgy, xscale, xshift);
    } else if (ba) {
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_c32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((k_conv3x3_c32<true>), dim3(grid), dim3(256), C32_LDS, (hipStream_t)stream, (const unsigned short*)x,
                           (const unsigned short*)wpk, (unsigned short*)y, bias, act, stats_partial, B, H, W, tiles_x, tiles_y, ntiles,
                           oscale, stats_nrep, mgx, mgy, (const float*)nullptr, (const float*)nullptr);
    } else {
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_c32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((k_conv3x3_c32<false>), dim3(grid), dim3(256), C32_LDS, (hipStream_t)stream, (const unsigned short*)x,
                           (const unsigned short*)wpk, (unsigned short*)y, bias, act, stats_partial, B, H, W, tiles_x, tiles_y, ntiles,
                           oscale, stats_nrep, mgx, mgy, (const float*)nullptr, (const float*)nullptr);
    }
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
