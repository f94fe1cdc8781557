// bf16 MFMA 3x3 convolution for gfx950, forward / data-gradient, large maps: double-buffered persistent kernel.
// Replaces tf.nn.conv2d 3x3 SAME (tfwrapper/layers.py:123) and the data gradient TF derives for it on maps with
// W % 32 == 0, H % 16 == 0, N % 64 == 0 (the 128x128 / 64x64 levels of PHiSeg at batch 64).
//
// Why this shape (measured on MI355X, tools/bench_fwd_ablate.py): in every "two independent blocks per CU" design the
// global -> LDS staging time and the MFMA time of a block ADD (loads-only 0.130 ms + MFMAs-only 0.182 ms = 0.311 ms for
// 128 -> 128 @ 128 x 128 at batch 64 in k_conv3x3_fwd_dma128): the blocks of a CU run load -> compute in lock-step.  Here ONE
// 4-wave block per CU -- one wave per SIMD, the whole 512-entry register file per lane -- owns two 75 KiB LDS stages:
//
//     chunk s   : 144 MFMAs per wave out of stage s & 1, with the 19 LDS-DMA instructions (buffer_load ... lds, 1 KiB each) of
//                 chunk s + 1 issued BETWEEN them into stage (s + 1) & 1; one s_waitcnt vmcnt(0) + s_barrier per chunk
//
// so the matrix pipe only waits for the issue slots of the DMA instructions, never for their data.  Blocks are persistent (a
// strided list of (16 x 32-pixel tile, 64-channel block) items; the chunk sequence runs across item boundaries), the output
// tile of a finished item is packed, transposed 16 pixels at a time through a private 2.25 KiB scratch per wave (no block-level
// synchronisation) and stored at the head of the next item's first chunk, after the barrier, so that its stores complete under
// that chunk's MFMAs.  A wave owns four tile rows (4 x 32 pixels) x 64 channels: an MFMA's 32 pixels are one tile row, so the A
// fragment of (row r, tap row kh) is patch row r + kh -- six patch-row reads serve the twelve (row, kh) pairs of a (kw, k-step)
// group and every filter fragment feeds four MFMAs: 12 ds_read_b128 per 24 MFMAs.
//
// LDS (159 KiB): stage 0 | stage 1 (each: 39 KiB patch of 18 x 34 pixels x 32 channels, 36 KiB slab of 9 taps x 64 channels x 32)
// | 4 x 2.25 KiB epilogue scratch.  Rows are 64 bytes = four 16-byte slots, slot ^= bits 2-3 of the patch column / channel
// (applied on the DMA source side: the LDS destination of buffer_load ... lds is lane-linear), which makes every ds_read_b128
// lane group cover all 16 slots of a 256-byte bank row exactly once.
#include <stdlib.h>

#include <type_traits>

#include "phx_common.h"

namespace {

constexpr int DB_AI = 39, DB_NI = 75;          // 1 KiB DMA instructions per chunk: 612 patch rows x 64 B (39), 576 slab rows (36)
constexpr int DB_NPL = 19;                     // DMA instructions per wave and chunk (wave w issues w, w + 4, ...)
constexpr int DB_A_BYTES = DB_AI * 1024;
constexpr int DB_STAGE = DB_NI * 1024;
constexpr int DB_OFF_SCR = 2 * DB_STAGE;
constexpr int DB_SCR_WAVE = 16 * 144;          // 16 pixels x (64 channels x 2 B + 16 B pad)
constexpr int DB_LDS_BYTES = DB_OFF_SCR + 4 * DB_SCR_WAVE;     // 162 816 <= 163 840
constexpr int DB_PROW = 34 * 64;               // bytes per patch row

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ unsigned long long* g_phx_db_trace = nullptr;      // dev: cycle stamps of block 0, wave 0 (phx_debug_set_trace)
#define DB_TRACE(slot)                                                                       \
    do {                                                                                     \
        if (g_phx_db_trace && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 4096)          \
            g_phx_db_trace[slot] = __builtin_readcyclecounter();                             \
    } while (0)

struct DBGeom {
    int tiles_x, tiles_y, ntiles, ncob, nitems;
};

// DBG (dev, tools/bench_fwd_ablate.py): bit 1 no patch loads, 2 no slab loads, 4 no MFMAs, 8 no output stores
// DPS: DMA instructions issued per 12-MFMA half-step (the 19 of a chunk go out in the first ceil(19 / DPS) half-steps)
template <bool BIASACT, int DBG, int DPS>
__global__ __launch_bounds__(256, 1) void k_conv3x3_fwd_db(const unsigned short* __restrict__ x,
                                                          const unsigned short* __restrict__ wpk,
                                                          unsigned short* __restrict__ y, const float* __restrict__ bias,
                                                          int act, float* __restrict__ stats_partial, int B, int H, int W, int K,
                                                          int N, DBGeom gm, const float* __restrict__ oscale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int nch = K / 32;
    const int nk = (gm.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // items of this block
    const int T = nk * nch;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, (int)(9u * N * K * 2u), 0x00020000);

    f32x16 acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // ---- this lane's share of a chunk's DMA: instruction n of wave w is j = w + 4 n; j < 39: 64 consecutive 16-byte pieces of
    // the patch (4 per pixel), else of the filter slab (4 per (tap, channel) row).  Item-independent part of the source offset
    // and the patch-border class of the piece are computed once; an item adds its base / masks its borders (set_item).
    int rel[DB_NPL];                           // n < 10: patch instruction j = w + 4 n (j < 39); n >= 10: slab instruction w + 4 (n - 10)
    unsigned ebits[10];                        // patch piece: 1 column 0, 2 column 33, 4 row 0, 8 row 17, 16 beyond the patch
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        const int j = wave + 4 * n;
        const int e = j * 64 + lane, pp = e >> 2, slot = e & 3;
        const int py = pp / 34, px = pp - py * 34;
        const int piece = slot ^ ((px >> 2) & 3);
        rel[n] = (((py - 1) * W + (px - 1)) * K) * 2 + piece * 16;
        ebits[n] = (px == 0 ? 1u : 0u) | (px == 33 ? 2u : 0u) | (py == 0 ? 4u : 0u) | (py == 17 ? 8u : 0u) | (pp >= 612 ? 16u : 0u);
    }
#pragma unroll
    for (int n = 10; n < DB_NPL; ++n) {
        const int e = (wave + 4 * (n - 10)) * 64 + lane, rb = e >> 2, slot = e & 3;
        const int tap = rb >> 6, nn = rb & 63;
        const int piece = slot ^ ((nn >> 2) & 3);
        rel[n] = ((tap * N + nn) * 32 + piece * 8) * 2;
    }
    unsigned voff[DB_NPL];
    int tx0 = 0, ty0 = 0, b0 = 0, n0 = 0;              // the item being STAGED
    // item k of this block.  Linear item id -> (tile, channel block): the N / 64 channel blocks of a tile get ids 8 apart (same
    // XCD, same round of the persistent grid), so the patch they share comes from HBM once.
    auto set_item = [&](int k) {
        const int id = (int)blockIdx.x + k * (int)gridDim.x;
        const int full = (gm.ntiles >> 3) * 8 * gm.ncob;
        int tile, cob;
        if (id < full) {
            const int g8 = id / (8 * gm.ncob), r = id - g8 * 8 * gm.ncob;
            tile = g8 * 8 + (r & 7);
            cob = r >> 3;
        } else {
            const int rem = gm.ntiles & 7, r = id - full;
            tile = (gm.ntiles & ~7) + r % rem;
            cob = r / rem;
        }
        int t = tile;
        tx0 = (t % gm.tiles_x) << 5; t /= gm.tiles_x;
        ty0 = (t % gm.tiles_y) << 4; t /= gm.tiles_y;
        b0 = t;
        n0 = cob * 64;
        const unsigned base = (unsigned)((((b0 * H + ty0) * W + tx0) * K) * 2);
        const unsigned edge = 16u | (tx0 == 0 ? 1u : 0u) | (tx0 + 32 == W ? 2u : 0u) | (ty0 == 0 ? 4u : 0u) |
                              (ty0 + 16 == H ? 8u : 0u);
#pragma unroll
        for (int n = 0; n < 10; ++n) voff[n] = ((DBG & 1) || (ebits[n] & edge)) ? 0xffffffffu : base + (unsigned)rel[n];   // outside: the range check returns zeros
#pragma unroll
        for (int n = 10; n < DB_NPL; ++n) voff[n] = (DBG & 2) ? 0xffffffffu : (unsigned)(n0 * 64 + rel[n]);
    };
    // DMA instruction n of channel chunk c of the staged item, into the stage at LDS byte offset so
    auto issue = [&](auto nc, int c, unsigned so) {
        constexpr int n = decltype(nc)::value;
        if constexpr (DBG & 16) return;        // dev: no DMA instructions at all
        if constexpr (n < 9)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + so + (wave + 4 * n) * 1024), 16, (int)voff[n], c * 64, 0, 0);
        else if constexpr (n == 9) {           // (patch instruction 39 does not exist)
            if (wave < 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + so + (wave + 36) * 1024), 16, (int)voff[n], c * 64, 0, 0);
        } else if constexpr (n < DB_NPL)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(smem + so + DB_A_BYTES + (wave + 4 * (n - 10)) * 1024), 16, (int)voff[n],
                                                     c * 9 * N * 64, 0, 0);
    };
    auto issue_range = [&](auto self, auto nc, auto endc, int c, unsigned so) {
        constexpr int n = decltype(nc)::value, end = decltype(endc)::value;
        if constexpr (n < end && n < DB_NPL) {
            issue(nc, c, so);
            self(self, std::integral_constant<int, n + 1>(), endc, c, so);
        }
    };

    // ---- one 32-channel chunk out of the stage at `so`: 12 half-steps of (operand reads of the next half-step, DPS DMA
    // instructions of the next chunk, 12 MFMAs).  Half-step t = (group g = t / 2 = (k-step ks, tap column kw), channel half
    // j = t % 2); the six patch rows of a group are read once (fa, by group parity), the three tap-row filter fragments per
    // half-step (fb, by step parity); reads are pinned AHEAD of the MFMAs they feed (sched_barrier).
    unsigned aK[3][2], bK[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aK[kw][ks] = (unsigned)((wave * 4 * 34 + l31 + kw) * 64 + (((ks * 2 + khalf) ^ (((l31 + kw) >> 2) & 3)) << 4));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bK[ks] = (unsigned)(DB_A_BYTES + l31 * 64 + (((ks * 2 + khalf) ^ ((l31 >> 2) & 3)) << 4));
    auto compute = [&](unsigned so, auto pfc, int cn, unsigned sn) {
        constexpr bool PF = decltype(pfc)::value;
        unsigned a0[3][2], b0k[2];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a0[kw][ks] = aK[kw][ks] + so;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b0k[ks] = bK[ks] + so;
        bf16x8 fa[2][6], fb[2][3];
        auto read_a = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int ks = g / 3, kw = g % 3;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr)
                fa[g & 1][rr] = *reinterpret_cast<const bf16x8*>(smem + a0[kw][ks] + rr * DB_PROW);
        };
        auto read_b = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int g = t / 2, j = t % 2, ks = g / 3, kw = g % 3;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                fb[t & 1][kh] = *reinterpret_cast<const bf16x8*>(smem + b0k[ks] + ((kh * 3 + kw) * 64 + j * 32) * 64);
        };
        read_a(std::integral_constant<int, 0>());
        read_b(std::integral_constant<int, 0>());
        auto steps = [&](auto self, auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < 12) {
                constexpr int g = t / 2, j = t % 2;
                if constexpr (t + 1 < 12) {
                    if constexpr (j == 1) read_a(std::integral_constant<int, g + 1>());
                    read_b(std::integral_constant<int, t + 1>());
                }
                __builtin_amdgcn_sched_barrier(0);     // the next half-step's operand reads stay AHEAD of this one's MFMAs
                if constexpr (!(DBG & 4)) {
                    auto taprow = [&](auto self2, auto khc) {
                        constexpr int kh = decltype(khc)::value;
                        if constexpr (kh < 3) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][i + kh], fb[t & 1][kh], acc[i][j], 0, 0, 0);
                            if constexpr (PF) {        // this half-step's DMA instructions, spread over its three tap rows
                                constexpr int lo = t * DPS + (kh * DPS) / 3, hi = t * DPS + ((kh + 1) * DPS) / 3;
                                issue_range(issue_range, std::integral_constant<int, lo>(), std::integral_constant<int, hi>(), cn, sn);
                            }
                            self2(self2, std::integral_constant<int, kh + 1>());
                        }
                    };
                    taprow(taprow, std::integral_constant<int, 0>());
                } else {
                    if constexpr (PF) issue_range(issue_range, std::integral_constant<int, t * DPS>(), std::integral_constant<int, (t + 1) * DPS>(), cn, sn);
                    acc[0][j][0] += (float)fa[g & 1][0][0] * (float)fb[t & 1][0][0] + (float)fa[g & 1][5][0] * (float)fb[t & 1][2][0];       // (keeps the operand reads alive)
                }
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, t + 1>());
            }
        };
        steps(steps, std::integral_constant<int, 0>());
    };

    // ---- epilogue of one item: bias / activation, bf16 packing, per-channel sums, the tile transposed 16 pixels at a time
    // through this wave's private scratch (LDS operations of one wave complete in order: no synchronisation), 16-byte stores
    const int odd = lane & 1;
    auto epilogue = [&](int ox0, int oy0, int ob0, int on0) {
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        if constexpr (BIASACT) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = bias ? bias[on0 + j * 32 + l31] : 0.f;
                const float sv = oscale ? oscale[on0 + j * 32 + l31] : 1.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(acc[i][j][r], sv, bv);
            }
            // the activation code is uniform per launch: ONE scalar branch here, not two per element (with act_fwd() inside the loops
            // every element carried the compare-and-branch pairs of the ReLU / softplus tests: a third of this epilogue's time)
            if (act == PHX_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
            } else if (act != PHX_ACT_ID) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = act_fwd(acc[i][j][r], act);
            }
        }
        unsigned char* scr = smem + DB_OFF_SCR + wave * DB_SCR_WAVE;
        unsigned char* lwp = scr + (4 * khalf + odd) * 144 + (l31 & ~1) * 2;
        const int rpix = lane >> 3, rq = lane & 7;     // read-back: piece (pixel rpix + 8 t, 16-byte slot rq)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned short* yrow = y + (((size_t)ob0 * H + oy0 + wave * 4 + i) * W + ox0) * N + on0 + rq * 8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int rp = 0; rp < 4; ++rp)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r0 = half * 8 + 2 * rp;
                        const unsigned w2 = f2bf_pk(acc[i][j][r0], acc[i][j][r0 + 1]);
                        const float ra_ = __uint_as_float(w2 << 16), rb_ = __uint_as_float(w2 & 0xffff0000u);
                        s1[j] += ra_ + rb_;
                        s2[j] += ra_ * ra_ + rb_ * rb_;
                        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w2, 0xB1, 0xf, 0xf, true);
                        const unsigned word = odd ? ((nb >> 16) | (w2 & 0xffff0000u)) : ((w2 & 0xffffu) | (nb << 16));
                        // pixel within the 16-pixel half: (r0 & 3) + 8 * ((r0 >> 2) & 1) + 4 * khalf (+ 1 on odd lanes)
                        *reinterpret_cast<unsigned*>(lwp + (((2 * rp) & 3) + 8 * (((2 * rp) >> 2) & 1)) * 144 + j * 64) = word;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint4 v = *reinterpret_cast<const uint4*>(scr + (rpix + 8 * t) * 144 + rq * 16);
                    if (!(DBG & 8) || ox0 < 0) *reinterpret_cast<uint4*>(yrow + (size_t)(half * 16 + rpix + 8 * t) * N) = v;
                }
            }
        }
        if (stats_partial) {
            // one row of partial sums per (tile, wave): [tile * 4 + wave][2][N]
            const int tile = ((ob0 * gm.tiles_y + (oy0 >> 4)) * gm.tiles_x + (ox0 >> 5));
            float* sp = stats_partial + ((size_t)(tile * 4 + wave) * 2) * N + on0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a = s1[j] + __shfl_xor(s1[j], 32, 64);
                const float bq = s2[j] + __shfl_xor(s2[j], 32, 64);
                if (khalf == 0) {
                    sp[j * 32 + l31] = a;
                    sp[N + j * 32 + l31] = bq;
                }
            }
        }
        zero_acc();
    };

    // ---- the chunk loop -----------------------------------------------------------------------------------------------------
    if (T <= 0) return;
    DB_TRACE(0);
    set_item(0);
    issue_range(issue_range, std::integral_constant<int, 0>(), std::integral_constant<int, DB_NPL>(), 0, 0u);
    int c = 0, k = 0;
    int ox0 = 0, oy0 = 0, ob0 = 0, on0 = 0;
    bool pend = false;                                 // a finished item whose accumulators still wait for their epilogue
    for (int s = 0; s < T; ++s) {
        // chunk s has landed (this wave's share: vmcnt; everybody's: barrier) and every wave is done reading the other stage
        DB_TRACE(4 * s + 3);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        DB_TRACE(4 * s + 4);
        const unsigned so = (s & 1) ? (unsigned)DB_STAGE : 0u, sn = (unsigned)DB_STAGE - so;
        const bool last = c == nch - 1;
        if (pend) {                                    // (its stores complete under this chunk's MFMAs)
            epilogue(ox0, oy0, ob0, on0);
            pend = false;
        }
        int cn = c + 1;
        if (last) {
            ox0 = tx0; oy0 = ty0; ob0 = b0; on0 = n0;
            cn = 0;
            if (s + 1 < T) set_item(k + 1);
        }
        DB_TRACE(4 * s + 5);
        if (s + 1 < T) compute(so, std::true_type(), cn, sn);
        else compute(so, std::false_type(), 0, 0u);
        if (last) { pend = true; c = 0; ++k; }
        else ++c;
    }
    if (pend) epilogue(ox0, oy0, ob0, on0);
}

}  // namespace

// PHX_FWD_DB=1: k_conv3x3_fwd_db takes the large-map shapes with N % 64 == 0 instead of k_conv3x3_fwd_dma128 (read per call).
// Off by default -- measured (round 3, tools/bench_fwd_db.py, tools/trace_db.py): bit-identical results, 0.32-0.39 ms against
// 0.33-0.34 ms on 128 -> 128 @ 128 x 128 stand-alone and 4 % slower in the training step.  Block 0's cycle stamps say why: a LONE
// wave per SIMD issues its 144 MFMAs of a chunk in 6.65 K cycles with nothing else going on (32 x 144 = 4.6 K would be the pipe's
// rate), the DMA instructions between them cost little in steady state (+0.1 K) but 1.5-2.4 K in the two chunks behind an item
// boundary, and the epilogue of an item (pack, statistics, transposition, stores: 7.7 K cycles) has no partner wave to hide under.
int phx_db_set_trace(void* dev_buf) {
    PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phx_db_trace), &dev_buf, sizeof(void*)));
    return PHX_OK;
}
bool phx_db_enabled() {
    const char* e = getenv("PHX_FWD_DB");
    return e ? atoi(e) != 0 : false;
}
int phx_db_partial_rows(int B, int H, int W) { return B * (H / 16) * (W / 32) * 4; }

int phx_db_launch(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial, int B, int H,
                  int W, int K, int N, const float* oscale, int dbg, void* stream) {
    DBGeom gm;
    gm.tiles_x = W / 32; gm.tiles_y = H / 16;
    gm.ntiles = B * gm.tiles_x * gm.tiles_y;
    gm.ncob = N / 64;
    gm.nitems = gm.ntiles * gm.ncob;
    const char* ge = getenv("PHX_DB_GRID");       // persistent grid size (default: one block per CU)
    const int ncu = ge && atoi(ge) > 0 ? atoi(ge) : 256;
    const int grid = gm.nitems < ncu ? gm.nitems : ncu;
    const bool ba = bias != nullptr || act != PHX_ACT_ID || oscale != nullptr;
    const char* de = getenv("PHX_DB_DPS");        // dev: DMA instructions per half-step (2, 4; default 4)
    const int dps = de ? atoi(de) : 2;
#define DB_LAUNCH(Av, Dv, Pv)                                                                                                     \
    do {                                                                                                                          \
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_fwd_db<Av, Dv, Pv>, hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS_BYTES)); \
        hipLaunchKernelGGL((k_conv3x3_fwd_db<Av, Dv, Pv>), dim3(grid), dim3(256), DB_LDS_BYTES, (hipStream_t)stream,                \
                           (const unsigned short*)x, (const unsigned short*)wpk, (unsigned short*)y, bias, act, stats_partial, B, \
                           H, W, K, N, gm, oscale);                                                                               \
    } while (0)
    if (ba) DB_LAUNCH(true, 0, 2);
    else if (dbg == 0 && dps == 2) DB_LAUNCH(false, 0, 2);
    else if (dbg == 0 && dps == 3) DB_LAUNCH(false, 0, 3);
    else switch (dbg) {
        case 1: DB_LAUNCH(false, 1, 4); break; case 2: DB_LAUNCH(false, 2, 4); break; case 3: DB_LAUNCH(false, 3, 4); break;
        case 4: DB_LAUNCH(false, 4, 4); break; case 16: DB_LAUNCH(false, 16, 4); break; case 20: DB_LAUNCH(false, 20, 4); break; case 7: DB_LAUNCH(false, 7, 4); break; case 8: DB_LAUNCH(false, 8, 4); break;
        default: DB_LAUNCH(false, 0, 4);
    }
#undef DB_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
