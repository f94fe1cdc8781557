// bf16 MFMA 3x3 convolution for gfx950, forward / data-gradient on large maps: the ANTI-PHASE PAIR kernel k_conv3x3_pp.
// Replaces tf.nn.conv2d 3x3 SAME (tfwrapper/layers.py:123) + bias / activation / batch-norm statistics epilogue and the data
// gradient TF derives for it, on maps with H % 16 == 0, W % 32 == 0 and enough 16 x 32-pixel tiles to fill the chip.
//
// Why this shape (round-3 ablation, profiles/r03_pmc_conv_power_ablation.txt): the one-stage kernel it replaces ran "DMA a chunk ->
// wait -> 144 MFMAs -> barrier" in two independent blocks per CU; the staging phase (327 kcyc) was longer than the matrix floor
// (295 kcyc) and only a quarter of it hid under the other block's MFMAs.  Here ONE 512-thread work-group per CU holds two halves of
// four waves (waves w and w + 4 share a SIMD).  Each half owns a 16 x 32-pixel tile x BN output channels; the halves run the SAME
// chunk sequence one phase apart, separated by s_barrier:
//
//     phase 2i     half A: 144 MFMAs of item i            half B: DMA patchB(i) + first half of slab(i + 1)  [+ epilogue of its tile]
//     phase 2i + 1 half A: DMA patchA(i + 1) + second     half B: 144 MFMAs of item i
//                  half of slab(i + 1)  [+ epilogue]
//
// (item = (tile pair, output-channel block, 32-channel chunk)).  A SIMD's matrix pipe is therefore fed by one wave at a time while
// its partner wave issues the LDS-DMA instructions (which cost 100-185 cycles apiece inside an MFMA stream, guide "LDS-DMA piece
// issue cost", and nothing from the partner), packs and stores a finished tile.  The two tiles of a pair take the same channel
// block, so the 9 x BN x 64 B filter slab of a chunk is staged ONCE for 1 024 pixels: 114 KiB staged per 37.7 MFLOP (331 FLOP/B,
// 246 before).  LDS: patchA | patchB (39 KiB each: 18 x 34 pixels x 64 B) + two slab buffers (36 KiB each) = 150 KiB.
// Blocks are persistent over a strided list of (pair, channel block) work items, so prologue and epilogue of consecutive tiles
// overlap the partner's MFMAs as well; the finished tile is transposed through the half's own quarter of the slab buffer it is
// about to refill (wave-private scratch: no barrier), 16-byte stores.
// The MFMA stream: per half-step 12 MFMAs with the six ds_read_b128 of the NEXT half-step's fragments pinned one behind every
// second MFMA (a burst of nine reads between two MFMA groups stalled the lone wave's pipe: 6.65 K cycles per chunk against 4.6 K).
#include <stdlib.h>

#include <type_traits>

#include "phx_common.h"

struct Dual {                                // (as in conv_mfma.hip: concat-free second input / output tensor)
    const unsigned short* x2;
    unsigned short* y2;
    int K1, N1;
};

#ifndef PP_FENCE          // dev: scheduling fence every PP_FENCE row pairs of the epilogue's pack loop (0: none)
#define PP_FENCE 2
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int PP_AI = 39;                    // 1 KiB DMA instructions per patch: 612 pixels (18 x 34) x 64 B
constexpr int PP_APW = 10;                   // ... per wave of a half (wave lw: pieces [10 lw, 10 lw + 10))
constexpr int PP_PATCH = PP_AI * 1024;       // 39 936 B
constexpr int PP_PROW = 34 * 64;             // bytes per patch row

struct PPGeom {
    int tiles_x, tiles_y, ntiles, npairs, ncob, nitems;
    unsigned mgx, mgy;                       // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y)
};

__device__ unsigned long long* g_pp_trace = nullptr;      // dev: cycle stamps of block 0 (phx_debug_set_trace)
// stamps go to the spare 6 KiB of LDS (a global store per stamp would sit in front of every s_waitcnt vmcnt(0): ~550 cycles each) and
// are copied out when the block ends
#define PP_TRACE(slot)                                                                           \
    do {                                                                                         \
        if (tracing && (slot) < 96)                                                              \
            *reinterpret_cast<unsigned long long*>(smem + OFF_TRACE + ((slot) * 8 + wave) * 8) = __builtin_readcyclecounter(); \
    } while (0)

// DBG (dev builds with -DPP_DEV_ABLATE, tools/bench_pp.py): 1 no patch DMA, 2 no slab DMA, 4 no MFMAs
// TR (launches without bias / activation / statistics: every data gradient): the MFMA operands are swapped -- D = filter x patch --
// so that a lane holds ONE pixel and four consecutive output channels per accumulator quad: the epilogue packs and writes 8 bytes per
// lane and quad (32 ds_write_b64 per tile and wave) instead of trading rows with the neighbour lane (64 DPP + 64 v_perm + 64
// ds_write_b32).  The epilogue is LDS-issue-bound (~47 cycles per LDS instruction beside the partner's operand reads and the DMA).
template <int BN, bool BIASACT, bool DUAL, int DBG, bool TR = false>
__global__ __launch_bounds__(512, 1) void k_conv3x3_pp(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wpk,
                                                       unsigned short* __restrict__ y, const float* __restrict__ bias, int act,
                                                       float* __restrict__ stats_partial, int B, int H, int W, int K, int N,
                                                       PPGeom gm, const float* __restrict__ oscale, int stats_nrep, Dual du) {
    constexpr int NJ = BN / 32;                       // 32-channel MFMA columns per wave
    constexpr int SLAB = 9 * BN * 64;                 // bytes per slab buffer
    constexpr int SP = SLAB / 2048;                   // 1 KiB DMA instructions per slab HALF (18 / 9)
    constexpr int CW = BN / 16;                       // ... contiguous per wave (its epilogue scratch: CW KiB), SP - 4 CW extras
    constexpr int NSW = CW + 1;
    constexpr int OFF_S0 = 0, OFF_PA = SLAB, OFF_S1 = SLAB + PP_PATCH, OFF_PB = 2 * SLAB + PP_PATCH, OFF_RED = 2 * SLAB + 2 * PP_PATCH;
    constexpr int OFF_TRACE = 160 * 1024 - 6144;      // dev: 96 stamps x 8 waves
    constexpr int OROW = BN * 2;                      // bytes per pixel of the transposed output tile (dense rows, piece XOR swizzle)
    constexpr int OSWZ = BN == 64 ? 64 : 0;
    constexpr int PPP = BN / 8;                       // 16-byte pieces per output pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;      // (wave: an SGPR -- LDS-DMA
    const int half = wave >> 2, lw = wave & 3;        // destinations and scalar offsets derived from it need no waterfall loop) half 0 = A (leads), 1 = B
    const int l31 = lane & 31, khalf = lane >> 5;
    const int nch = K / 32;
    const bool tracing = g_pp_trace != nullptr && blockIdx.x == 0 && lane == 0;
    const int K1 = (DUAL && du.x2 != nullptr) ? du.K1 : K;

    // ---- DMA plan of this wave's patch pieces -------------------------------------------------------------------------------------
    // piece n = instruction j = 10 lw + n fills LDS slot e = 64 j + lane with 16 bytes of patch pixel pp = e / 4; the 16-byte slot
    // index is XORed with bits 2-3 of the patch column on the SOURCE side (the LDS destination of buffer_load ... lds is lane-linear),
    // so that every ds_read_b128 lane group of the operand reads covers a whole bank row.  The per-piece values (patch row / column,
    // source piece, which image edges can cut it off) are RECOMPUTED in every load phase (~12 VALU per piece, in a phase that has
    // thousands of idle issue slots): ten more live registers would not fit beside the MFMA stream, and a spilled plan is worse than
    // none -- its scratch reloads share vmcnt with the DMA instructions and serialise them (measured: 2-7 K cycles to issue ten).
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * K1 * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? du.x2 : x), 0, DUAL ? (int)((unsigned)B * H * W * (K - K1) * 2u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, (int)(9u * N * K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)((unsigned)B * H * W * N * 2u), 0x00020000);

    // operand addresses (as in the one-stage kernel): A fragment of (tile row r, tap row kh), tap column kw, k-step ks = patch row
    // 4 lw + r + kh, pixel l31 + kw, 16-byte slot (2 ks + khalf) ^ swizzle(column); B fragment of (tap, column block j) = slab row
    // tap * BN + 32 j + l31.  Kept as ONE base + the six 2-bit slots packed in a register (2 VALU per tap column and k-step inside the
    // MFMA stream, where they are free) instead of eight address registers: the stream runs at the 256-register limit.
    const unsigned patch_off = half ? OFF_PB : OFF_PA;
    unsigned char* const patch = smem + patch_off;
    const unsigned a0 = patch_off + (unsigned)((lw * 4 * 34 + l31) * 64);
    unsigned aslots = 0;
#pragma unroll
    for (int g = 0; g < 6; ++g) aslots |= (unsigned)((((g / 3) * 2 + khalf) ^ (((l31 + g % 3) >> 2) & 3))) << (2 * g);
    const unsigned bK0 = (unsigned)(l31 * 64 + ((khalf ^ ((l31 >> 2) & 3)) << 4));       // (k-step 1: ^ 32)

    // ---- work list -----------------------------------------------------------------------------------------------------------------
    // work item w -> (tile pair, channel block): the ncob channel blocks of a pair get ids 8 apart (same XCD, same round), so the
    // patch they share comes from HBM once
    const int nwork = ((int)blockIdx.x < gm.nitems) ? (gm.nitems - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = nwork * nch;                    // (tile, chunk) items of this half
    // current tile of this half (all wave-uniform)
    int wk = 0;                                        // index into the work list
    int t_id = 0, tx0 = 0, ty0 = 0, b0 = 0, n0 = 0;
    bool t_valid = false;
    unsigned pbase = 0, pout = 16u;
    auto set_work = [&](int k) __attribute__((always_inline)) {
        const int w = (int)blockIdx.x + k * (int)gridDim.x;
        const int full = (gm.npairs >> 3) * 8 * gm.ncob;
        int pr, cob;
        if (w < full) {
            const int grp = w / (8 * gm.ncob), r = w - grp * 8 * gm.ncob;
            pr = phx_band8(grp * 8 + (r & 7), gm.npairs);          // (XCD bands: phx_common.h -- neighbouring pairs share halos in one L2)
            cob = r >> 3;
        } else {
            const int rem = gm.npairs & 7, r = w - full;
            pr = (gm.npairs & ~7) + r % rem;
            cob = r / rem;
        }
        const int t = 2 * pr + half;
        t_id = t;
        t_valid = t < gm.ntiles;
        const int q1 = gm.tiles_x == 1 ? t : (int)__umulhi((unsigned)t, gm.mgx);
        tx0 = (t - q1 * gm.tiles_x) << 5;
        const int q2 = gm.tiles_y == 1 ? q1 : (int)__umulhi((unsigned)q1, gm.mgy);
        ty0 = (q1 - q2 * gm.tiles_y) << 4;
        b0 = q2;
        n0 = cob * BN;
        pbase = (unsigned)((b0 * H + ty0 - 1) * W + tx0 - 1);                              // patch pixel (0, 0); may wrap below zero
        pout = 16u | (tx0 == 0 ? 1u : 0u) | (tx0 + 32 >= W ? 2u : 0u) | (ty0 == 0 ? 4u : 0u) | (ty0 + 16 >= H ? 8u : 0u);
        if (!t_valid || (DBG & 1)) pout |= 32u;        // no tile (odd tile count) / dev: every piece reads as zero
    };
    // DMA of this half's patch for chunk c of the current tile
    auto load_patch = [&](int c) __attribute__((always_inline)) {
        const bool second = DUAL && c * 32 >= K1;
        const unsigned stride = (unsigned)((second ? K - K1 : K1) * 2);
        const unsigned pb = pbase * stride;
        const int so = second ? c * 64 - K1 * 2 : c * 64;
        int ln = lane;
        asm volatile("" : "+v"(ln));                   // (opaque: nothing lane-derived below is hoisted out of the phase loop)
        const int pp0 = lw * (PP_APW * 16) + (ln >> 2), slot = ln & 3;
        // a piece lies outside the image only in patch column 0 / 33 or patch row 0 / 17 of a tile on the matching image edge (pout);
        // the rows behind the patch (pp >= 612 <=> py >= 18) belong to no pixel
        const int xlo = (int)(pout & 1u), xw = 33 - (int)((pout >> 1) & 1u) - xlo;
        const int ylo = (int)((pout >> 2) & 1u), yw = 17 - (int)((pout >> 3) & 1u) - ylo;
        const bool none = (pout & 32u) != 0u;
#pragma unroll
        for (int n = 0; n < PP_APW; ++n) {
            const int j = lw * PP_APW + n;
            if (j < PP_AI) {
                const int pp = pp0 + n * 16;           // (per piece, independent of the others: ten chains for the scheduler to interleave)
                const int py = (int)(((unsigned)pp * 1928u) >> 16), px = pp - py * 34;      // pp / 34, pp % 34 (pp < 1024)
                const bool bad = none || (unsigned)(px - xlo) > (unsigned)xw || (unsigned)(py - ylo) > (unsigned)yw;
                const unsigned vo = bad ? 0xffffffffu : pb + __umul24((unsigned)(py * W + px), stride) + (unsigned)((slot ^ ((px >> 2) & 3)) << 4);
                if (DUAL && second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (lds_ptr_t)(patch + j * 1024), 16, (int)vo, so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(patch + j * 1024), 16, (int)vo, so, 0, 0);
            }
        }
    };
    // DMA of this half's share of the slab of (channel block at n0s, chunk c) into slab buffer sb: half h owns pieces [h SP, h SP + SP),
    // wave lw the CW contiguous ones from h SP + CW lw (its epilogue scratch) and, for lw < SP - 4 CW, one of the extras behind them
    auto load_slab = [&](int c, int n0s, int sb) __attribute__((always_inline)) {
        if (DBG & 2) return;
        // piece j of a slab = (tap j / (BN / 16), rows 16 (j % (BN / 16)) .. + 15): only the SCALAR offset depends on j
        unsigned char* const dst = smem + (sb ? OFF_S1 : OFF_S0);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned voff_s = (unsigned)((ln >> 2) * 64 + (((ln & 3) ^ ((ln >> 4) & 3)) << 4));
        const int cbase = (c * 9 * N + n0s) * 64;
#pragma unroll
        for (int n = 0; n < NSW; ++n) {
            const int j = half * SP + (n < CW ? CW * lw + n : 4 * CW + lw);
            if (n < CW || lw < SP - 4 * CW) {
                const int so = cbase + (j / (BN / 16)) * N * 64 + (j % (BN / 16)) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + j * 1024), 16, (int)voff_s, so, 0, 0);
            }
        }
    };

    f32x16 acc[4][NJ];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- the matrix phase: 144 (72) MFMAs of one chunk, operand fragments double-buffered, reads pinned between the MFMAs ---------
    // FIRST (first chunk of a tile): the first MFMA into every accumulator takes srcC = 0 -- no 128 v_mov per tile in the epilogue phase
    auto compute = [&](int sb, auto firstc) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(firstc)::value;
        const unsigned bb0 = bK0 + (unsigned)(sb ? OFF_S1 : OFF_S0);
        bf16x8 fa[2][6], fb[2][3];
        unsigned ag = 0;                                // address of the tap column / k-step being prefetched
        auto rd_a = [&](auto gc, auto rc) {
            constexpr int g = decltype(gc)::value, rr = decltype(rc)::value;
            constexpr int kw = g % 3;
            if constexpr (rr == 0 || (NJ == 2 && rr == 3)) ag = a0 + (((aslots >> (2 * g)) & 3u) << 4);
            fa[g & 1][rr] = *reinterpret_cast<const bf16x8*>(smem + ag + kw * 64 + rr * PP_PROW);
        };
        auto rd_b = [&](auto tc, auto kc) {
            constexpr int t = decltype(tc)::value, kh = decltype(kc)::value;
            constexpr int g = t / NJ, j = t % NJ, ks = (g / 3) & 1, kw = g % 3;
            fb[t & 1][kh] = *reinterpret_cast<const bf16x8*>(smem + (bb0 ^ (ks * 32)) + ((kh * 3 + kw) * BN + j * 32) * 64);
        };
#define IC(v) std::integral_constant<int, (v)>()
        // head: what the first MFMAs need first
        rd_b(IC(0), IC(0)); rd_a(IC(0), IC(0)); rd_a(IC(0), IC(1)); rd_a(IC(0), IC(2)); rd_a(IC(0), IC(3));
        rd_b(IC(0), IC(1)); rd_a(IC(0), IC(4)); rd_b(IC(0), IC(2)); rd_a(IC(0), IC(5));
        __builtin_amdgcn_sched_barrier(0);
        auto step = [&](auto self, auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < 6 * NJ) {
                constexpr int g = t / NJ, j = t % NJ;
                constexpr bool more = t + 1 < 6 * NJ;
                auto slot = [&](auto sc) {
                    constexpr int s = decltype(sc)::value, kh = s / 4, i = s % 4;
                    if constexpr (!(DBG & 4)) {
                        const bf16x8 opa = TR ? fb[t & 1][kh] : fa[g & 1][i + kh], opb = TR ? fa[g & 1][i + kh] : fb[t & 1][kh];
                        if constexpr (FIRST && g == 0 && kh == 0)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opa, opb, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opa, opb, acc[i][j], 0, 0, 0);
                    }
                    // one read of the next half-step's fragments behind every second MFMA
                    if constexpr (more && (s & 1) == 0) {
                        constexpr int q = s / 2;                                     // 0 .. 5
                        if constexpr (NJ == 2) {
                            if constexpr (j == 0) {          // next: same group, column block 1 -- its filter fragments, then rows 0-2 of group g + 1
                                if constexpr (q < 3) rd_b(IC(t + 1), IC(q));
                                else if constexpr (g + 1 < 6) rd_a(IC(g + 1), IC(q - 3));
                            } else {                         // next: group g + 1, column block 0
                                if constexpr (q == 0) rd_b(IC(t + 1), IC(0));
                                else if constexpr (q == 1) rd_a(IC(g + 1), IC(3));
                                else if constexpr (q == 2) rd_b(IC(t + 1), IC(1));
                                else if constexpr (q == 3) rd_a(IC(g + 1), IC(4));
                                else if constexpr (q == 4) rd_b(IC(t + 1), IC(2));
                                else rd_a(IC(g + 1), IC(5));
                            }
                        }
                    }
                    if constexpr (more && NJ == 1) {         // nine reads per twelve MFMAs: slots 0-8
                        if constexpr (s == 0) rd_b(IC(t + 1), IC(0));
                        else if constexpr (s >= 1 && s <= 4) rd_a(IC(g + 1), IC(s - 1));
                        else if constexpr (s == 5) rd_b(IC(t + 1), IC(1));
                        else if constexpr (s == 6) rd_a(IC(g + 1), IC(4));
                        else if constexpr (s == 7) rd_b(IC(t + 1), IC(2));
                        else if constexpr (s == 8) rd_a(IC(g + 1), IC(5));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                slot(IC(0)); slot(IC(1)); slot(IC(2)); slot(IC(3)); slot(IC(4)); slot(IC(5));
                slot(IC(6)); slot(IC(7)); slot(IC(8)); slot(IC(9)); slot(IC(10)); slot(IC(11));
                self(self, IC(t + 1));
            }
        };
        step(step, IC(0));
#undef IC
    };

    // ---- epilogue of the tile (et_*) whose accumulators this half holds: bias / activation, pack, statistics, transposition through
    // the wave's scratch (its CW KiB of slab buffer `sb`, which this half refills afterwards), 16-byte stores ------------------------
    int et_id = 0, et_tx0 = 0, et_ty0 = 0, et_b0 = 0, et_n0 = 0;
    bool et_valid = false;
    bool stats_pending = false;
    int sp_id = 0, sp_n0 = 0;
    float* const red = reinterpret_cast<float*>(smem + OFF_RED) + wave * 2 * BN;       // [which][BN] of this wave
    auto epilogue = [&](int sb) __attribute__((always_inline)) {
        if (!et_valid) return;
        // (everything lane-derived is recomputed here from an opaque copy of the lane id: hoisted out of the phase loop these values
        // would stay live across the MFMA stream, which has no register to spare)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l31e = ln & 31, khalfe = ln >> 5, odd = ln & 1;
        unsigned char* const scr = smem + (sb ? OFF_S1 : OFF_S0) + (half * SP + CW * lw) * 1024;
        const unsigned psel = odd ? 0x03020706u : 0x05040100u;
        const bool do_stats = stats_partial != nullptr;
        float s1[NJ][2], s2[NJ][2];                    // (two accumulators per statistic: half the dependent chain)
        [[maybe_unused]] float bv[NJ], sv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            s1[j][0] = s1[j][1] = s2[j][0] = s2[j][1] = 0.f;
            if constexpr (BIASACT) {
                bv[j] = bias ? bias[et_n0 + j * 32 + l31e] : 0.f;
                sv[j] = oscale ? oscale[et_n0 + j * 32 + l31e] : 1.f;
            }
        }
        // lane part of a store: piece p = lane + 64 it of the pass's 32 PPP pieces -> pixel row p / PPP, 16-byte piece q = p % PPP
        const int prow = ln / PPP, q = ln % PPP;
        // (TR: the 16-byte piece index of a row is XORed with the row's low bits -- the 8-byte writes of 16 consecutive pixels spread over
        // eight pieces, two lanes per bank pair)
        const unsigned char* const lr = scr + prow * OROW + (TR ? ((q ^ (prow & (PPP - 1))) * 16) : ((q * 16) ^ ((prow & 1) ? OSWZ : 0)));
        unsigned char* const lwp = scr + (4 * khalfe + odd) * OROW + (l31e & ~1) * 2;
        // destination of piece (pass i, it): pixel (et_ty0 + 4 lw + i, et_tx0 + prow + it * 64 / PPP), channels et_n0 + 8 q ..
        [[maybe_unused]] unsigned short* ybase = y;
        [[maybe_unused]] int yld = N, ych = et_n0 + q * 8;
        if constexpr (DUAL)
            if (du.y2) {
                if (ych < du.N1) yld = du.N1;
                else { ybase = du.y2; yld = N - du.N1; ych -= du.N1; }
            }
        const unsigned vo_y = (unsigned)((prow * N + q * 8) * 2);
        // one pass = one tile row of the wave (32 pixels x BN channels): pack pairs of accumulator rows (bias / scale / activation on
        // the way: AM = 0 none, 1 ReLU -- uniform per launch, one branch per tile), statistics of the values as stored,
        // exchange with the neighbour lane (DPP) so that a lane holds two adjacent channels of ONE pixel, 32-bit LDS writes,
        // read back 16 bytes per lane, store
        // Software pipeline over the four passes: W0 R0 | W1 S0 R1 | W2 S1 R2 | W3 S2 R3 | S3 -- the read-back of pass i returns under
        // the packing of pass i + 1 (a wave's LDS operations execute in order, so the writes of pass i + 1 cannot overtake the reads
        // of pass i although they share the scratch); left to the compiler the four passes ran write -> read -> WAIT -> store.
        u32x4_t vst[2][PPP / 2];
        auto pack_write = [&](auto ic, auto amc) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, AM = decltype(amc)::value & 3;
            constexpr bool ST = (decltype(amc)::value & 4) != 0;
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                // (scheduling fence per row pair -- per two without the bias arithmetic: left alone, the machine scheduler hoists the cheap head of all sixteen chains
                // of a pass -- bias multiply-adds, conversions -- in front of the first LDS write and spills the results)
                if (PP_FENCE && (BIASACT || rp % PP_FENCE == 0)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int r0 = 2 * rp;
                    float v0 = acc[i][j][r0], v1 = acc[i][j][r0 + 1];
                    if constexpr (BIASACT) {
                        v0 = fmaf(v0, sv[j], bv[j]); v1 = fmaf(v1, sv[j], bv[j]);
                        if constexpr (AM == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    }
                    const unsigned w2 = f2bf_pk(v0, v1);
                    if constexpr (ST) {            // of the values as stored: v_dot2c_f32_bf16 (products of bf16 pairs are exact in fp32)
                        const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w2);
                        s1[j][rp & 1] = __builtin_amdgcn_fdot2_f32_bf16(wv, __builtin_bit_cast(bf16x2_t, 0x3f803f80u), s1[j][rp & 1], false);
                        s2[j][rp & 1] = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, s2[j][rp & 1], false);
                    }
                    const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w2, 0xB1, 0xf, 0xf, true);
                    *reinterpret_cast<unsigned*>(lwp + ((r0 & 3) + 8 * (r0 >> 2)) * OROW + ((j * 64) ^ (odd ? OSWZ : 0))) = __builtin_amdgcn_perm(nb, w2, psel);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto pack_write_tr = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            unsigned char* const wr = scr + l31e * OROW + 8 * khalfe;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
                    const u32x2_t w = {f2bf_pk(acc[i][j][4 * qd], acc[i][j][4 * qd + 1]), f2bf_pk(acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3])};
                    *reinterpret_cast<u32x2_t*>(wr + (((j * 4 + qd) ^ (l31e & (PPP - 1))) * 16)) = w;
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_back = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int it = 0; it < PPP / 2; ++it) vst[i & 1][it] = *reinterpret_cast<const u32x4_t*>(lr + it * (64 / PPP) * OROW);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto store = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const int oy = et_ty0 + 4 * lw + i;
            const int pix0 = (et_b0 * H + oy) * W + et_tx0;
#pragma unroll
            for (int it = 0; it < PPP / 2; ++it) {
                if constexpr (DUAL) {
                    *reinterpret_cast<u32x4_t*>(ybase + ((size_t)pix0 + prow + it * (64 / PPP)) * yld + ych) = vst[i & 1][it];
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(vst[i & 1][it], rsy, (int)vo_y, (pix0 + it * (64 / PPP)) * N * 2 + et_n0 * 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#define IC(v) std::integral_constant<int, (v)>()
        const int am = ((BIASACT && act == PHX_ACT_RELU) ? 1 : 0) + (do_stats ? 4 : 0);
#define PP_PASSES(m)                                                                                     \
    do {                                                                                                 \
        PP_TRACE(88);                                                                                    \
        pack_write(IC(0), IC(m)); PP_TRACE(89); read_back(IC(0)); PP_TRACE(90);                          \
        pack_write(IC(1), IC(m)); PP_TRACE(91); store(IC(0)); PP_TRACE(92); read_back(IC(1));            \
        pack_write(IC(2), IC(m)); PP_TRACE(93); store(IC(1)); read_back(IC(2));                          \
        pack_write(IC(3), IC(m)); store(IC(2)); read_back(IC(3)); PP_TRACE(94);                          \
        store(IC(3));                                                                                    \
        PP_TRACE(95);                                                                                    \
    } while (0)
        if constexpr (TR) {
            PP_TRACE(88);
            pack_write_tr(IC(0)); PP_TRACE(89); read_back(IC(0)); PP_TRACE(90);
            pack_write_tr(IC(1)); store(IC(0)); read_back(IC(1));
            pack_write_tr(IC(2)); store(IC(1)); read_back(IC(2));
            pack_write_tr(IC(3)); store(IC(2)); read_back(IC(3));
            store(IC(3));
            PP_TRACE(95);
        } else if (am == 0) PP_PASSES(0);
        else if (am == 4) PP_PASSES(4);
        else if constexpr (BIASACT) {                 // (identity / ReLU only: the launcher refuses other activations)
            if (am == 1) PP_PASSES(1);
            else PP_PASSES(5);
        }
#undef PP_PASSES
#undef IC
        if (do_stats) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float a0 = s1[j][0] + s1[j][1], b0q = s2[j][0] + s2[j][1];
                const float a = a0 + __shfl_xor(a0, 32, 64);
                const float bq = b0q + __shfl_xor(b0q, 32, 64);
                if (khalfe == 0) {
                    red[j * 32 + l31e] = a;
                    red[BN + j * 32 + l31e] = bq;
                }
            }
            stats_pending = true;
            sp_id = et_id;
            sp_n0 = et_n0;
        }
    };
    // cross-wave sum of a finished tile's statistics (behind a barrier that followed its epilogue): threads 0 .. 2 BN - 1 of the half
    auto flush_stats = [&]() __attribute__((always_inline)) {
        if (!stats_pending) return;
        stats_pending = false;
        const int th = lw * 64 + lane;
        if (th < 2 * BN) {
            const int which = th / BN, n = th % BN;
            const float* const rh = reinterpret_cast<const float*>(smem + OFF_RED) + half * 4 * 2 * BN;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += rh[w * 2 * BN + which * BN + n];
            // stats_nrep > 0: stats_partial is the replicated accumulator sums[rep][N][2] (tile t -> replica t % stats_nrep, atomics)
            if (stats_nrep > 0) atomicAdd(&stats_partial[((size_t)(sp_id % stats_nrep) * N + sp_n0 + n) * 2 + which], v);
            else stats_partial[((size_t)sp_id * 2 + which) * N + sp_n0 + n] = v;
        }
    };
#define PP_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (total == 0) return;
    if constexpr (DBG & 4) zero_acc();
    set_work(0);
    // chunk / channel block of the item after the current one (B stages its share of a slab one item ahead of its patch)
    int c = 0;                                        // chunk of the current item within its tile
    auto next_slab = [&](int* cn, int* n0n) __attribute__((always_inline)) {
        *cn = c + 1;
        *n0n = n0;
        if (*cn == nch) {                             // first chunk of the next work item: its channel block
            *cn = 0;
            const int w = (int)blockIdx.x + (wk + 1) * (int)gridDim.x;
            const int full = (gm.npairs >> 3) * 8 * gm.ncob;
            *n0n = (w < full ? (w % (8 * gm.ncob)) >> 3 : (w - full) / (gm.npairs & 7)) * BN;
        }
    };
    // phase -1: A loads patchA(0) and its share of slab(0); B its share of slab(0)
    if (half == 0) load_patch(0);
    load_slab(0, n0, 0);
    PP_BARRIER();
    if (half) {                                       // phase 0 of B (A computes item 0): patchB(0), its share of slab(1)
        load_patch(0);
        if (1 < total) {
            int cn, n0n;
            next_slab(&cn, &n0n);
            load_slab(cn, n0n, 1);
        }
        PP_BARRIER();
    }
    // both halves run the same body one phase apart: [matrix phase of item `it`] barrier [load phase in front of item it + 1] barrier
    for (int it = 0; it < total; ++it) {
        flush_stats();
        if (it < 11) PP_TRACE(it * 8 + 0);
        if (c == 0) compute(it & 1, std::true_type());
        else compute(it & 1, std::false_type());
        if (it < 11) PP_TRACE(it * 8 + 1);
        const bool last = c + 1 == nch;
        if (last) {
            et_id = t_id; et_tx0 = tx0; et_ty0 = ty0; et_b0 = b0; et_n0 = n0; et_valid = t_valid;
            c = 0;
            ++wk;
            if (it + 1 < total) set_work(wk);
        } else ++c;
        PP_BARRIER();
        if (it < 11) PP_TRACE(it * 8 + 2);
        if (it + 1 < total) load_patch(c);
        if (it < 11) PP_TRACE(it * 8 + 3);
        if (last) {
            epilogue((it + 1 + half) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the scratch has been read before the DMA overwrites it
        }
        if (it < 11) PP_TRACE(it * 8 + 4);
        if (half == 0) {
            if (it + 1 < total) load_slab(c, n0, (it + 1) & 1);
        } else if (it + 2 < total) {
            int cn, n0n;
            next_slab(&cn, &n0n);
            load_slab(cn, n0n, it & 1);
        }
        if (it < 11) PP_TRACE(it * 8 + 5);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (it < 11) PP_TRACE(it * 8 + 6);
        PP_BARRIER();
        if (it < 11) PP_TRACE(it * 8 + 7);
    }
    if (half == 0) PP_BARRIER();                      // (B is one phase behind)
    flush_stats();
    if (g_pp_trace != nullptr && blockIdx.x == 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < 768; i += 512) g_pp_trace[i] = *reinterpret_cast<const unsigned long long*>(smem + OFF_TRACE + i * 8);
    }
#undef PP_BARRIER
}

}  // namespace

#ifdef PHX_DEBUG_BUILD                    // libphx_dbg.so (tests): persistent grid size, 0 = one block per CU -- a constant in libphx.so
static int g_pp_grid = 0;
int phx_pp_set_grid(int blocks) {
    g_pp_grid = blocks;
    return PHX_OK;
}
#else
static constexpr int g_pp_grid = 0;
#endif

int phx_pp_set_trace(void* dev_buf) {
    PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), &dev_buf, sizeof(void*)));
    return PHX_OK;
}

// shapes the pair kernel takes: 16 x 32-pixel tiles, 32-channel chunks, 32 / 64-channel blocks
bool phx_pp_shape_ok(int B, int H, int W, int K, int N) {
    return H % 16 == 0 && W % 32 == 0 && K % 32 == 0 && N % 32 == 0 && (double)B * H * W * (K > N ? K : N) < 2147483648.0 &&
           (long)B * (H / 16) * (W / 32) < 65536;
}

int phx_pp_launch(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial, int B, int H, int W,
                  int K, int N, const float* oscale, int stats_nrep, Dual du, int dbg, void* stream) {
    PHX_REQUIRE(phx_pp_shape_ok(B, H, W, K, N), PHX_E_SHAPE, "conv3x3_pp: H % 16, W % 32, K % 32, N % 32 == 0 required");
    PHX_REQUIRE(act == PHX_ACT_ID || act == PHX_ACT_RELU, PHX_E_INVAL, "conv3x3_pp: identity / ReLU epilogue only");
    const int bn = N % 64 == 0 ? 64 : 32;
    PPGeom gm;
    gm.tiles_x = W / 32; gm.tiles_y = H / 16;
    gm.ntiles = B * gm.tiles_x * gm.tiles_y;
    gm.npairs = (gm.ntiles + 1) / 2;
    gm.ncob = N / bn;
    gm.nitems = gm.npairs * gm.ncob;
    gm.mgx = (unsigned)((0x100000000ull + gm.tiles_x - 1) / gm.tiles_x);
    gm.mgy = (unsigned)((0x100000000ull + gm.tiles_y - 1) / gm.tiles_y);
    static int ncu_dev = 0;
    if (!ncu_dev) {
        int dev = 0;
        hipDeviceProp_t pr;
        ncu_dev = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    }
    const int ncu = g_pp_grid > 0 ? g_pp_grid : ncu_dev;
    // equal rounds: the grid that gives every block the same number of work items (no block waits for a straggler's extra item)
    int grid = gm.nitems < ncu ? gm.nitems : ncu;
    if (gm.nitems > ncu) {
        const int rounds = (gm.nitems + ncu - 1) / ncu;
        grid = (gm.nitems + rounds - 1) / rounds;
    }
    const bool ba = bias != nullptr || act != PHX_ACT_ID || oscale != nullptr;
    const bool dual = du.x2 != nullptr || du.y2 != nullptr;
    const size_t lds64 = 160 * 1024, lds32 = 160 * 1024;      // (one block per CU by design; the tail holds the dev stamps)
#define PP_LAUNCH(BNv, Av, Dv, Gv) do { if (!(Av) && (Gv) == 0 && stats_partial == nullptr) PP_LAUNCH1(BNv, false, Dv, 0, true); else PP_LAUNCH1(BNv, Av, Dv, Gv, false); } while (0)
#define PP_LAUNCH1(BNv, Av, Dv, Gv, Tv)                                                                                            \
    do {                                                                                                                          \
        auto kf = k_conv3x3_pp<BNv, Av, Dv, Gv, Tv>;                                                                                  \
        static bool at = false;                                                                                                   \
        if (!at) { PHX_CHECK_HIP(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); at = true; } \
        hipLaunchKernelGGL(kf, dim3(grid), dim3(512), BNv == 64 ? lds64 : lds32, (hipStream_t)stream, (const unsigned short*)x,    \
                           (const unsigned short*)wpk, (unsigned short*)y, bias, act, stats_partial, B, H, W, K, N, gm, oscale,   \
                           stats_nrep, du);                                                                                       \
    } while (0)
    if (bn == 64) {
        if (dual) { if (ba) PP_LAUNCH(64, true, true, 0); else PP_LAUNCH(64, false, true, 0); }
        else if (ba) PP_LAUNCH(64, true, false, 0);
#ifdef PP_DEV_ABLATE          // dev builds (tools/ppc.sh -DPP_DEV_ABLATE): staging / matrix phases switched off
        else if (dbg == 1) PP_LAUNCH(64, false, false, 1);
        else if (dbg == 2) PP_LAUNCH(64, false, false, 2);
        else if (dbg == 3) PP_LAUNCH(64, false, false, 3);
        else if (dbg == 4) PP_LAUNCH(64, false, false, 4);
#endif
        else PP_LAUNCH(64, false, false, 0);
    } else {
        if (dual) { if (ba) PP_LAUNCH(32, true, true, 0); else PP_LAUNCH(32, false, true, 0); }
        else if (ba) PP_LAUNCH(32, true, false, 0);
        else PP_LAUNCH(32, false, false, 0);
    }
    (void)dbg;
#undef PP_LAUNCH
#undef PP_LAUNCH1
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
