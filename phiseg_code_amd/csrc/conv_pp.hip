// bf16 MFMA 3x3 convolution for gfx950, forward / data-gradient, large maps: "ping-pong" persistent kernel.
// Replaces tf.nn.conv2d 3x3 SAME (tfwrapper/layers.py:123) and the data gradient TF derives for it on maps with
// W % 32 == 0, H % 16 == 0 (the 128x128 / 64x64 levels of PHiSeg at batch 64).
//
// Why this shape (measured on MI355X, tools/bench_fwd_ablate.py): in the 256-pixel kernels of conv_mfma.hip and in every
// "two independent blocks per CU" variant the global->LDS staging time and the MFMA time of a block ADD -- the two blocks
// of a CU run the same load -> compute cycle in lock-step (loads-only 0.130 ms + MFMAs-only 0.182 ms = 0.311 ms for
// 128->128 @ 128x128, batch 64).  Here ONE 8-wave block per CU holds two 4-wave groups that alternate by construction:
//
//     phase 2s+1 : group A multiplies step s            | group B stages its input patch of step s   (LDS-DMA)
//     phase 2s+2 : group B multiplies step s            | group A stages patch + filter slab of step s+1, and, after the last
//                                                         chunk of a tile, writes its output tile
//     one s_barrier per phase; a step = one 32-channel chunk of one 16 x 32-pixel tile per group
//
// so every SIMD always has one wave in its MFMA segment and its partner in the memory segment (matrix beside memory).
// The two groups work on two neighbouring pixel tiles of the same 64-channel block and SHARE the filter slab (staged once per
// step by group A, double-buffered): 115 KiB staged per 2 x 18.9 MFLOP = 328 FLOP per staged byte (164 in the 256-pixel
// kernels).  A wave owns four tile rows (4 x 32 pixels) x 64 channels; an MFMA's 32 pixels are one tile row, so the A
// fragment of (row r, tap row kh) is patch row r + kh: six patch-row reads serve the twelve (row, kh) pairs of a (kw, k-step)
// group and every filter fragment feeds four MFMAs -- 12 ds_read_b128 per 24 MFMAs.  Blocks are persistent: each walks a
// strided list of (tile pair, channel block) items, so prologue and epilogue of one item run under the other group's MFMAs.
//
// LDS (160 KiB): patch A 39 KiB | slab 0 36 KiB | slab 1 36 KiB | patch B 39 KiB | 4 x 2.5 KiB epilogue scratch (one per wave
// of the group that is in its memory segment).  Rows are 64 bytes = four 16-byte slots, slot ^= bits 2-3 of the patch
// column / channel (applied on the DMA source side: the LDS destination of buffer_load ... lds is lane-linear), which makes
// every ds_read_b128 lane group cover all 16 slots of a 256-byte bank row exactly once.
#include <stdlib.h>

#include <type_traits>

#include "phx_common.h"

namespace {

constexpr int PP_PATCH_BYTES = 39 * 1024;      // 612 patch pixels x 64 B, rounded up to 1 KiB DMA instructions
constexpr int PP_SLAB_BYTES = 36 * 1024;       // 9 taps x 64 channels x 64 B
constexpr int PP_OFF_PA = 0;
constexpr int PP_OFF_SL = PP_PATCH_BYTES;
constexpr int PP_OFF_PB = PP_PATCH_BYTES + 2 * PP_SLAB_BYTES;
constexpr int PP_OFF_SCR = 2 * PP_PATCH_BYTES + 2 * PP_SLAB_BYTES;
constexpr int PP_SCR_WAVE = 2560;              // 16 pixels x 144 B
constexpr int PP_LDS_BYTES = PP_OFF_SCR + 4 * PP_SCR_WAVE;     // 163840 = 160 KiB
constexpr int PP_PROW = 34 * 64;               // bytes per patch row

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct PPGeom {
    int tiles_x, tiles_y, npairs, ncob, nitems;
};

template <bool BIASACT, int DBG>
__global__ __launch_bounds__(512, 1) void k_conv3x3_pp(const unsigned short* __restrict__ x,
                                                      const unsigned short* __restrict__ wpk,
                                                      unsigned short* __restrict__ y, const float* __restrict__ bias,
                                                      int act, float* __restrict__ stats_partial, int B, int H, int W, int K,
                                                      int N, PPGeom gm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = wave >> 2, gw = wave & 3;          // group A / B; row group (tile rows 4 gw .. 4 gw + 3)
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

    // item k of this block -> (tile of this group, channel block).  Linear item id -> (pair, cob): the N / 64 channel blocks of
    // a tile pair get ids 8 apart (same XCD, same round of the persistent grid), so the patches they share come from HBM once.
    int tx0 = 0, ty0 = 0, b0 = 0, n0 = 0;              // current item (staging side)
    auto setup_item = [&](int k) {
        const int id = (int)blockIdx.x + k * (int)gridDim.x;
        const int full = (gm.npairs >> 3) * 8 * gm.ncob;
        int pair, cob;
        if (id < full) {
            const int g8 = id / (8 * gm.ncob), r = id - g8 * 8 * gm.ncob;
            pair = g8 * 8 + (r & 7);
            cob = r >> 3;
        } else {
            const int rem = gm.npairs & 7, r = id - full;
            pair = (gm.npairs & ~7) + r % rem;
            cob = r / rem;
        }
        int t = pair * 2 + grp;
        tx0 = (t % gm.tiles_x) << 5; t /= gm.tiles_x;
        ty0 = (t % gm.tiles_y) << 4; t /= gm.tiles_y;
        b0 = t;
        n0 = cob * 64;
    };

    // ---- staging: this wave's share of the 39 patch instructions (and, group A, of the 36 slab instructions) of chunk c ----
    auto issue_patch = [&](int c, int lds_off) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                   // keeps the offsets out of long-lived registers (they spilled): recomputed per call
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            const int j = gw + 4 * n;
            if (j < 39) {
                const int e = j * 64 + ln, pp = e >> 2, slot = e & 3;
                const int py = pp / 34, px = pp - py * 34;
                const int piece = slot ^ ((px >> 2) & 3);
                const int gx = tx0 + px - 1, gy = ty0 + py - 1;
                unsigned off = 0xffffffffu;            // outside the image / patch: the buffer range check returns zeros
                if (pp < 612 && gx >= 0 && gx < W && gy >= 0 && gy < H && !(DBG & 1)) off = (unsigned)(((((DBG & 16) ? 0 : b0 * H + gy) * W + ((DBG & 16) ? px : gx)) * K) * 2 + piece * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + lds_off + j * 1024), 16, (int)off, c * 64, 0, 0);
                __builtin_amdgcn_sched_barrier(0);     // (offsets are computed one at a time: clustered they spilled)
            }
        }
    };
    auto issue_slab = [&](int c, int lds_off) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            const int j = gw + 4 * n;
            const int e = j * 64 + ln, rb = e >> 2, slot = e & 3;
            const int tap = rb >> 6, nn = rb & 63;
            const int piece = slot ^ ((nn >> 2) & 3);
            const unsigned off = (DBG & 2) ? 0xffffffffu : (unsigned)(((tap * N + n0 + nn) * 32 + piece * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(smem + lds_off + j * 1024), 16, (int)off, c * 9 * N * 64, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- MFMA segment: one 32-channel chunk, 6 groups of (12 ds_read_b128, 24 MFMAs) --------------------------------------
    unsigned aK[3][2], bK[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aK[kw][ks] = (unsigned)((grp ? PP_OFF_PB : PP_OFF_PA) + (gw * 4 * 34 + l31 + kw) * 64 +
                                    (((ks * 2 + khalf) ^ (((l31 + kw) >> 2) & 3)) << 4));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bK[ks] = (unsigned)(PP_OFF_SL + l31 * 64 + (((ks * 2 + khalf) ^ ((l31 >> 2) & 3)) << 4));
    auto compute = [&](int slab_sel) {
        const unsigned sb = (unsigned)(slab_sel * PP_SLAB_BYTES);
        // 12 half-steps per chunk: step t = (group g = t / 2 = (k-step ks, tap column kw), channel half j = t % 2).  The six patch
        // rows of a group are read once (fa, double-buffered by group parity), the three tap-row filter fragments per half-step
        // (fb, double-buffered by step parity): 18 fragments live instead of 24, one half-step (12 MFMAs) of read-ahead.
        bf16x8 fa[2][6], fb[2][3];
        auto read_a = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int ks = g / 3, kw = g % 3;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr)
                fa[g & 1][rr] = *reinterpret_cast<const bf16x8*>(smem + aK[kw][ks] + rr * PP_PROW);
        };
        auto read_b = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int g = t / 2, j = t % 2, ks = g / 3, kw = g % 3;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                fb[t & 1][kh] = *reinterpret_cast<const bf16x8*>(smem + sb + bK[ks] + ((kh * 3 + kw) * 64 + j * 32) * 64);
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
                if constexpr (!(DBG & 4))
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][i + kh], fb[t & 1][kh], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, t + 1>());
            }
        };
        steps(steps, std::integral_constant<int, 0>());
    };

    // ---- epilogue of one item (memory segment of the owning group): bias / activation, bf16 packing, per-channel sums, the
    // tile transposed 16 pixels at a time through this wave's private scratch (no block-level synchronisation), 16-byte stores
    const int odd = lane & 1;
    auto epilogue = [&](int ox0, int oy0, int ob0, int on0) {
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        if constexpr (BIASACT) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = bias ? bias[on0 + j * 32 + l31] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = act_fwd(acc[i][j][r] + bv, act);
            }
        }
        unsigned char* scr = smem + PP_OFF_SCR + gw * PP_SCR_WAVE;
        unsigned char* lwp = scr + (4 * khalf + odd) * 144 + (l31 & ~1) * 2;
        const int rpix = lane >> 3, rq = lane & 7;     // read-back: piece (pixel rpix + 8 t, 16-byte slot rq)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned short* yrow = y + (((size_t)ob0 * H + oy0 + gw * 4 + i) * W + ox0) * N + on0 + rq * 8;
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
            // one row of partial sums per (tile, wave): [tile * 4 + gw][2][N]
            const int tile = ((ob0 * gm.tiles_y + (oy0 >> 4)) * gm.tiles_x + (ox0 >> 5));
            float* sp = stats_partial + ((size_t)(tile * 4 + gw) * 2) * N + on0;
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

    // ---- the phase loop ---------------------------------------------------------------------------------------------------
    if (grp == 0) {
        setup_item(0);
        issue_patch(0, PP_OFF_PA);
        issue_slab(0, PP_OFF_SL);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    int c = 0, k = 0;
    for (int s = 0; s < T; ++s) {
        // phase 2s+1: A multiplies step s | B stages its patch of step s (and first stores the tile it finished last phase)
        if (grp == 0) {
            compute(s & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if (c == 0) {
                const int ox0 = tx0, oy0 = ty0, ob0 = b0, on0 = n0;
                setup_item(k);
                issue_patch(0, PP_OFF_PB);
                if (k > 0) epilogue(ox0, oy0, ob0, on0);
            } else {
                issue_patch(c, PP_OFF_PB);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        // phase 2s+2: B multiplies step s | A stores its tile after the last chunk, stages patch + slab of step s+1
        if (grp == 1) {
            compute(s & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            const bool last = c == nch - 1;
            const int ox0 = tx0, oy0 = ty0, ob0 = b0, on0 = n0;
            if (s + 1 < T) {
                if (last) setup_item(k + 1);
                const int cn = last ? 0 : c + 1;
                issue_patch(cn, PP_OFF_PA);
                issue_slab(cn, PP_OFF_SL + ((s + 1) & 1) * PP_SLAB_BYTES);
            }
            if (last) epilogue(ox0, oy0, ob0, on0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        if (++c == nch) { c = 0; ++k; }
    }
    if (grp == 1 && T > 0) epilogue(tx0, ty0, b0, n0);
}

int pp_mode() {
    const char* e = getenv("PHX_FWD_PP");          // 0 (default): never -- experimental; 1: policy; 2: whenever eligible (tests)
    return e ? atoi(e) : 0;
}

}  // namespace

// shape gate of the ping-pong kernel (shared with conv_mfma.hip's dispatcher)
bool phx_pp_eligible(int B, int H, int W, int K, int N) {
    const int m = pp_mode();
    if (!m || H % 16 != 0 || W % 32 != 0 || N % 64 != 0 || K % 32 != 0) return false;
    const long ntl = (long)B * (H / 16) * (W / 32);
    if (ntl % 2 != 0) return false;
    if ((double)B * H * W * (K > N ? K : N) >= 2147483648.0) return false;
    if (m == 2) return true;
    return (ntl / 2) * (N / 64) >= 256;            // at least one item per CU
}
int phx_pp_partial_rows(int B, int H, int W) { return B * (H / 16) * (W / 32) * 4; }

int phx_pp_launch(const void* x, const void* wpk, void* y, const float* bias, int act, float* stats_partial, int B, int H,
                  int W, int K, int N, void* stream) {
    PPGeom gm;
    gm.tiles_x = W / 32; gm.tiles_y = H / 16;
    gm.npairs = B * gm.tiles_x * gm.tiles_y / 2;
    gm.ncob = N / 64;
    gm.nitems = gm.npairs * gm.ncob;
    const char* ge = getenv("PHX_PP_GRID");       // persistent grid size (default: one block per CU)
    const int ncu = ge && atoi(ge) > 0 ? atoi(ge) : 256;
    const int grid = gm.nitems < ncu ? gm.nitems : ncu;
    const bool ba = bias != nullptr || act != PHX_ACT_ID;
    const char* dbe = getenv("PHX_DBG_ABLATE");       // dev: bit 1 no patch loads, 2 no slab loads, 4 no MFMAs, 8 no output stores
    const int dbg = dbe ? atoi(dbe) : 0;
#define PP_LAUNCH(Av, Dv)                                                                                                         \
    do {                                                                                                                          \
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_pp<Av, Dv>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES)); \
        hipLaunchKernelGGL((k_conv3x3_pp<Av, Dv>), dim3(grid), dim3(512), PP_LDS_BYTES, (hipStream_t)stream,                        \
                           (const unsigned short*)x, (const unsigned short*)wpk, (unsigned short*)y, bias, act, stats_partial, B, \
                           H, W, K, N, gm);                                                                                       \
    } while (0)
    if (ba) PP_LAUNCH(true, 0);
    else switch (dbg) {
        case 1: PP_LAUNCH(false, 1); break; case 2: PP_LAUNCH(false, 2); break; case 3: PP_LAUNCH(false, 3); break;
        case 4: PP_LAUNCH(false, 4); break; case 7: PP_LAUNCH(false, 7); break; case 8: PP_LAUNCH(false, 8); break;
        case 11: PP_LAUNCH(false, 11); break; case 15: PP_LAUNCH(false, 15); break; case 16: PP_LAUNCH(false, 16); break;
        case 24: PP_LAUNCH(false, 24); break; default: PP_LAUNCH(false, 0);
    }
#undef PP_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
