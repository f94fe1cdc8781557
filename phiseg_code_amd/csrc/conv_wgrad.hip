// bf16 MFMA 3x3 convolution for gfx950: FILTER gradient (M = Cin, N = Cout, K = pixels; LDS transpose reads), the gradient TF
// derives for tf.nn.conv2d 3x3 SAME (tfwrapper/layers.py:123) with respect to the filter.  Split from conv_mfma.hip (forward /
// data gradient) so that the two halves compile in parallel; tile geometry and debug stamps are shared through conv_common.h.
#include "conv_common.h"

// ---- filter gradient --------------------------------------------------------------------------------------
// Block tile: TCI input channels x TCO output channels (32 or 64 each) x all 9 taps.  The 4 waves split the tile
// into 32x32 sub-tiles (WI x WJ) and, when the tile has fewer than four sub-tiles, the pixel (k) steps WK ways;
// each wave keeps 9 accumulators (one per tap).  Reduction index = pixel, so both MFMA operands need "k" along
// the pixel axis of channel-contiguous NHWC data: ds_read_b64_tr_b16 (LDS transpose read) delivers, for a
// 16-lane group, column (lane&15) of a 4 (pixels) x 16 (channels) block -- 4 k-values per lane per read.
template <int RB>   // RB = bytes per pixel row in LDS (64 or 128); 128-byte rows XOR-swizzle their halves
__device__ __forceinline__ int wswz(int pix, int byte_in_row) {
    if (RB == 128) return pix * 128 + (byte_in_row ^ (((pix >> 1) & 1) << 6));
    return pix * RB + byte_in_row;
}

// BIGP selects the bound on the per-thread staging pieces: false -> 16x16 / 8x8x4 tiles, true -> 4x4x16 / 2x2x64 tiles
// (body shared by the one-layer kernel and the multi-layer one below: bx / by / bz / gdx / gdy stand in for blockIdx, gridDim)
template <int TCI, int TCO, bool BIGP, bool FAST16>
__device__ __forceinline__ void conv3x3_wgrad_body(const unsigned short* __restrict__ x0, const unsigned short* __restrict__ dy,
                                                   float* __restrict__ dw, float* __restrict__ ws, int B, int H, int W, int Cin,
                                                   int Cout, MTile g, int ntiles, int tiles_per_block, const int bx,
                                                   const int by, const int bz, const int gdx, const int gdy,
                                                   const unsigned short* __restrict__ x2 = nullptr, const int K1 = 0) {
    // concat-free input (struct Dual): this block's TCI input channels lie in x0 (channels [0, K1), pixel stride K1) or in x2
    // (channels [K1, Cin), pixel stride Cin - K1); K1 % TCI == 0.  xC = pixel stride, xc0 = first channel inside the source.
    const bool src2 = x2 != nullptr && by * TCI >= K1;
    const unsigned short* __restrict__ x = src2 ? x2 : x0;
    const int xC = x2 == nullptr ? Cin : (src2 ? Cin - K1 : K1);
    const int xc0 = by * TCI - (src2 ? K1 : 0);
    constexpr int WI = TCI / 32, WJ = TCO / 32, WK = 4 / (WI * WJ);
    constexpr int RBX = TCI * 2, RBD = TCO * 2;
    constexpr int QX = TCI / 8, QD = TCO / 8;                        // 16-byte pieces per pixel
    constexpr int NXI = ((BIGP ? 1024 : 400) * QX + 255) / 256;      // pieces of the input patch per thread
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int pw = tw + 2, ph = th + 2;
    const int npatch = g.tb * ph * pw;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                    // [npatch][RBX]
    unsigned char* sD = smem + npatch * RBX;     // [256][RBD]
    const int ci0 = by * TCI, co0 = bz * TCO;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = wave % WI, wj = (wave / WI) % WJ, wk = wave / (WI * WJ);
    const int c16 = lane & 15, cb16 = (lane >> 4) & 1, khalf = lane >> 5;
    // this lane supplies the 8-byte chunk of pixel-slot (c16>>2) and channels 16*cb16 + 4*(c16&3) .. +3
    const int chan_byte_x = (wi * 32 + cb16 * 16 + (c16 & 3) * 4) * 2;
    const int chan_byte_d = (wj * 32 + cb16 * 16 + (c16 & 3) * 4) * 2;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    // FAST16 per-lane LDS constants: pixel column lx_r = 8*khalf + 4*r + (c16>>2) of the lane's two pixel slots
    unsigned dcon[2], xcon[2][3][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lxr = khalf * 8 + r * 4 + (c16 >> 2);
        dcon[r] = (unsigned)wswz<RBD>(lxr, chan_byte_d);                  // + ks*16*RBD: the row term never flips the swizzle
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int par = 0; par < 2; ++par)                            // par = parity of the patch row
                xcon[r][kw][par] = (unsigned)((lxr + kw) * RBX +
                                              (RBX == 128 ? (chan_byte_x ^ (((((lxr + kw) >> 1) & 1) ^ par) << 6)) : chan_byte_x));
    }

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // staging plan (tile independent): patch / tile coordinates of this thread's 16-byte pieces, packed px | py<<8 | pb<<16
    int planx[NXI], pland[QD];
#pragma unroll
    for (int it = 0; it < NXI; ++it) {
        const int i = threadIdx.x + it * 256;
        const int pp = i / QX;
        planx[it] = -1;
        if (pp < npatch) {
            if constexpr (FAST16) planx[it] = (pp % 18) | ((pp / 18) << 8);
            else {
                int px, py, pb;
                patch_coords(g, pp, pw, ph, &px, &py, &pb);
                planx[it] = px | (py << 8) | (pb << 16);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < QD; ++it) {
        const int m = (threadIdx.x + it * 256) / QD;
        pland[it] = (m & (tw - 1)) | (((m >> g.tws) & (th - 1)) << 8) | ((m >> (g.tws + g.ths)) << 16);
    }
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 rx[NXI], rd[QD];
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * xC * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * H * W * Cout * 2u), 0x00020000);
    const int qxb = (threadIdx.x % QX) * 16, qdb = (threadIdx.x % QD) * 16;      // 256 % QX == 0: the same for every piece
    // global -> registers for tile t.  Branch-free: a piece outside the image / batch gets buffer offset 0xffffffff and
    // reads zeros.  (With `if (inside) load` the compiler drained the load queue at every branch join: 40 pieces x ~350
    // cycles = 6.8 us of prologue on the small-map tiles, a third of the launch.)
    auto prefetch = [&](int t) {
        const int tx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
        const int ty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
        const int b0 = t * g.tb;
#pragma unroll
        for (int it = 0; it < NXI; ++it) {
            const int gx = tx0 + (planx[it] & 255) - 1, gy = ty0 + ((planx[it] >> 8) & 255) - 1, gb = b0 + (planx[it] >> 16);
            const bool ok = planx[it] >= 0 && (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && gb < B;
            const unsigned vo = ok ? (unsigned)(((gb * H + gy) * W + gx) * xC * 2 + qxb) : 0xffffffffu;
            rx[it] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)vo, xc0 * 2, 0);
        }
#pragma unroll
        for (int it = 0; it < QD; ++it) {
            const int ox = tx0 + (pland[it] & 255), oy = ty0 + ((pland[it] >> 8) & 255), ob = b0 + (pland[it] >> 16);
            const bool ok = ox < W && oy < H && ob < B;
            const unsigned vo = ok ? (unsigned)(((ob * H + oy) * W + ox) * Cout * 2 + qdb) : 0xffffffffu;
            rd[it] = __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)vo, co0 * 2, 0);
        }
    };
    // FAST16 kernels stage with raw buffer loads instead: a piece outside the image gets offset 0xffffffff, which the buffer
    // range check turns into zeros -- no branches, so single pieces can be issued between the MFMAs of the k-steps and the
    // global-load path (~12 B/clk/CU, the scarce resource of this kernel) works underneath the matrix pipe.
    int ptx0 = 0, pty0 = 0, pb0 = 0;             // tile being prefetched
    auto decode = [&](int t) {
        ptx0 = (t % g.tiles_x) << g.tws; t /= g.tiles_x;
        pty0 = (t % g.tiles_y) << g.ths; t /= g.tiles_y;
        pb0 = t * g.tb;
    };
    auto prefetch_piece = [&](auto idxc) {       // piece idx (input patch first, then dy) of the tile at (ptx0, pty0, pb0)
        constexpr int idx = decltype(idxc)::value;
        if constexpr (PHX_ABLATE & 16) return;
        if constexpr (idx < NXI) {
            const int gx = ptx0 + (planx[idx] & 255) - 1, gy = pty0 + ((planx[idx] >> 8) & 255) - 1, gb = pb0 + (planx[idx] >> 16);
            const bool ok = planx[idx] >= 0 && gx >= 0 && gx < W && gy >= 0 && gy < H && gb < B;
            const unsigned vo = ok ? (unsigned)(((gb * H + gy) * W + gx) * xC * 2 + qxb) : 0xffffffffu;
            rx[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsx, vo, xc0 * 2, 0);
        } else if constexpr (idx < NXI + QD) {
            constexpr int it = idx - NXI;
            const int ox = ptx0 + (pland[it] & 255), oy = pty0 + ((pland[it] >> 8) & 255), ob = pb0 + (pland[it] >> 16);
            const bool ok = ox < W && oy < H && ob < B;
            const unsigned vo = ok ? (unsigned)(((ob * H + oy) * W + ox) * Cout * 2 + qdb) : 0xffffffffu;
            rd[it] = __builtin_amdgcn_raw_buffer_load_b128(rsd, vo, co0 * 2, 0);
        }
    };
    auto prefetch_range = [&](auto self, auto lo, auto hi) {     // pieces [lo, hi)
        constexpr int l = decltype(lo)::value, h = decltype(hi)::value;
        if constexpr (l < h && l < NXI + QD) {
            prefetch_piece(lo);
            self(self, std::integral_constant<int, l + 1>(), hi);
        }
    };
    const int t_begin = bx * tiles_per_block;
    const int t_end = min(ntiles, t_begin + tiles_per_block);
    PHX_BLOCKLOG_BEGIN();
    PHX_TRACE(0);
    if (t_begin < t_end) {
        if constexpr (FAST16) {
            decode(t_begin);
            prefetch_range(prefetch_range, std::integral_constant<int, 0>(), std::integral_constant<int, NXI + QD>());
        } else {
            prefetch(t_begin);
        }
    }
    PHX_TRACE(1);

    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();                         // previous tile fully consumed
        if (t == t_begin) PHX_TRACE(2);
#pragma unroll
        for (int it = 0; it < NXI; ++it) {
            const int i = threadIdx.x + it * 256;
            if (planx[it] >= 0) *reinterpret_cast<u32x4*>(sX + wswz<RBX>(i / QX, (i % QX) * 16)) = rx[it];
        }
#pragma unroll
        for (int it = 0; it < QD; ++it) {
            const int i = threadIdx.x + it * 256;
            *reinterpret_cast<u32x4*>(sD + wswz<RBD>(i / QD, (i % QD) * 16)) = rd[it];
        }
        __syncthreads();
        if (t == t_begin) PHX_TRACE(3);
        const bool more = t + 1 < t_end;
        if constexpr (FAST16) { if (more) decode(t + 1); }
        else if (more) prefetch(t + 1);          // next tile's global loads fly under this tile's MFMAs
        if (t == t_begin) PHX_TRACE(4);
        if (FAST16) {
            // 16x16 tiles (tb = 1): every LDS address is (row * const) + per-lane constant, and the swizzle bit is
            // parity(row) ^ per-lane bit, so with the k-steps taken in (even, odd) pairs all row terms are immediates.
            // Software pipeline over this wave's k-steps (ks = wk + WK * step): the 20 transpose reads of step s+1 are
            // issued before the 9 MFMAs of step s (fragment registers double-buffered by step parity, order pinned with
            // sched_barrier) -- with one wave per SIMD nothing else hides the LDS latency.
            constexpr int NSTEP = 16 / WK;
            s16x4 fd[2][2], fx[2][9][2];
            auto read_step = [&](unsigned xb, unsigned db, auto parc, auto bufc) {
                constexpr int P = decltype(parc)::value;          // parity of this k-step's tile row
                constexpr int Bf = decltype(bufc)::value;
                fd[Bf][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + db + dcon[0]));
                fd[Bf][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + db + dcon[1]));
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int par = (P + kh) & 1;
                        fx[Bf][kh * 3 + kw][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(smem + xb + kh * 18 * RBX + xcon[0][kw][par]));
                        fx[Bf][kh * 3 + kw][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(smem + xb + kh * 18 * RBX + xcon[1][kw][par]));
                    }
            };
            auto mfma_step = [&](auto bufc) {
                constexpr int Bf = decltype(bufc)::value;
                const s16x8 dtmp = {fd[Bf][0][0], fd[Bf][0][1], fd[Bf][0][2], fd[Bf][0][3], fd[Bf][1][0], fd[Bf][1][1], fd[Bf][1][2], fd[Bf][1][3]};
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, dtmp);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const s16x8 atmp = {fx[Bf][k][0][0], fx[Bf][k][0][1], fx[Bf][k][0][2], fx[Bf][k][0][3],
                                        fx[Bf][k][1][0], fx[Bf][k][1][1], fx[Bf][k][1][2], fx[Bf][k][1][3]};
                    if constexpr (PHX_ABLATE & 32) acc[k][k] += (float)atmp[0] * (float)dtmp[k & 7];
                    else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, atmp), bfrag, acc[k], 0, 0, 0);
                }
            };
            const unsigned xbase = 0, dbase = (unsigned)(npatch * RBX);
            // WK == 1: ks = step, parity compile-time per step.  WK = 2, 4: parity(ks) = parity(wk) for every step.
            constexpr int IPK = (NXI + QD + NSTEP - 1) / NSTEP;      // next-tile pieces issued per k-step
            auto pipeline = [&](auto wparc, auto pfc) {
                constexpr int WP = decltype(wparc)::value;
                constexpr bool PF = decltype(pfc)::value;
                auto ksof = [&](int si) { return WK == 1 ? si : wk + WK * si; };
                read_step(xbase + ksof(0) * 18 * RBX, dbase + ksof(0) * 16 * RBD, std::integral_constant<int, WP>(), std::integral_constant<int, 0>());
                auto steps = [&](auto self, auto stepc) {
                    constexpr int SI = decltype(stepc)::value;
                    if constexpr (SI < NSTEP) {
                        if constexpr (SI + 1 < NSTEP)
                            read_step(xbase + ksof(SI + 1) * 18 * RBX, dbase + ksof(SI + 1) * 16 * RBD,
                                      std::integral_constant<int, WK == 1 ? ((SI + 1) & 1) : WP>(), std::integral_constant<int, (SI + 1) & 1>());
                        if constexpr (PF)
                            prefetch_range(prefetch_range, std::integral_constant<int, SI * IPK>(), std::integral_constant<int, SI * IPK + IPK>());
                        mfma_step(std::integral_constant<int, SI & 1>());
                        __builtin_amdgcn_sched_barrier(0);
                        self(self, std::integral_constant<int, SI + 1>());
                    }
                };
                steps(steps, std::integral_constant<int, 0>());
            };
            if (more) {
                if (WK == 1 || !(wk & 1)) pipeline(std::integral_constant<int, 0>(), std::true_type());
                else pipeline(std::integral_constant<int, 1>(), std::true_type());
            } else {
                if (WK == 1 || !(wk & 1)) pipeline(std::integral_constant<int, 0>(), std::false_type());
                else pipeline(std::integral_constant<int, 1>(), std::false_type());
            }
        } else {
            for (int ks = wk; ks < 16; ks += WK) {
                // the two pixel slots this lane addresses in this k-step (r = 0, 1): m = 16*ks + 8*khalf + 4*r + (c16>>2)
                int mpix[2], ppix[2];
    #pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int m = ks * 16 + khalf * 8 + r * 4 + (c16 >> 2);
                    const int lx = m & (tw - 1), ly = (m >> g.tws) & (th - 1), lb = m >> (g.tws + g.ths);
                    mpix[r] = m;
                    ppix[r] = (lb * ph + ly) * pw + lx;
                }
                const s16x4 d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sD + wswz<RBD>(mpix[0], chan_byte_d)));
                const s16x4 d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sD + wswz<RBD>(mpix[1], chan_byte_d)));
                const s16x8 dtmp = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, dtmp);
    #pragma unroll
                for (int kh = 0; kh < 3; ++kh)
    #pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int sh = kh * pw + kw;
                        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(sX + wswz<RBX>(ppix[0] + sh, chan_byte_x)));
                        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(sX + wswz<RBX>(ppix[1] + sh, chan_byte_x)));
                        const s16x8 atmp = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        const bf16x8 afrag = __builtin_bit_cast(bf16x8, atmp);
                        acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[kh * 3 + kw], 0, 0, 0);
                    }
            }
        }
    }
    // C layout: col = lane&31 -> co, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> ci
    PHX_TRACE(5);
    if (ws) {
        // A CU issues fp32 atomics at ~1 lane/clock (measured: 36.8 K lane-atomics = 46 us per block), so the partial
        // tile goes to a workspace with plain coalesced stores: ws[(cblock * gdx + bx)][wk][9][TCI][TCO];
        // k_wgrad_reduce sums the slices into dw.
        const size_t cb = (size_t)bz * gdy + by;
        float* wp = ws + ((cb * gdx + bx) * WK + wk) * (size_t)(9 * TCI * TCO);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cil = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wp[(k * TCI + cil) * TCO + wj * 32 + (lane & 31)] = acc[k][r];
            }
    } else {
        const int co = co0 + wj * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                atomicAdd(&dw[((size_t)k * Cin + ci) * Cout + co], acc[k][r]);
            }
    }
    PHX_TRACE(6);
    PHX_BLOCKLOG_END();
}

template <int TCI, int TCO, bool BIGP, bool FAST16>
__global__ __launch_bounds__(256, 1) void k_conv3x3_wgrad(const unsigned short* __restrict__ x,
                                                          const unsigned short* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ ws, int B, int H,
                                                          int W, int Cin, int Cout, MTile g, int ntiles,
                                                          int tiles_per_block, const unsigned short* __restrict__ x2, int K1) {
    conv3x3_wgrad_body<TCI, TCO, BIGP, FAST16>(x, dy, dw, ws, B, H, W, Cin, Cout, g, ntiles, tiles_per_block, blockIdx.x,
                                               blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, x2, K1);
}
// Small-map filter gradients are leaves of the backward graph and latency-bound (a few tiles, 9-36 blocks, ~20 us each, in
// the middle of the posterior / prior / likelihood chains).  The engine defers them: ONE launch per kernel variant runs the
// jobs of all such layers side by side after the lanes have joined.  jobs[j].blk0 = first block of job j (ascending).
struct WgMJob {
    const unsigned short* x; const unsigned short* dy; float* dw; float* ws;
    int B, H, W, Cin, Cout;
    MTile g;
    int ntiles, tpb, gdx, gdy, gdz, blk0;
    const unsigned short* x2; int K1, pad_;      // concat-free input (struct Dual); x2 == NULL: single tensor
};
template <int TCI, int TCO, bool BIGP>
__global__ __launch_bounds__(256, 1) void k_conv3x3_wgrad_multi(const WgMJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                         // last job with blk0 <= blockIdx.x (uniform per block)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgMJob j = jobs[lo];
    const int local = blockIdx.x - j.blk0;
    conv3x3_wgrad_body<TCI, TCO, BIGP, false>(j.x, j.dy, j.dw, j.ws, j.B, j.H, j.W, j.Cin, j.Cout, j.g, j.ntiles, j.tpb,
                                              local % j.gdx, (local / j.gdx) % j.gdy, local / (j.gdx * j.gdy), j.gdx, j.gdy, j.x2, j.K1);
}

// ---- filter gradient, 16x16 tiles, LDS-DMA staging ---------------------------------------------------------------
// Same tiling, LDS image and k-step code as k_conv3x3_wgrad<.., FAST16>, but the input patch and the dy tile go global ->
// LDS directly (buffer_load ... lds): no staging registers and no ds_write pass, so the kernel fits 256 registers and TWO
// blocks share a CU -- one block's loads (the global-load path, ~12 B/clk/CU, is what bounds this kernel) run under the
// other block's MFMAs.  The DMA writes lane-linear (wave-uniform base + lane * 16), so the 64-byte half swizzle of the
// 128-byte rows is applied on the SOURCE side (each lane fetches the piece that belongs in its slot); pieces outside the
// image carry offset 0xffffffff and the buffer range check writes zeros for them.  Partial filters go to the workspace.
template <int TCI, int TCO>
__device__ __forceinline__ void conv3x3_wgrad_dma_body(const unsigned short* __restrict__ x0, const unsigned short* __restrict__ dy,
                                                       float* __restrict__ ws, int B, int H, int W, int Cin, int Cout, MTile g,
                                                       int ntiles, int tiles_per_block, const int bx, const int by, const int bz,
                                                       const int gdx, const int gdy,
                                                       const unsigned short* __restrict__ x2 = nullptr, const int K1 = 0,
                                                       const float* __restrict__ xscale = nullptr, const float* __restrict__ xshift = nullptr) {
    // xscale != NULL (round 5, the 32-channel input blocks only: TCI == 32): x is the PRE-normalisation tensor of the layer in front; when
    // the patch has landed every thread rewrites its pieces of it in place as a = relu(x * xscale[ci] + xshift[ci]) (out-of-image pieces
    // back to zero), as k_conv3x3_c32 does in the forward pass -- the filter gradient of a convolution whose input activation was never
    // written (phx_conv3x3_mfma_bf16_xf).  These launches are HBM-bound (matrix pipe 17 % busy): the pass is free, the bytes are not.
    const bool src2 = x2 != nullptr && by * TCI >= K1;       // concat-free input: see conv3x3_wgrad_body
    const unsigned short* __restrict__ x = src2 ? x2 : x0;
    const int xC = x2 == nullptr ? Cin : (src2 ? Cin - K1 : K1);
    const int xc0 = by * TCI - (src2 ? K1 : 0);
    constexpr int WI = TCI / 32, WJ = TCO / 32, WK = 4 / (WI * WJ);
    constexpr int RBX = TCI * 2, RBD = TCO * 2;
    constexpr int QX = TCI / 8, QD = TCO / 8;                        // 16-byte pieces per pixel
    constexpr int NPATCH = 324;                                      // 18 x 18
    constexpr int XI = (NPATCH * QX + 63) / 64, DI = 256 * QD / 64;  // wave-instructions (1 KiB each) per tile
    constexpr int XN = (XI + 3) / 4, DN = DI / 4;                    // per wave
    constexpr int SD_OFF = XI * 1024;                                // dy tile starts on the next 1 KiB boundary
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ci0 = by * TCI, co0 = bz * TCO;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = wave % WI, wj = (wave / WI) % WJ, wk = wave / (WI * WJ);
    const int c16 = lane & 15, cb16 = (lane >> 4) & 1, khalf = lane >> 5;
    const int chan_byte_x = (wi * 32 + cb16 * 16 + (c16 & 3) * 4) * 2;
    const int chan_byte_d = (wj * 32 + cb16 * 16 + (c16 & 3) * 4) * 2;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    unsigned dcon[2], xcon[2][3][2];             // per-lane LDS constants, see k_conv3x3_wgrad
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lxr = khalf * 8 + r * 4 + (c16 >> 2);
        dcon[r] = (unsigned)(SD_OFF + wswz<RBD>(lxr, chan_byte_d));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int par = 0; par < 2; ++par)
                xcon[r][kw][par] = (unsigned)((lxr + kw) * RBX +
                                              (RBX == 128 ? (chan_byte_x ^ (((((lxr + kw) >> 1) & 1) ^ par) << 6)) : chan_byte_x));
    }
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // DMA plan: lane `lane` of wave-instruction j = wave + 4 n fills LDS slot e = 64 j + lane with the SOURCE piece p of
    // patch pixel (px, py) / tile pixel m.  The 64 x 64 kernel keeps the (tile independent) plan in registers; the others
    // (two MFMA code paths' worth of registers short) recompute it per tile, a dozen integer ops per load.
    constexpr bool RPLAN = WK == 1;
    auto plan_x = [&](int n) -> int {
        const int e = (wave + 4 * n) * 64 + lane;
        const int pp = e / QX, ps = e % QX;
        const int p = RBX == 128 ? (ps ^ (((pp >> 1) & 1) << 2)) : ps;
        return pp < NPATCH ? ((pp % 18) | ((pp / 18) << 8) | (p << 16)) : -1;
    };
    auto plan_d = [&](int n) -> int {
        const int e = (wave + 4 * n) * 64 + lane;
        const int m = e / QD, ps = e % QD;
        const int p = RBD == 128 ? (ps ^ (((m >> 1) & 1) << 2)) : ps;
        return (m & 15) | ((m >> 4) << 8) | (p << 16);
    };
    int planx[RPLAN ? XN : 1], pland[RPLAN ? DN : 1];
    if constexpr (RPLAN) {
#pragma unroll
        for (int n = 0; n < XN; ++n) planx[n] = plan_x(n);
#pragma unroll
        for (int n = 0; n < DN; ++n) pland[n] = plan_d(n);
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * xC * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * H * W * Cout * 2u), 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int t_begin = bx * tiles_per_block;
    const int t_end = min(ntiles, t_begin + tiles_per_block);
    PHX_BLOCKLOG_BEGIN();
    for (int t = t_begin; t < t_end; ++t) {
        int tt = t;
        const int tx0 = (tt % g.tiles_x) << 4; tt /= g.tiles_x;
        const int ty0 = (tt % g.tiles_y) << 4; tt /= g.tiles_y;
        const int b0 = tt;
        __syncthreads();                         // previous tile fully consumed
#pragma unroll
        for (int n = 0; n < XN; ++n) {
            if (wave + 4 * n < XI) {
                int pk;
                if constexpr (RPLAN) pk = planx[n]; else pk = plan_x(n);
                const int gx = tx0 + (pk & 255) - 1, gy = ty0 + ((pk >> 8) & 255) - 1;
                const bool ok = pk >= 0 && gx >= 0 && gx < W && gy >= 0 && gy < H;
                const unsigned vo = ok ? (unsigned)((((b0 * H + gy) * W + gx) * xC) * 2 + (pk >> 16) * 16) : 0xffffffffu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + (wave + 4 * n) * 1024), 16, vo, xc0 * 2, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < DN; ++n) {
            int pk;
            if constexpr (RPLAN) pk = pland[n]; else pk = plan_d(n);
            const int ox = tx0 + (pk & 255), oy = ty0 + ((pk >> 8) & 255);
            const bool ok = ox < W && oy < H;
            const unsigned vo = ok ? (unsigned)((((b0 * H + oy) * W + ox) * Cout) * 2 + (pk >> 16) * 16) : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lds_ptr_t)(smem + SD_OFF + (wave + 4 * n) * 1024), 16, vo, co0 * 2, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (TCI == 32) {
            if (xscale != nullptr) {                 // (uniform per launch)
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                typedef __attribute__((ext_vector_type(4))) float f32x4_t;
                const int sp = threadIdx.x & 3;
                const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(xscale + ci0 + sp * 8), c1 = *reinterpret_cast<const f32x4_t*>(xscale + ci0 + sp * 8 + 4);
                const f32x4_t d0 = *reinterpret_cast<const f32x4_t*>(xshift + ci0 + sp * 8), d1 = *reinterpret_cast<const f32x4_t*>(xshift + ci0 + sp * 8 + 4);
                const float xsc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                const float xsh[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    const int pp = (int)(threadIdx.x >> 2) + 64 * n;
                    if (pp < NPATCH) {
                        const int py = (int)(((unsigned)pp * 3641u) >> 16), px = pp - py * 18;        // pp / 18 for pp < 512
                        const int gx = tx0 + px - 1, gy = ty0 + py - 1;
                        const bool inside = gx >= 0 && gx < W && gy >= 0 && gy < H;
                        u32x4_t* const q = reinterpret_cast<u32x4_t*>(smem + pp * RBX + sp * 16);
                        const u32x4_t r = *q;
                        u32x4_t o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float a0 = fmaxf(fmaf(__uint_as_float(r[k] << 16), xsc[2 * k], xsh[2 * k]), 0.f);
                            const float a1 = fmaxf(fmaf(__uint_as_float(r[k] & 0xffff0000u), xsc[2 * k + 1], xsh[2 * k + 1]), 0.f);
                            o[k] = inside ? f2bf_pk(a0, a1) : 0u;
                        }
                        *q = o;
                    }
                }
                __syncthreads();
            }
        }

        // k-steps of this wave, software pipelined as in k_conv3x3_wgrad
        constexpr int NSTEP = 16 / WK;
        s16x4 fd[2][2], fx[2][9][2];
        auto read_step = [&](unsigned xb, unsigned db, auto parc, auto bufc) {
            constexpr int P = decltype(parc)::value;
            constexpr int Bf = decltype(bufc)::value;
            fd[Bf][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + db + dcon[0]));
            fd[Bf][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + db + dcon[1]));
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int par = (P + kh) & 1;
                    fx[Bf][kh * 3 + kw][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(smem + xb + kh * 18 * RBX + xcon[0][kw][par]));
                    fx[Bf][kh * 3 + kw][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(smem + xb + kh * 18 * RBX + xcon[1][kw][par]));
                }
        };
        auto mfma_step = [&](auto bufc) {
            constexpr int Bf = decltype(bufc)::value;
            const s16x8 dtmp = {fd[Bf][0][0], fd[Bf][0][1], fd[Bf][0][2], fd[Bf][0][3], fd[Bf][1][0], fd[Bf][1][1], fd[Bf][1][2], fd[Bf][1][3]};
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, dtmp);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const s16x8 atmp = {fx[Bf][k][0][0], fx[Bf][k][0][1], fx[Bf][k][0][2], fx[Bf][k][0][3],
                                    fx[Bf][k][1][0], fx[Bf][k][1][1], fx[Bf][k][1][2], fx[Bf][k][1][3]};
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, atmp), bfrag, acc[k], 0, 0, 0);
            }
        };
        // wave group wk takes the NSTEP consecutive k-steps ks = wk * NSTEP + step (NSTEP is even or 16: the parity of ks
        // -- which selects the swizzle immediates -- is the parity of step, a compile-time value)
        const unsigned xw = (unsigned)(wk * NSTEP * 18 * RBX), dw_ = (unsigned)(wk * NSTEP * 16 * RBD);
        read_step(xw, dw_, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
        auto steps = [&](auto self, auto stepc) {
            constexpr int SI = decltype(stepc)::value;
            if constexpr (SI < NSTEP) {
                if constexpr (SI + 1 < NSTEP)
                    read_step(xw + (SI + 1) * 18 * RBX, dw_ + (SI + 1) * 16 * RBD, std::integral_constant<int, (SI + 1) & 1>(),
                              std::integral_constant<int, (SI + 1) & 1>());
                mfma_step(std::integral_constant<int, SI & 1>());
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, SI + 1>());
            }
        };
        steps(steps, std::integral_constant<int, 0>());
    }
    // The WK wave groups of a block hold partial sums of the SAME filter tile (they split the tile's pixels): summed here through
    // LDS (the staging buffers are free now) in log2(WK) rounds -- wave groups with bit `half` set hand their 36 KiB of
    // accumulators to the group below -- so a block writes ONE partial filter instead of WK (the 32 x 32 layers at 128 x 128 wrote
    // 52 MB of partials per launch, a quarter of the kernel's time, and the reduction read them back).
    if constexpr (WK > 1) {
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
#pragma unroll
        for (int half = 1; half < WK; half <<= 1) {
            const int sel = wk & (2 * half - 1);
            f32x4_t* p = red + (size_t)(wi + WI * (wj + WJ * (wk / (2 * half)))) * (9 * 4 * 64) + lane;
            __syncthreads();                     // staging tile / previous round consumed
            if (sel == half)
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        p[(k * 4 + q) * 64] = f32x4_t{acc[k][4 * q], acc[k][4 * q + 1], acc[k][4 * q + 2], acc[k][4 * q + 3]};
            __syncthreads();
            if (sel == 0)
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4_t v = p[(k * 4 + q) * 64];
                        acc[k][4 * q] += v[0]; acc[k][4 * q + 1] += v[1]; acc[k][4 * q + 2] += v[2]; acc[k][4 * q + 3] += v[3];
                    }
        }
    }
    // partial tile -> workspace (see k_conv3x3_wgrad): ws[(cblock * gdx + bx)][9][TCI][TCO]; C layout: col = lane&31 -> co,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> ci
    if (WK == 1 || wk == 0) {
        const size_t cb = (size_t)bz * gdy + by;
        float* wp = ws + (cb * gdx + bx) * (size_t)(9 * TCI * TCO);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cil = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wp[(k * TCI + cil) * TCO + wj * 32 + (lane & 31)] = acc[k][r];
            }
    }
    PHX_BLOCKLOG_END();
}

template <int TCI, int TCO>
__global__ __launch_bounds__(256, 2) void k_conv3x3_wgrad_dma(const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ dy, float* __restrict__ ws,
                                                              int B, int H, int W, int Cin, int Cout, MTile g, int ntiles,
                                                              int tiles_per_block, const unsigned short* __restrict__ x2, int K1,
                                                              const float* __restrict__ xscale, const float* __restrict__ xshift) {
    conv3x3_wgrad_dma_body<TCI, TCO>(x, dy, ws, B, H, W, Cin, Cout, g, ntiles, tiles_per_block, blockIdx.x, blockIdx.y, blockIdx.z,
                                     gridDim.x, gridDim.y, x2, K1, xscale, xshift);
}
// multi-layer form (see k_conv3x3_wgrad_multi): the 16x16-tile layers with few tiles (H = 16 at batch 64)
template <int TCI, int TCO>
__global__ __launch_bounds__(256, 2) void k_conv3x3_wgrad_dma_multi(const WgMJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgMJob j = jobs[lo];
    const int local = blockIdx.x - j.blk0;
    conv3x3_wgrad_dma_body<TCI, TCO>(j.x, j.dy, j.ws, j.B, j.H, j.W, j.Cin, j.Cout, j.g, j.ntiles, j.tpb, local % j.gdx,
                                     (local / j.gdx) % j.gdy, local / (j.gdx * j.gdy), j.gdx, j.gdy, j.x2, j.K1);
}

// ---- filter gradient, 16x16 tiles, 64 x 64 channel blocks: the ANTI-PHASE form (round 4) -------------------------------------------
// The two-blocks-per-CU kernel above has the structure the forward kernel had until round 3 -- "DMA a tile -> wait -> 144 MFMAs ->
// barrier", two independent blocks drifting in and out of phase (matrix-pipe utilisation 0.56 in situ).  Here ONE 512-thread
// work-group holds two halves of four waves (waves w and w + 4 share a SIMD); the halves take alternate pixel tiles of the block's
// range into their own 73 KiB stage and run one phase apart, an s_barrier per phase:
//     phase 2k      A: 16 k-steps x 9 MFMAs on its tile k          B: LDS-DMA of its tile k
//     phase 2k + 1  A: LDS-DMA of its tile k + 1                   B: 16 k-steps x 9 MFMAs on its tile k
// so a SIMD's matrix pipe is fed by one wave at a time while its partner issues the DMA instructions.  There is no per-tile
// epilogue (the nine accumulators per wave run over all tiles); at the end B hands its accumulators to A through LDS and the block
// writes ONE partial filter -- half the partial filters (workspace traffic, reduction work) of two 256-thread blocks.
// The 20 transpose reads of the next k-step are pinned between the 9 MFMAs of the running one (two or four behind each): a burst
// in front of them stalls a lone wave's matrix pipe.
__device__ __forceinline__ void conv3x3_wgrad_pp_body(const unsigned short* __restrict__ x0, const unsigned short* __restrict__ dy,
                                                      float* __restrict__ ws, int B, int H, int W, int Cin, int Cout, MTile g,
                                                      int ntiles, int tiles_per_block, const int bx, const int by, const int bz,
                                                      const int gdx, const int gdy,
                                                      const unsigned short* __restrict__ x2 = nullptr, const int K1 = 0) {
    constexpr int TCI = 64, TCO = 64;
    const bool src2 = x2 != nullptr && by * TCI >= K1;       // concat-free input: see conv3x3_wgrad_body
    const unsigned short* __restrict__ x = src2 ? x2 : x0;
    const int xC = x2 == nullptr ? Cin : (src2 ? Cin - K1 : K1);
    const int xc0 = by * TCI - (src2 ? K1 : 0);
    constexpr int RBX = 128, RBD = 128, QX = 8, QD = 8, NPATCH = 324;
    constexpr int XI = (NPATCH * QX + 63) / 64, DI = 256 * QD / 64;      // 41 + 32 wave-instructions (1 KiB each) per tile
    constexpr int XN = (XI + 3) / 4, DN = DI / 4;
    constexpr int SD_OFF = XI * 1024, STAGE = SD_OFF + 256 * RBD;        // 74 752 B per half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int co0 = bz * TCO;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int half = wave >> 2, lw = wave & 3;
    const int wi = lw & 1, wj = lw >> 1;
    const unsigned sbase = (unsigned)(half * STAGE);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    unsigned dcon[2], xcon[2][3][2];             // per-lane LDS constants, see k_conv3x3_wgrad
    {
        const int c16 = lane & 15, cb16 = (lane >> 4) & 1, khalf = lane >> 5;
        const int chan_byte_x = (wi * 32 + cb16 * 16 + (c16 & 3) * 4) * 2;
        const int chan_byte_d = (wj * 32 + cb16 * 16 + (c16 & 3) * 4) * 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int lxr = khalf * 8 + r * 4 + (c16 >> 2);
            dcon[r] = sbase + (unsigned)(SD_OFF + wswz<RBD>(lxr, chan_byte_d));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int par = 0; par < 2; ++par)
                    xcon[r][kw][par] = sbase + (unsigned)((lxr + kw) * RBX + (chan_byte_x ^ (((((lxr + kw) >> 1) & 1) ^ par) << 6)));
        }
    }
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)B * H * W * xC * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * H * W * Cout * 2u), 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // DMA of tile t into this half's stage: lane `lane` of wave-instruction j = lw + 4 n fills LDS slot e = 64 j + lane with the
    // SOURCE piece p of patch pixel (px, py) / tile pixel m (the 64-byte half swizzle of the 128-byte rows is applied on the source
    // side); recomputed per tile from an opaque copy of the lane id (no registers to spare beside the MFMA stream)
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int tt = t;
        const int tx0 = (tt % g.tiles_x) << 4; tt /= g.tiles_x;
        const int ty0 = (tt % g.tiles_y) << 4; tt /= g.tiles_y;
        const int b0 = tt;
#pragma unroll
        for (int n = 0; n < XN; ++n) {
            const int j = lw + 4 * n;
            if (j < XI) {
                const int e = j * 64 + ln;
                const int pp = e >> 3, ps = e & 7;
                const int p = ps ^ (((pp >> 1) & 1) << 2);
                const int py = (int)(((unsigned)pp * 3641u) >> 16), px = pp - py * 18;        // pp / 18 for pp < 512
                const int gx = tx0 + px - 1, gy = ty0 + py - 1;
                const bool ok = pp < NPATCH && gx >= 0 && gx < W && gy >= 0 && gy < H;
                const unsigned vo = ok ? (unsigned)((((b0 * H + gy) * W + gx) * xC) * 2 + p * 16) : 0xffffffffu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + sbase + j * 1024), 16, vo, xc0 * 2, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < DN; ++n) {
            const int j = lw + 4 * n;
            const int e = j * 64 + ln;
            const int m = e >> 3, ps = e & 7;
            const int p = ps ^ (((m >> 1) & 1) << 2);
            const int ox = tx0 + (m & 15), oy = ty0 + (m >> 4);
            const bool ok = ox < W && oy < H;
            const unsigned vo = ok ? (unsigned)((((b0 * H + oy) * W + ox) * Cout) * 2 + p * 16) : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lds_ptr_t)(smem + sbase + SD_OFF + j * 1024), 16, vo, co0 * 2, 0, 0);
        }
    };
    // the 16 k-steps of a tile.  k-step s (tile row s) multiplies the dy fragment of row s with the x fragments of patch rows s, s + 1,
    // s + 2 (kh = 0, 1, 2) at the three column shifts kw: the fragment of (patch row R, kw) serves three k-steps, so it is read ONCE and
    // kept in a rotating window of four rows -- 8 transpose reads per step (one patch row + the dy row) instead of 20; with four waves
    // reading beside the partner half's DMA the 20-read form was bound by LDS bandwidth (5.1 K cycles a tile against 4.6 K of MFMA).
    // The reads of step s + 1 (dy first, then patch row s + 3, used last by the kh = 2 MFMAs) sit one behind each MFMA of step s.
    auto compute = [&]() __attribute__((always_inline)) {
        s16x4 fd[2][2], fx[4][3][2];
        auto rd_d = [&](auto stepc, auto rc) {
            constexpr int SI = decltype(stepc)::value, r = decltype(rc)::value;
            fd[SI & 1][r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(smem + (unsigned)(SI * 16 * RBD) + dcon[r]));
        };
        auto rd_x = [&](auto rowc, auto kwc, auto rc) {
            constexpr int R = decltype(rowc)::value, kw = decltype(kwc)::value, r = decltype(rc)::value;
            fx[R & 3][kw][r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(smem + (unsigned)(R * 18 * RBX) + xcon[r][kw][R & 1]));
        };
#define IC(v) std::integral_constant<int, (v)>()
        rd_d(IC(0), IC(0)); rd_d(IC(0), IC(1));
        {
            auto head = [&](auto self, auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 9) { rd_x(IC(i / 3), IC(i % 3), IC(0)); rd_x(IC(i / 3), IC(i % 3), IC(1)); self(self, IC(i + 1)); }
            };
            head(head, IC(0));
        }
        __builtin_amdgcn_sched_barrier(0);
        auto steps = [&](auto self, auto stepc) {
            constexpr int SI = decltype(stepc)::value;
            if constexpr (SI < 16) {
                constexpr int Bf = SI & 1;
                constexpr bool more = SI + 1 < 16;
                const s16x8 dtmp = {fd[Bf][0][0], fd[Bf][0][1], fd[Bf][0][2], fd[Bf][0][3], fd[Bf][1][0], fd[Bf][1][1], fd[Bf][1][2], fd[Bf][1][3]};
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, dtmp);
                auto slot = [&](auto kc) {
                    constexpr int k = decltype(kc)::value, kh = k / 3, kw = k % 3, Rs = (SI + kh) & 3;
                    const s16x8 atmp = {fx[Rs][kw][0][0], fx[Rs][kw][0][1], fx[Rs][kw][0][2], fx[Rs][kw][0][3],
                                        fx[Rs][kw][1][0], fx[Rs][kw][1][1], fx[Rs][kw][1][2], fx[Rs][kw][1][3]};
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, atmp), bfrag, acc[k], 0, 0, 0);
                    if constexpr (more) {
                        if constexpr (k < 2) rd_d(IC(SI + 1), kc);
                        else if constexpr (k < 8) rd_x(IC(SI + 3), IC((k - 2) >> 1), IC((k - 2) & 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                slot(IC(0)); slot(IC(1)); slot(IC(2)); slot(IC(3)); slot(IC(4)); slot(IC(5)); slot(IC(6)); slot(IC(7)); slot(IC(8));
                self(self, IC(SI + 1));
            }
        };
        steps(steps, IC(0));
#undef IC
    };
#define WPP_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
    const int t_begin = bx * tiles_per_block;
    const int t_end = min(ntiles, t_begin + tiles_per_block);
    const int n = t_end - t_begin;
    const int nA = (n + 1) >> 1, nh = half ? (n >> 1) : nA;       // tiles of A / of this half: t_begin + half + 2 k
    PHX_BLOCKLOG_BEGIN();
    if (half == 0 && nh > 0) load_tile(t_begin);
    WPP_BARRIER();
    if (half) {
        if (nh > 0) load_tile(t_begin + 1);
        WPP_BARRIER();
    }
    for (int k = 0; k < nA; ++k) {
        if (k < nh) compute();
        WPP_BARRIER();
        if (k + 1 < nh) load_tile(t_begin + half + 2 * (k + 1));
        WPP_BARRIER();
    }
    if (half == 0) WPP_BARRIER();
    // B -> A through LDS (both stages are free), then ONE partial filter per block: ws[(cblock * gdx + bx)][9][64][64]; C layout:
    // col = lane & 31 -> co, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> ci
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    f32x4_t* const pr = reinterpret_cast<f32x4_t*>(smem) + (size_t)lw * (9 * 4 * 64) + lane;
    if (half)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) pr[(k * 4 + q) * 64] = f32x4_t{acc[k][4 * q], acc[k][4 * q + 1], acc[k][4 * q + 2], acc[k][4 * q + 3]};
    WPP_BARRIER();
#undef WPP_BARRIER
    if (half == 0) {
        const size_t cb = (size_t)bz * gdy + by;
        float* wp = ws + (cb * gdx + bx) * (size_t)(9 * TCI * TCO);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t v = pr[(k * 4 + q) * 64];
                acc[k][4 * q] += v[0]; acc[k][4 * q + 1] += v[1]; acc[k][4 * q + 2] += v[2]; acc[k][4 * q + 3] += v[3];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cil = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wp[(k * TCI + cil) * TCO + wj * 32 + (lane & 31)] = acc[k][r];
            }
        }
    }
    PHX_BLOCKLOG_END();
}

__global__ __launch_bounds__(512, 1) void k_conv3x3_wgrad_pp(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                             float* __restrict__ ws, int B, int H, int W, int Cin, int Cout, MTile g,
                                                             int ntiles, int tiles_per_block, const unsigned short* __restrict__ x2,
                                                             int K1) {
    conv3x3_wgrad_pp_body(x, dy, ws, B, H, W, Cin, Cout, g, ntiles, tiles_per_block, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x,
                          gridDim.y, x2, K1);
}
__global__ __launch_bounds__(512, 1) void k_conv3x3_wgrad_pp_multi(const WgMJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgMJob j = jobs[lo];
    // XCD-aware order (work-group id mod 8 is the XCD; every job starts on a multiple of 8): the gdy x gdz channel blocks that read the
    // same pixel slice are 8 ids apart -- one XCD, dispatched together -- so the slice is fetched into ONE L2 instead of up to four
    const int local = blockIdx.x - j.blk0;
    const int cb = j.gdy * j.gdz, t = local >> 3;
    // (wgrad_plan gives deferred jobs a multiple of 8 slices whenever the layer has that many tiles: padding holes leave XCDs idle --
    // work-groups are bound to XCDs statically -- and cost 0.81 ms instead of 0.65 for the launch)
    const int c = t % cb, bx = (t / cb) * 8 + (local & 7);
    if (bx >= j.gdx) return;
    conv3x3_wgrad_pp_body(j.x, j.dy, j.ws, j.B, j.H, j.W, j.Cin, j.Cout, j.g, j.ntiles, j.tpb, bx, c % j.gdy, c / j.gdy, j.gdx, j.gdy,
                          j.x2, j.K1);
}

// dw[k][ci][co] += sum over the nslice partial tiles written by the filter-gradient kernels.  A thread owns four
// consecutive co entries (16-byte loads); block = 64 such quads x 4 slice groups; gridDim.y further splits the slices
// (one atomic per entry per y-block); four slices are in flight per thread.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, int nslice, int Cin,
                                                  int Cout, int tci, int tco, int bx, int by, int ny) {
    const int tile_elems = 9 * tci * tco;
    const int ntile_ci = Cin / tci;
    const size_t total = (size_t)9 * Cin * Cout;
    const size_t i = ((size_t)bx * 64 + (threadIdx.x & 63)) * 4;
    const int sg = threadIdx.x >> 6;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < total) {
        const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), k = (int)(i / ((size_t)Cout * Cin));
        const int cb = (co / tco) * ntile_ci + (ci / tci);
        const float* p = ws + (size_t)cb * nslice * tile_elems + (k * tci + ci % tci) * tco + co % tco;
        const int step = ny * 4;
        int sidx = by * 4 + sg;
        for (; sidx + 3 * step < nslice; sidx += 4 * step) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)sidx * tile_elems);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(sidx + step) * tile_elems);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(sidx + 2 * step) * tile_elems);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(sidx + 3 * step) * tile_elems);
            a += (v0 + v1) + (v2 + v3);
        }
        for (; sidx < nslice; sidx += step) a += *reinterpret_cast<const f32x4*>(p + (size_t)sidx * tile_elems);
    }
    __shared__ f32x4 red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    if (sg == 0 && i < total) {
        const f32x4 r = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(&dw[i + q], r[q]);
    }
}
__global__ void k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int nslice, int Cin, int Cout,
                               int tci, int tco) {
    wgrad_reduce_body(ws, dw, nslice, Cin, Cout, tci, tco, blockIdx.x, blockIdx.y, gridDim.y);
}
// The reductions are leaves of the backward graph: instead of one small launch behind every filter-gradient kernel (81 per
// step, ~10 us each on the latency-bound small-map chains) ONE launch at the end of the backward pass sums the partial
// filters of all layers.  jobs[j].blk0 = first block of job j in the flat grid (ascending).
struct WgrJob {
    const float* ws; float* dw;
    int nslice, Cin, Cout, tci, tco, gx, gy, blk0;
};
__global__ void k_wgrad_reduce_multi(const WgrJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                         // last job with blk0 <= blockIdx.x (uniform per block)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgrJob j = jobs[lo];
    const int local = blockIdx.x - j.blk0;
    wgrad_reduce_body(j.ws, j.dw, j.nslice, j.Cin, j.Cout, j.tci, j.tco, local % j.gx, local / j.gx, j.gy);
}


// debug pointers of this translation unit (called by phx_debug_set_trace / _blocklog in conv_mfma.hip)
int phx_wgrad_set_debug(void* trace_buf, void* blocklog_buf, int which) {
    unsigned long long* p = (unsigned long long*)(which == 0 ? trace_buf : blocklog_buf);
    if (which == 0) PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phx_trace), &p, sizeof(p)));
    else PHX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phx_blocklog), &p, sizeof(p)));
    return PHX_OK;
}

extern "C" {

static bool wgrad_dma_enabled() { return true; }
// dynamic LDS of the LDS-DMA filter-gradient kernels: the staged tile, or the 2 x 36 KiB of the wave-group reduction
static size_t wgrad_dma_lds(int tci, int tco) {
    if (tci == 64 && tco == 64) return (size_t)2 * (41 * 1024 + 256 * 128);      // k_conv3x3_wgrad_pp: a stage per half
    const size_t stage = (size_t)((324 * (tci / 8) + 63) / 64) * 1024 + (size_t)256 * tco * 2;
    const size_t red = (tci == 64 && tco == 64) ? 0 : (size_t)2 * 9 * 16 * 64 * sizeof(float);
    return stage > red ? stage : red;
}
static int wgrad_plan(int B, int H, int W, int Cin, int Cout, MTile* g, int* tci, int* tco, int* gx, int* tpb, int* wk,
                      int target_override = 0, int K1 = 0) {
    *g = make_mtile(B, H, W);
    const int ntiles = g->tiles_x * g->tiles_y * g->tiles_b;
    *tci = Cin % 64 == 0 ? 64 : 32;
    if (K1 > 0 && K1 % 64 != 0) *tci = 32;         // concat-free input: a block's input channels lie in ONE of the two tensors
    *tco = Cout % 64 == 0 ? 64 : 32;
    *wk = 4 / ((*tci / 32) * (*tco / 32));
    if (wgrad_dma_enabled() && g->tws == 4 && g->ths == 4 && g->tb == 1) *wk = 1;   // the LDS-DMA kernel sums its wave groups in LDS
    const int cblocks = (Cin / *tci) * (Cout / *tco);
    // measured on MI355X (tools/bench_wgrad.py, LDS-DMA kernels, two blocks per CU): ~384 blocks; 512 when few channel
    // blocks share the pixel tiles; fewer when there are few pixel tiles, because every block writes (and k_wgrad_reduce
    // re-reads) a full 9*TCI*TCO partial filter
    int target_blocks = cblocks <= 4 && *tci == 64 && *tco == 64 ? 512 : 384;
    if (ntiles <= 256 && target_blocks > 256) target_blocks = 256;
    // the anti-phase kernel (64 x 64 channel blocks on 16 x 16 tiles) runs ONE 512-thread work-group per CU
    const bool pp = wgrad_dma_enabled() && g->tws == 4 && g->ths == 4 && g->tb == 1 && *tci == 64 && *tco == 64;
    if (pp) target_blocks = 256;
    // deferred multi-layer launches (see below); a 512-thread anti-phase block is two of the 256-thread blocks the target counts
    // (measured in situ, pp jobs at 96 / 64 / 48 / 32: launch 0.690 / 0.669 / 0.633 / 0.723 ms, reduction 0.276 / 0.254 / 0.226 / 0.222 ms)
    if (target_override > 0) target_blocks = pp ? (target_override + 1) / 2 : target_override;
    int split = (target_blocks + cblocks - 1) / cblocks;
    if (pp && target_override <= 0) split = target_blocks / cblocks;              // (never a second round of a few blocks)
    if (pp && target_override > 0) {
        // whole slices per XCD (k_conv3x3_wgrad_pp_multi) -- but never more slices than the stand-alone plan takes (256 / cblocks):
        // callers size the workspace with phx_conv3x3_wgrad_ws_bytes, i.e. from the stand-alone split (>= 33 channel blocks, e.g.
        // 384 -> 384: the round-up to 8 exceeded the stand-alone 7 and the job rejected its own workspace)
        const int alone = 256 / cblocks > 0 ? 256 / cblocks : 1;
        split = (split + 7) & ~7;
        if (split > alone) split = alone;
    }
    if (split > ntiles) split = ntiles;
    if (split < 1) split = 1;
    *tpb = (ntiles + split - 1) / split;
    *gx = (ntiles + *tpb - 1) / *tpb;
    return ntiles;
}

size_t phx_conv3x3_wgrad_ws_bytes_dual(int B, int H, int W, int Cin, int Cout, int K1) {
    MTile g; int tci, tco, gx, tpb, wk;
    wgrad_plan(B, H, W, Cin, Cout, &g, &tci, &tco, &gx, &tpb, &wk, 0, K1);
    return (size_t)(Cin / tci) * (Cout / tco) * gx * wk * 9 * tci * tco * sizeof(float);
}
size_t phx_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout) { return phx_conv3x3_wgrad_ws_bytes_dual(B, H, W, Cin, Cout, 0); }

static int wgrad_atomic_tiles() {
    return phx_deterministic() ? 0 : 4;        // deterministic mode: always partial filters + ordered reduction
}
static void wgrad_reduce_geometry(int Cin, int Cout, int nslice, int* rgx, int* rgy) {
    const size_t total = (size_t)9 * Cin * Cout;
    int gy = nslice / 16;
    if (gy < 1 || phx_deterministic()) gy = 1;     // (gy > 1: several blocks add into one filter element)
    if (gy > 16) gy = 16;
    *rgx = (int)((total / 4 + 63) / 64);
    *rgy = gy;
}
/* plan6 = {uses_workspace, nslice, tci, tco, reduce grid x, reduce grid y} of the launch phx_conv3x3_wgrad_mfma_bf16 makes */
int phx_conv3x3_wgrad_reduce_plan(int B, int H, int W, int Cin, int Cout, int* plan6) {
    return phx_conv3x3_wgrad_reduce_plan_dual(B, H, W, Cin, Cout, 0, plan6);
}
int phx_conv3x3_wgrad_reduce_plan_dual(int B, int H, int W, int Cin, int Cout, int K1, int* plan6) {
    MTile g; int tci, tco, gx, tpb, wk;
    const int ntiles = wgrad_plan(B, H, W, Cin, Cout, &g, &tci, &tco, &gx, &tpb, &wk, 0, K1);
    plan6[0] = ntiles > wgrad_atomic_tiles();
    plan6[1] = gx * wk; plan6[2] = tci; plan6[3] = tco;
    wgrad_reduce_geometry(Cin, Cout, gx * wk, &plan6[4], &plan6[5]);
    return PHX_OK;
}
int phx_wgrad_reduce_multi(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    PHX_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, PHX_E_INVAL, "wgrad_reduce_multi: empty job list");
    hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const WgrJob*)jobs_dev, njobs);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
static int wgrad_impl(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes, int B, int H,
                      int W, int Cin, int Cout, bool reduce, void* stream, const void* x2 = nullptr, int K1 = 0,
                      const float* xscale = nullptr, const float* xshift = nullptr);
/* Deferred small-map filter gradients (see k_conv3x3_wgrad_multi).  phx_conv3x3_wgrad_multi_job fills ONE job record of
 * phx_conv3x3_wgrad_multi_job_bytes() bytes in HOST memory for the launch phx_conv3x3_wgrad_mfma_bf16_partial would make;
 * info = {variant (0: not deferred), blocks, dynamic LDS bytes, uses_workspace, nslice, tci, tco, reduce grid x, y} (9 ints).  The caller concatenates the records of one variant (blk0 = running sum of blocks), copies them to the
 * device and calls phx_conv3x3_wgrad_multi once. */
int phx_conv3x3_wgrad_multi_job_bytes(void) { return (int)sizeof(WgMJob); }
int phx_conv3x3_wgrad_multi_job(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes, int B,
                                int H, int W, int Cin, int Cout, int blocks_target, int blk0, void* job_out, int* info4) {
    return phx_conv3x3_wgrad_multi_job_dual(x, nullptr, 0, dy, dw_hwio, workspace, workspace_bytes, B, H, W, Cin, Cout, blocks_target, blk0,
                                            job_out, info4);
}
int phx_conv3x3_wgrad_multi_job_dual(const void* x, const void* x2, int K1, const void* dy, float* dw_hwio, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int blocks_target, int blk0,
                                     void* job_out, int* info4) {
    PHX_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, PHX_E_SHAPE, "conv3x3_wgrad_multi_job: Cin % 32 == 0 and Cout % 32 == 0 required");
    PHX_REQUIRE(x2 == nullptr || (K1 > 0 && K1 < Cin && K1 % 32 == 0), PHX_E_SHAPE, "conv3x3_wgrad_multi_job: 0 < K1 < Cin, K1 % 32 == 0");
    if (x2 == nullptr) K1 = 0;
    MTile g; int tci, tco, gx, tpb, wk;
    // blocks_target > 0: pixel-tile split of THIS job (a multi-layer launch has thousands of blocks in all, so a layer needs far
    // fewer partial filters than when it runs alone -- less workspace traffic for the launch and for the reduction)
    const int ntiles = wgrad_plan(B, H, W, Cin, Cout, &g, &tci, &tco, &gx, &tpb, &wk, blocks_target, K1);
    for (int i = 0; i < 9; ++i) info4[i] = 0;
    const bool fast16 = g.tws == 4 && g.ths == 4 && g.tb == 1;
    if (fast16) {
        // 16x16 tiles: the LDS-DMA kernel with a workspace; measured: deferring up to 1024 tiles (H <= 64 at batch 64) helps,
        // the 128x128 layers are as fast inline (their inputs are still in the Infinity Cache right after the backward
        // normalisation pass)
        const int dtl = 1024;
        if (!workspace || ntiles > dtl || ntiles <= wgrad_atomic_tiles() || !wgrad_dma_enabled()) return PHX_OK;
    }
    const int npatch = g.tb * ((1 << g.ths) + 2) * ((1 << g.tws) + 2);
    const bool use_ws = workspace != nullptr && ntiles > wgrad_atomic_tiles();
    if (use_ws)      // the partial filters of THIS plan (blocks_target may split the pixel tiles finer than the stand-alone launch)
        PHX_REQUIRE(workspace_bytes >= (size_t)(Cin / tci) * (Cout / tco) * gx * wk * 9 * tci * tco * sizeof(float), PHX_E_INVAL,
                    "conv3x3_wgrad_multi_job: workspace too small for this blocks_target");
    WgMJob j;
    j.x = (const unsigned short*)x; j.dy = (const unsigned short*)dy; j.dw = dw_hwio; j.ws = use_ws ? (float*)workspace : nullptr;
    j.B = B; j.H = H; j.W = W; j.Cin = Cin; j.Cout = Cout; j.g = g; j.ntiles = ntiles; j.tpb = tpb;
    j.gdx = gx; j.gdy = Cin / tci; j.gdz = Cout / tco; j.blk0 = blk0;
    j.x2 = (const unsigned short*)x2; j.K1 = K1; j.pad_ = 0;
    memcpy(job_out, &j, sizeof(j));
    info4[0] = 1 + (tco == 64 ? 1 : 0) + (tci == 64 ? 2 : 0) + (fast16 ? 8 : npatch > 400 ? 4 : 0);     // 9..12: LDS-DMA kernels
    info4[1] = j.gdx * j.gdy * j.gdz;
    if (info4[0] == 12) info4[1] = (j.gdx + 7) / 8 * 8 * j.gdy * j.gdz;      // k_conv3x3_wgrad_pp_multi: slices padded to the 8 XCDs
    info4[2] = fast16 ? (int)wgrad_dma_lds(tci, tco) : npatch * tci * 2 + 256 * tco * 2;
    info4[3] = use_ws;
    info4[4] = gx * wk; info4[5] = tci; info4[6] = tco;                       // reduction job of this launch (phx_wgrad_reduce_multi)
    wgrad_reduce_geometry(Cin, Cout, gx * wk, &info4[7], &info4[8]);
    return PHX_OK;
}
int phx_conv3x3_wgrad_multi(const void* jobs_dev, int njobs, int total_blocks, int variant, size_t lds_bytes, void* stream) {
    PHX_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && variant >= 1 && variant <= 12, PHX_E_INVAL, "conv3x3_wgrad_multi: bad arguments");
    static bool attr_set = false;
#define WM_ATTR(A, Bq, C) PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_multi<A, Bq, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    if (!attr_set) {
        WM_ATTR(32, 32, false); WM_ATTR(32, 64, false); WM_ATTR(64, 32, false); WM_ATTR(64, 64, false);
        WM_ATTR(32, 32, true); WM_ATTR(32, 64, true); WM_ATTR(64, 32, true); WM_ATTR(64, 64, true);
#define WMD_ATTR(A, Bq) PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_dma_multi<A, Bq>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        WMD_ATTR(32, 32); WMD_ATTR(32, 64); WMD_ATTR(64, 32);
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_pp_multi, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#undef WMD_ATTR
        attr_set = true;
    }
#undef WM_ATTR
#define WM_LAUNCH(A, Bq, C)                                                                                           \
    hipLaunchKernelGGL((k_conv3x3_wgrad_multi<A, Bq, C>), dim3((unsigned)total_blocks), dim3(256), lds_bytes,          \
                       (hipStream_t)stream, (const WgMJob*)jobs_dev, njobs)
    switch (variant - 1) {
        case 0: WM_LAUNCH(32, 32, false); break;
        case 1: WM_LAUNCH(32, 64, false); break;
        case 2: WM_LAUNCH(64, 32, false); break;
        case 3: WM_LAUNCH(64, 64, false); break;
        case 4: WM_LAUNCH(32, 32, true); break;
        case 5: WM_LAUNCH(32, 64, true); break;
        case 6: WM_LAUNCH(64, 32, true); break;
        case 7: WM_LAUNCH(64, 64, true); break;
#define WMD_LAUNCH(A, Bq)                                                                                             \
    hipLaunchKernelGGL((k_conv3x3_wgrad_dma_multi<A, Bq>), dim3((unsigned)total_blocks), dim3(256), lds_bytes,         \
                       (hipStream_t)stream, (const WgMJob*)jobs_dev, njobs)
        case 8: WMD_LAUNCH(32, 32); break;
        case 9: WMD_LAUNCH(32, 64); break;
        case 10: WMD_LAUNCH(64, 32); break;
        default:
            hipLaunchKernelGGL(k_conv3x3_wgrad_pp_multi, dim3((unsigned)total_blocks), dim3(512), lds_bytes, (hipStream_t)stream,
                               (const WgMJob*)jobs_dev, njobs);
            break;
#undef WMD_LAUNCH
    }
#undef WM_LAUNCH
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}
int phx_conv3x3_wgrad_mfma_bf16(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes,
                                int B, int H, int W, int Cin, int Cout, void* stream) {
    return wgrad_impl(x, dy, dw_hwio, workspace, workspace_bytes, B, H, W, Cin, Cout, true, stream);
}
int phx_conv3x3_wgrad_mfma_bf16_partial(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes,
                                        int B, int H, int W, int Cin, int Cout, void* stream) {
    return wgrad_impl(x, dy, dw_hwio, workspace, workspace_bytes, B, H, W, Cin, Cout, false, stream);
}
// Filter gradient of a convolution that read the PRE-normalisation tensor of the layer in front (phx_conv3x3_mfma_bf16_xf): x is that
// tensor, xscale / xshift [Cin] the producer's coefficients; the LDS-DMA kernel with 32-channel input blocks re-forms
// a = relu(x * xscale + xshift) in place in its staged patch.  Stand-alone launch, partial filters into the workspace, no reduction
// (as phx_conv3x3_wgrad_mfma_bf16_partial).
int phx_conv3x3_wgrad_xf_supported(int B, int H, int W, int Cin, int Cout) {
    if (Cin != 32 || Cout % 32 != 0 || !wgrad_dma_enabled()) return 0;
    MTile g; int tci, tco, gx, tpb, wk;
    const int ntiles = wgrad_plan(B, H, W, Cin, Cout, &g, &tci, &tco, &gx, &tpb, &wk, 0, 0);
    return (tci == 32 && g.tws == 4 && g.ths == 4 && g.tb == 1 && ntiles > 1024) ? 1 : 0;      // (> 1024 tiles: never a deferred multi-layer job)
}
int phx_conv3x3_wgrad_mfma_bf16_partial_xf(const void* x, const float* xscale, const float* xshift, const void* dy, float* dw_hwio,
                                           void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream) {
    PHX_REQUIRE(xscale != nullptr && xshift != nullptr && workspace != nullptr && phx_conv3x3_wgrad_xf_supported(B, H, W, Cin, Cout), PHX_E_SHAPE,
                "conv3x3_wgrad_partial_xf: xscale / xshift / workspace, shape see phx_conv3x3_wgrad_xf_supported");
    return wgrad_impl(x, dy, dw_hwio, workspace, workspace_bytes, B, H, W, Cin, Cout, false, stream, nullptr, 0, xscale, xshift);
}
int phx_conv3x3_wgrad_mfma_bf16_dual(const void* x, const void* x2, int K1, const void* dy, float* dw_hwio, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int reduce, void* stream) {
    PHX_REQUIRE(x2 != nullptr && K1 > 0 && K1 < Cin && K1 % 32 == 0, PHX_E_SHAPE, "conv3x3_wgrad_mfma_dual: x2, 0 < K1 < Cin, K1 % 32 == 0");
    return wgrad_impl(x, dy, dw_hwio, workspace, workspace_bytes, B, H, W, Cin, Cout, reduce != 0, stream, x2, K1);
}
static int wgrad_impl(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes, int B, int H,
                      int W, int Cin, int Cout, bool reduce, void* stream, const void* x2, int K1, const float* xscale,
                      const float* xshift) {
    PHX_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, PHX_E_SHAPE, "conv3x3_wgrad_mfma: Cin % 32 == 0 and Cout % 32 == 0 required");
    if (x2 == nullptr) K1 = 0;
    MTile g; int tci, tco, gx, tpb, wk;
    const int ntiles = wgrad_plan(B, H, W, Cin, Cout, &g, &tci, &tco, &gx, &tpb, &wk, 0, K1);
    const int tw = 1 << g.tws, th = 1 << g.ths;
    const int npatch = g.tb * (th + 2) * (tw + 2);
    float* ws = nullptr;
    if (workspace) {
        PHX_REQUIRE(workspace_bytes >= phx_conv3x3_wgrad_ws_bytes_dual(B, H, W, Cin, Cout, K1), PHX_E_INVAL, "conv3x3_wgrad_mfma: workspace too small");
        ws = (float*)workspace;
        // a handful of pixel tiles (H <= 4 at batch 64): the partial filters are few, so adding them straight into dw with
        // atomics beats the extra k_wgrad_reduce launch on the latency-bound small-map chains (26 -> 20 us at 4 x 4)
        if (ntiles <= wgrad_atomic_tiles()) ws = nullptr;
    }
    static bool attr_set = false;
    if (!attr_set) {
#define WG_ATTR(A, Bq, C, F)                                                                                          \
    PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad<A, Bq, C, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        WG_ATTR(64, 64, false, false); WG_ATTR(64, 32, false, false); WG_ATTR(32, 64, false, false); WG_ATTR(32, 32, false, false);
        WG_ATTR(64, 64, true, false); WG_ATTR(64, 32, true, false); WG_ATTR(32, 64, true, false); WG_ATTR(32, 32, true, false);
        WG_ATTR(64, 64, false, true); WG_ATTR(64, 32, false, true); WG_ATTR(32, 64, false, true); WG_ATTR(32, 32, false, true);
#undef WG_ATTR
        attr_set = true;
    }
    // 16x16 tiles with a workspace: the LDS-DMA kernel (two blocks per CU)
    if (wgrad_dma_enabled() && ws && g.tws == 4 && g.ths == 4 && g.tb == 1) {
        static bool dattr = false;
#define WD_ATTR(A, Bq) PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_dma<A, Bq>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        if (!dattr) {
            WD_ATTR(64, 32); WD_ATTR(32, 64); WD_ATTR(32, 32);
            PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_pp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            dattr = true;
        }
#undef WD_ATTR
#define WD_LAUNCH(A, Bq)                                                                                              \
    hipLaunchKernelGGL((k_conv3x3_wgrad_dma<A, Bq>), dim3(gx, Cin / A, Cout / Bq), dim3(256),                         \
                       wgrad_dma_lds(A, Bq), (hipStream_t)stream,                                                     \
                       (const unsigned short*)x, (const unsigned short*)dy, ws, B, H, W, Cin, Cout, g, ntiles, tpb,    \
                       (const unsigned short*)x2, K1, xscale, xshift)
        PHX_REQUIRE(xscale == nullptr || tci == 32, PHX_E_SHAPE, "conv3x3_wgrad_xf: 32-channel input blocks only");
        if (tci == 64 && tco == 64)
            hipLaunchKernelGGL(k_conv3x3_wgrad_pp, dim3(gx, Cin / 64, Cout / 64), dim3(512), wgrad_dma_lds(64, 64), (hipStream_t)stream,
                               (const unsigned short*)x, (const unsigned short*)dy, ws, B, H, W, Cin, Cout, g, ntiles, tpb,
                               (const unsigned short*)x2, K1);
        else if (tci == 64) WD_LAUNCH(64, 32);
        else if (tco == 64) WD_LAUNCH(32, 64);
        else WD_LAUNCH(32, 32);
#undef WD_LAUNCH
        PHX_CHECK_LAUNCH();
    } else {
    PHX_REQUIRE(xscale == nullptr, PHX_E_SHAPE, "conv3x3_wgrad_xf: the LDS-DMA kernel's shapes only (16 x 16 tiles, workspace)");
    const size_t sh = (size_t)npatch * tci * 2 + (size_t)256 * tco * 2;
#define WG_LAUNCH(A, Bq, C, F)                                                                                            \
    hipLaunchKernelGGL((k_conv3x3_wgrad<A, Bq, C, F>), dim3(gx, Cin / A, Cout / Bq), dim3(256), sh, (hipStream_t)stream,  \
                       (const unsigned short*)x, (const unsigned short*)dy, dw_hwio, ws, B, H, W, Cin, Cout, g, ntiles, tpb, \
                       (const unsigned short*)x2, K1)
#define WG_LAUNCH2(A, Bq)                                                                  \
    do {                                                                                   \
        if (g.tws == 4 && g.ths == 4 && g.tb == 1) WG_LAUNCH(A, Bq, false, true);          \
        else if (npatch <= 400) WG_LAUNCH(A, Bq, false, false);                            \
        else WG_LAUNCH(A, Bq, true, false);                                                \
    } while (0)
    if (tci == 64 && tco == 64) WG_LAUNCH2(64, 64);
    else if (tci == 64) WG_LAUNCH2(64, 32);
    else if (tco == 64) WG_LAUNCH2(32, 64);
    else WG_LAUNCH2(32, 32);
#undef WG_LAUNCH2
#undef WG_LAUNCH
    PHX_CHECK_LAUNCH();
    }
    if (ws && reduce) {
        const int nslice = gx * wk;
        int rgx, rgy;
        wgrad_reduce_geometry(Cin, Cout, nslice, &rgx, &rgy);
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)rgx, rgy), dim3(256), 0, (hipStream_t)stream, ws, dw_hwio, nslice,
                           Cin, Cout, tci, tco);
        PHX_CHECK_LAUNCH();
    }
    return PHX_OK;
}

}  // extern "C"

