// Shared by the convolution translation units (conv_mfma.hip: forward / data gradient, conv_wgrad.hip: filter gradient): tile
// geometry of the 256-pixel kernels, LDS constants, debug stamps.  The debug pointers are per translation unit (no relocatable
// device code): phx_debug_set_trace / _blocklog set every copy.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "phx_common.h"
#include <cstring>

#ifndef PHX_ABLATE      // dev only (tools/build_variant.sh): fwd 1 no global loads, 2 no LDS staging stores, 4 no MFMAs, 8 no output stores; wgrad 16 no global loads, 32 no MFMAs
#define PHX_ABLATE 0
#endif
#ifndef PHX_FRAG_DEPTH  // operand-fragment prefetch distance (in 4-MFMA groups) of the 256-pixel kernels
#define PHX_FRAG_DEPTH 2
#endif
#define KC 32            // input channels per LDS stage (two MFMA k-steps)
#define ROWB 80          // bytes per pixel / filter row in LDS: 32 bf16 + 16 B pad -> conflict-free ds_read_b128
// Pitch of one 18-pixel patch row of the 16-wide tiles.  A ds_read_b128 is served in 16-lane groups {0-3, 12-15, 20-27}, ...:
// twelve pixels of one tile row and four of the next.  With 80-byte pixels the 16-byte slot (mod 256 B) of pixel p is 5p
// mod 16, a permutation of a row's 16 pixels; the second row's pixels fill exactly the first row's gaps when the row pitch
// is a multiple of 256 B.  18 * 80 = 1440 is not (two 2-way conflicts per group: measured 40 % of the LDS cycles); 1536 is.
#define PITCH16 1536

struct MTile {
    int tws, ths, tb, tiles_x, tiles_y, tiles_b;
    unsigned mpw, mpp;      // ceil(2^20 / (tw + 2)), ceil(2^20 / ((tw + 2) (th + 2))): exact quotients for dividends < 4096
    int rpitch, ipitch;     // forward / data-gradient LDS image of the small-map tiles: bytes per patch row / per image patch
};
// halo-patch index -> (x, y, batch) by multiply-shift: a runtime integer division costs ~40 instructions, and the staging
// plans of the small-map tiles do three per 16-byte piece (48 per thread -- microseconds of a launch that has 1 us of MFMAs)
static void mtile_magic(MTile* g) {
    const unsigned pw = (1u << g->tws) + 2, ph = (1u << g->ths) + 2;
    g->mpw = ((1u << 20) + pw - 1) / pw;
    g->mpp = ((1u << 20) + pw * ph - 1) / (pw * ph);
    // LDS pitches of the packed small-map tiles (several images per 256-pixel tile).  A ds_read_b128 is served in the 16-lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (x 2 half waves) and is conflict-free when a group's 16 pixels fall on the 16 different
    // 16-byte slots of a 256-byte bank row.  With 80-byte pixels (slot 5 x mod 16) the DENSE images are 3- to 4-way conflicted (a
    // group spans two to four tile rows / images whose slot sets collide: measured 50-55 % of the LDS cycles, and the operand reads
    // are what bounds these kernels); padded to the pitches below -- found by enumeration -- every group is a permutation.
    g->rpitch = (int)pw * ROWB;
    g->ipitch = (int)ph * g->rpitch;
    if (g->tws == 3 && g->ths == 3) { g->rpitch = 56 * 16; g->ipitch = 560 * 16; }
    else if (g->tws == 2 && g->ths == 2) { g->rpitch = 36 * 16; g->ipitch = 224 * 16; }
    else if (g->tws == 1 && g->ths == 1) { g->rpitch = 22 * 16; g->ipitch = 92 * 16; }
}
__device__ __forceinline__ void patch_coords(const MTile& g, int pp, int pw, int ph, int* px, int* py, int* pb) {
    // (24-bit multiplies: full rate, v_mul_lo_u32 is quarter rate; every operand here is < 2^21)
    const int b = (int)(__umul24((unsigned)pp, g.mpp) >> 20);
    const int rem = pp - (int)__umul24((unsigned)b, (unsigned)(pw * ph));
    const int y = (int)(__umul24((unsigned)rem, g.mpw) >> 20);
    *pb = b; *py = y; *px = rem - (int)__umul24((unsigned)y, (unsigned)pw);
}
static MTile make_mtile(int B, int H, int W) {
    MTile g;
    int tw = 1, th = 1;
    g.tws = g.ths = 0;
    while (tw < W && tw < 16) { tw <<= 1; g.tws++; }
    while (th < H && th < 16) { th <<= 1; g.ths++; }
    g.tb = 256 / (tw * th);
    g.tiles_x = (W + tw - 1) / tw;
    g.tiles_y = (H + th - 1) / th;
    g.tiles_b = (B + g.tb - 1) / g.tb;
    mtile_magic(&g);
    return g;
}


static __device__ unsigned long long* g_phx_trace = nullptr;      // debug: phase timestamps of block (0,0,0), thread 0
static __device__ unsigned long long* g_phx_blocklog = nullptr;   // debug: per-block {start, end, HW_ID | XCC_ID << 32, realtime}
#define PHX_BLOCKLOG_BEGIN() const unsigned long long bl_t0 = g_phx_blocklog ? __builtin_readcyclecounter() : 0ull
#define PHX_BLOCKLOG_END()                                                                               \
    do {                                                                                                 \
        if (g_phx_blocklog && threadIdx.x == 0) {                                                        \
            const size_t bi = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;    \
            g_phx_blocklog[bi * 4 + 0] = bl_t0;                                                          \
            g_phx_blocklog[bi * 4 + 1] = __builtin_readcyclecounter();                                   \
            g_phx_blocklog[bi * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |          \
                                         ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);   \
            g_phx_blocklog[bi * 4 + 3] = wall_clock64();                                                 \
        }                                                                                                \
    } while (0)
#define PHX_TRACE(slot)                                                                                  \
    do {                                                                                                 \
        if (g_phx_trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)    \
            g_phx_trace[slot] = __builtin_readcyclecounter();                                            \
    } while (0)

