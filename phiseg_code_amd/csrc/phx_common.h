// Common device/host helpers for libphx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/phx.h"

struct bf16_t {
    unsigned short u;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                  // RNE
    return (unsigned short)(u >> 16);
}

// two floats -> packed bf16 pair {lo = a, hi = b}: ONE v_cvt_pk_bf16_f32 on gfx950 (hardware round-to-nearest-even)
typedef __attribute__((ext_vector_type(2))) float phx_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 phx_bf16x2;
__device__ __forceinline__ unsigned f2bf_pk(float a, float b) {
    const phx_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, phx_bf16x2));
}

#ifndef PHX_TILE_BANDS  // XCD k (work-group id mod 8) takes the k-th contiguous eighth of a launch: bit 0 the convolutions' pixel tiles (phx_band8), bit 1 the resize kernels' blocks; 0: round-robin (dev A/B builds, tools/build_variant.sh)
#define PHX_TILE_BANDS 3
#endif
// Slot s of a launch's tile sequence (slots go round-robin over the 8 XCDs: XCD = s mod 8, each with its own L2) -> pixel tile.
// Neighbouring tiles share a one-pixel halo (an 18 x 18 patch for a 16 x 16 tile: 27 % more than the tile, 34 x 18 for 32 x 16:
// 20 %); handed out round-robin, the four neighbours of a tile sit in four other L2s and every halo comes from HBM again (measured,
// round 5: the convolution family fetches 1.25x its algorithmic bytes).  Banded, XCD k walks tiles k n/8 .. (k + 1) n/8 - 1 in
// row-major order: the horizontal neighbour is the next slot of the same XCD, the vertical one tiles_x slots later -- in flight
// in the same L2 at the same time.  The last n mod 8 tiles keep their slots.  Placement only: the tile -> value map is untouched.
__device__ __forceinline__ int phx_band8(int s, int n) {
    return ((PHX_TILE_BANDS & 1) && s < (n & ~7)) ? (s & 7) * (n >> 3) + (s >> 3) : s;
}

template <typename T> __device__ __forceinline__ float ldf(const T* p, size_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, size_t i) { return bf2f(p[i].u); }
template <typename T> __device__ __forceinline__ void stf(T* p, size_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, size_t i, float v) { p[i].u = f2bf(v); }
// value as it will read back after storage in T (bf16 rounding), for statistics consistency
template <typename T> __device__ __forceinline__ float roundf_as(float v);
template <> __device__ __forceinline__ float roundf_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float roundf_as<bf16_t>(float v) { return bf2f(f2bf(v)); }

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == PHX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == PHX_ACT_SOFTPLUS) return v > 20.f ? v : log1pf(expf(v));
    return v;
}
// derivative of act evaluated from the PRE-activation value
__device__ __forceinline__ float act_grad_pre(float pre, int act) {
    if (act == PHX_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
    if (act == PHX_ACT_SOFTPLUS) return 1.f / (1.f + expf(-pre));
    return 1.f;
}
// derivative of act evaluated from the stored OUTPUT value
__device__ __forceinline__ float act_grad_out(float y, int act) {
    if (act == PHX_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == PHX_ACT_SOFTPLUS) return 1.f - expf(-y);
    return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- host side ------------------------------------------------------------------------------------
void phx_set_error(const char* fmt, ...);
#define PHX_CHECK_HIP(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            phx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PHX_E_RUNTIME;                                                              \
        }                                                                                      \
    } while (0)
#define PHX_CHECK_LAUNCH()                                                                          \
    do {                                                                                            \
        hipError_t _e = hipGetLastError();                                                          \
        if (_e != hipSuccess) {                                                                     \
            phx_set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e));  \
            return PHX_E_LAUNCH;                                                                    \
        }                                                                                           \
    } while (0)
#define PHX_REQUIRE(cond, code, msg)                              \
    do {                                                          \
        if (!(cond)) {                                            \
            phx_set_error("%s:%d %s", __FILE__, __LINE__, msg);   \
            return code;                                          \
        }                                                         \
    } while (0)

// PHX_DETERMINISTIC=1 (read once): every cross-block floating-point reduction takes a fixed summation order -- single-block or
// one-slot-per-block launches, ordered second stages, integer fixed-point for the loss partials -- so that two runs of the same
// step are bit-identical.  Slower (fewer blocks on the reduction kernels); see DESIGN.md section 4.
static inline bool phx_deterministic() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PHX_DETERMINISTIC"); v = (e && atoi(e) != 0) ? 1 : 0; }
    return v == 1;
}

// dispatch on a storage dtype code
#define PHX_DT_SWITCH(dt, T, ...)                                      \
    do {                                                               \
        if ((dt) == PHX_F32) { typedef float T; __VA_ARGS__; }         \
        else if ((dt) == PHX_BF16) { typedef bf16_t T; __VA_ARGS__; }  \
        else { phx_set_error("bad dtype %d", (int)(dt)); return PHX_E_INVAL; } \
    } while (0)

static inline int phx_grid_for(size_t n, int block, int max_blocks = 4096) {
    size_t g = (n + block - 1) / block;
    if (g > (size_t)max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (int)g;
}
