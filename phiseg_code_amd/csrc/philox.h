// Philox4x32-10 + Box-Muller on the device; contract identical to oracle/philox.py.
#pragma once
#include "phx_common.h"

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0;
        unsigned n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 4 standard normals for (block, sample, stream, step)
__device__ __forceinline__ void philox_normal4(unsigned block, unsigned sample, unsigned stream, unsigned step,
                                               unsigned long long seed, float n[4]) {
    unsigned x[4];
    philox4x32_10(block, sample, stream, step, (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), x);
    const float two24 = 1.0f / 16777216.0f;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        float u1 = ((float)(x[j] >> 8) + 1.0f) * two24;
        float u2 = (float)(x[j + 1] >> 8) * two24;
        float r = sqrtf(-2.0f * logf(u1));
        float s, c;
        sincospif(2.0f * u2, &s, &c);
        n[j] = r * c;
        n[j + 1] = r * s;
    }
}
