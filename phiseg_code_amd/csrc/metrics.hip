// Validation metrics on the device: generalised energy distance, variance-NCC and per-label
// Dice exactly as phiseg_model._do_validation scores one image (phiseg_model.py:586-613 calling utils.py:270-370), for a
// batch of images in three launches.  The reference evaluates N*M + N^2 + M^2 pairwise IoU distances per image in Python
// loops on the host; here the label maps become bit planes (one ballot per 64 pixels and label) and every (mask, mask) pair
// is one wave of popcounts.
#include "phx_common.h"

#define MT_MAXC 8
#define MT_MAXM 8

// ---- A: per pixel -- arg-max label maps, mean soft-max arg-max, cross-entropy maps and their moments ------------------
// acc[i][0] = sum a, [1] = sum a^2, [2 + 3 j ..] = sum v_j, sum v_j^2, sum a v_j   (a = E_ss map, v_j = E_sy[j] map)
// planes[img][mask][c][word] : bit p % 64 of word p / 64 = (mask label at pixel p == c); masks 0..N-1 samples, N mean arg-max,
// N+1..N+M annotations, N+M+1 sref -- the pair kernel works on these (1 bit per pixel and label instead of a byte per pixel)
__global__ void k_metrics_pixel(const float* __restrict__ sm, const unsigned char* __restrict__ gt,
                                const unsigned char* __restrict__ sref, unsigned long long* __restrict__ planes,
                                double* __restrict__ acc, int N, int M, int P, int C) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int NMK = N + M + 2, W64 = (P + 63) / 64, word = p >> 6;
    unsigned long long* pl = planes + (size_t)img * NMK * MT_MAXC * W64;
    auto put_planes = [&](int mask, int label) {         // whole wave: one ballot per label (label < 0: pixel beyond P)
#pragma unroll
        for (int c = 0; c < MT_MAXC; ++c)
            if (c < C) {
                const unsigned long long bits = __ballot(label == c);
                if ((threadIdx.x & 63) == 0 && (blockIdx.x * blockDim.x + (threadIdx.x & ~63)) < P)
                    pl[((size_t)mask * MT_MAXC + c) * W64 + word] = bits;
            }
    };
    float vals[2 + 3 * MT_MAXM];
#pragma unroll
    for (int k = 0; k < 2 + 3 * MT_MAXM; ++k) vals[k] = 0.f;
    const bool in = p < P;
    float mean[MT_MAXC], slog[MT_MAXC];
#pragma unroll
    for (int c = 0; c < MT_MAXC; ++c) mean[c] = slog[c] = 0.f;
    // four samples per trip, loads first (a thread's trips are serially dependent: the loads in flight set the speed)
    for (int n0 = 0; n0 < N; n0 += 4) {
        float v[4][MT_MAXC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < MT_MAXC; ++c)
                v[u][c] = (in && n0 + u < N && c < C) ? sm[(((size_t)img * N + n0 + u) * P + p) * C + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (n0 + u >= N) break;                      // uniform
            int best = -1;
            if (in) {
                best = 0;
                float bv = v[u][0];
#pragma unroll
                for (int c = 0; c < MT_MAXC; ++c)
                    if (c < C) {
                        mean[c] += v[u][c];
                        slog[c] += logf(v[u][c] + 1e-8f);
                        if (v[u][c] > bv) { bv = v[u][c]; best = c; }   // first maximum wins, like np.argmax
                    }
            }
            put_planes(n0 + u, best);
        }
    }
    int bmi = -1;
    if (in) {
        const float invn = 1.f / (float)N;
        float ess = 0.f, bm = -1.f;
        bmi = 0;
#pragma unroll
        for (int c = 0; c < MT_MAXC; ++c)
            if (c < C) {
                const float mc = mean[c] * invn;
                ess -= mc * slog[c];
                if (mc > bm) { bm = mc; bmi = c; }
            }
        ess *= invn;
        vals[0] = ess;
        vals[1] = ess * ess;
#pragma unroll
        for (int j = 0; j < MT_MAXM; ++j)
            if (j < M) {
                const int g = gt[((size_t)img * M + j) * P + p];
                float sl = 0.f;
#pragma unroll
                for (int c = 0; c < MT_MAXC; ++c) sl = (c == g) ? slog[c] : sl;
                const float v = -sl * invn;
                vals[2 + 3 * j] = v;
                vals[3 + 3 * j] = v * v;
                vals[4 + 3 * j] = ess * v;
            }
    }
    put_planes(N, bmi);
    for (int j = 0; j < M; ++j) put_planes(N + 1 + j, in ? (int)gt[((size_t)img * M + j) * P + p] : -1);
    put_planes(N + M + 1, in ? (int)sref[(size_t)img * P + p] : -1);
    __shared__ double red[4][2 + 3 * MT_MAXM];
    const int nv = 2 + 3 * M;
    for (int k = 0; k < nv; ++k) {
        double d = (double)vals[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = d;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double d = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) d += red[w][threadIdx.x];
        atomicAdd(&acc[(size_t)img * (2 + 3 * MT_MAXM) + threadIdx.x], d);
    }
}

// ---- B: one wave per (mask a, mask b) pair -- per-label counts and intersections by popcount over the bit planes ---------
// pair p < npairs: masks a < b among {N samples, M annotations}; pair npairs: (arg-max of the mean soft-max, sref) for the Dice
// stats[i][pair][c][3] = {|a == c|, |b == c|, |a == c and b == c|}
__global__ void k_metrics_pairs(const unsigned long long* __restrict__ planes, int* __restrict__ stats, int N, int M, int P,
                                int C, int label0) {
    const int img = blockIdx.y, K = N + M, npairs = K * (K - 1) / 2, NMK = N + M + 2, W64 = (P + 63) / 64;
    const int pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair > npairs) return;
    int ia, ib;
    if (pair < npairs) {
        int idx = pair, a = 0;
        while (idx >= K - 1 - a) { idx -= K - 1 - a; ++a; }
        const int b = a + 1 + idx;
        ia = a < N ? a : a + 1;                       // plane index: annotations sit behind the mean-arg-max plane
        ib = b < N ? b : b + 1;
    } else {
        ia = N;
        ib = N + M + 1;
    }
    const unsigned long long* pa = planes + ((size_t)img * NMK + ia) * MT_MAXC * W64;
    const unsigned long long* pb = planes + ((size_t)img * NMK + ib) * MT_MAXC * W64;
    for (int c = (pair < npairs ? label0 : 0); c < C; ++c) {   // the GED only looks at labels label0 .. C-1
        int x = 0, y = 0, z = 0;
        for (int w = lane; w < W64; w += 64) {
            const unsigned long long ua = pa[(size_t)c * W64 + w], ub = pb[(size_t)c * W64 + w];
            x += __popcll(ua);
            y += __popcll(ub);
            z += __popcll(ua & ub);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            x += __shfl_xor(x, o, 64);
            y += __shfl_xor(y, o, 64);
            z += __shfl_xor(z, o, 64);
        }
        if (lane == 0) {
            int* st = stats + (((size_t)img * (npairs + 1) + pair) * MT_MAXC + c) * 3;
            st[0] = x; st[1] = y; st[2] = z;
        }
    }
}

// ---- C: one block per image -- distances, GED, NCC, Dice --------------------------------------------------------------------
__global__ void k_metrics_final(const int* __restrict__ stats, const double* __restrict__ acc, int N, int M, int P, int C,
                                int label0, float* __restrict__ out) {
    const int img = blockIdx.x, K = N + M, npairs = K * (K - 1) / 2;
    double sy = 0.0, ss = 0.0, yy = 0.0;
    for (int pr = threadIdx.x; pr < npairs; pr += blockDim.x) {
        int idx = pr, a = 0;
        while (idx >= K - 1 - a) { idx -= K - 1 - a; ++a; }
        const int b = a + 1 + idx;
        const int* st = stats + ((size_t)img * (npairs + 1) + pr) * MT_MAXC * 3;
        double iou = 0.0;
        for (int c = label0; c < C; ++c) {
            const int na = st[c * 3], nb = st[c * 3 + 1], ni = st[c * 3 + 2];
            if (na == 0 && nb == 0) iou += 1.0;
            else if (na != 0 && nb != 0) iou += (double)ni / (double)(na + nb - ni);
        }
        const double d = 1.0 - iou / (double)(C - label0);
        if (b < N) ss += d; else if (a >= N) yy += d; else sy += d;
    }
    __shared__ double red[3][64];
    red[0][threadIdx.x] = sy; red[1][threadIdx.x] = ss; red[2][threadIdx.x] = yy;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int q = 0; q < (int)blockDim.x; ++q) { t0 += red[0][q]; t1 += red[1][q]; t2 += red[2][q]; }
        // the reference sums over ordered pairs including i == j (distance 0): ordered sums = 2 x unordered sums
        const double ged = 2.0 / ((double)N * M) * t0 - 2.0 * t1 / ((double)N * N) - 2.0 * t2 / ((double)M * M);
        float* o = out + (size_t)img * (2 + MT_MAXC);
        o[0] = (float)ged;
        const double* ac = acc + (size_t)img * (2 + 3 * MT_MAXM);
        const double n = (double)P, ma = ac[0] / n, va = ac[1] / n - ma * ma;
        double ncc = 0.0;
        for (int j = 0; j < M; ++j) {
            const double mv = ac[2 + 3 * j] / n, vv = ac[3 + 3 * j] / n - mv * mv, cav = ac[4 + 3 * j] / n - ma * mv;
            ncc += cav / (sqrt(va) * sqrt(vv));
        }
        o[1] = (float)(ncc / M);
        const int* sd = stats + ((size_t)img * (npairs + 1) + npairs) * MT_MAXC * 3;
        for (int c = 0; c < C; ++c) {
            const int na = sd[c * 3], nb = sd[c * 3 + 1], ni = sd[c * 3 + 2];
            o[2 + c] = (na == 0 && nb == 0) ? 1.f : ((na == 0 || nb == 0) ? 0.f : (float)(2.0 * ni / (double)(na + nb)));
        }
    }
}

static size_t mt_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" {

size_t phx_validation_metrics_ws_bytes(int I, int N, int M, int P, int C) {
    (void)C;
    const size_t K = (size_t)N + M, npairs = K * (K - 1) / 2;
    return mt_align((size_t)I * (N + M + 2) * MT_MAXC * ((P + 63) / 64) * 8) + mt_align((size_t)I * (2 + 3 * MT_MAXM) * sizeof(double)) +
           mt_align((size_t)I * (npairs + 1) * MT_MAXC * 3 * sizeof(int));
}

int phx_validation_metrics(const float* sm, const unsigned char* gt, const unsigned char* sref, void* work, size_t work_bytes,
                           int I, int N, int M, int P, int C, int label0, float* out, void* stream) {
    PHX_REQUIRE(I > 0 && N > 0 && M > 0 && P > 0, PHX_E_SHAPE, "validation_metrics: empty input");
    PHX_REQUIRE(C >= 2 && C <= MT_MAXC && M <= MT_MAXM && label0 >= 0 && label0 < C, PHX_E_SHAPE,
                "validation_metrics: 2 <= C <= 8, M <= 8, 0 <= label0 < C");
    PHX_REQUIRE(work_bytes >= phx_validation_metrics_ws_bytes(I, N, M, P, C), PHX_E_INVAL, "validation_metrics: workspace too small");
    const size_t K = (size_t)N + M, npairs = K * (K - 1) / 2;
    unsigned long long* planes = (unsigned long long*)work;
    double* acc = (double*)((char*)work + mt_align((size_t)I * (N + M + 2) * MT_MAXC * ((P + 63) / 64) * 8));
    int* stats = (int*)((char*)acc + mt_align((size_t)I * (2 + 3 * MT_MAXM) * sizeof(double)));
    PHX_CHECK_HIP(hipMemsetAsync(acc, 0, (size_t)I * (2 + 3 * MT_MAXM) * sizeof(double), (hipStream_t)stream));
    hipLaunchKernelGGL(k_metrics_pixel, dim3((P + 255) / 256, I), dim3(256), 0, (hipStream_t)stream, sm, gt, sref, planes, acc, N,
                       M, P, C);
    PHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_metrics_pairs, dim3((unsigned)(npairs + 1 + 3) / 4, I), dim3(256), 0, (hipStream_t)stream, planes, stats,
                       N, M, P, C, label0);
    PHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_metrics_final, dim3(I), dim3(64), 0, (hipStream_t)stream, stats, acc, N, M, P, C, label0, out);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
