// Mini-batch producer on the device: gather + random-annotator selection + augmentation (rotation, crop-scale, flips) of
// LIDC-shaped training data that lives in HBM -- what the reference does per step on the training thread with numpy / OpenCV
// (data/batch_provider.py:43-67 next_batch, 131-137 _select_random_label, 140-272 _augmentation_function; image helpers
// utils.py:18-38).  At > 4 k images/s the reference's synchronous host pipeline would be the wall.
//
// One block per output image.  The two resamplings of the reference are two PASSES with an intermediate image (rotate, then
// crop + resize: each interpolates on its own grid, they do not compose into one), the intermediate lives in LDS (128 x 128
// fp32 = 64 KiB, 192 x 192 = 144 KiB).  Arithmetic follows OpenCV's published algorithms step by step so that the result is
// reproducible against the CPU restatement (oracle/augment.py): cv2.warpAffine's 1/32-pixel fixed-point source grid with
// float32 table weights and BORDER_CONSTANT 0; cv2.resize INTER_LINEAR with float32 coefficients, horizontal then vertical
// pass; label maps are interpolated as one-hot planes in double (CV_64F) and arg-maxed (utils.py:24-38).  No FMA contraction.
#pragma clang fp contract(off)

#include "phx_common.h"

struct PhxAugParam {
    int src;            // index into the resident data set
    int annot;          // annotator whose mask is used (lidc: 0 .. 3)
    int flags;          // bit 0 rotate, 1 crop-scale, 2 fliplr, 3 flipud
    int r_y, p_x, p_y;  // crop-scale: square side and origin (batch_provider.py:216-219)
    double iM[6];       // rotation: inverse of cv2.getRotationMatrix2D((cols/2, rows/2), angle, 1), row major 2 x 3
};

namespace {

constexpr int AUG_ROT = 1, AUG_SCALE = 2, AUG_FLIPLR = 4, AUG_FLIPUD = 8;

__device__ __forceinline__ void resize_coeff(int d, int src, int dst, int* s0, int* s1, float* a0, float* a1) {
    const double scale = (double)src / (double)dst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    *s0 = s;
    *s1 = min(s + 1, src - 1);
    *a0 = 1.f - f;
    *a1 = f;
}

// fixed-point source grid of cv2.warpAffine for destination pixel (x, y): top-left source pixel and the four float32 weights
__device__ __forceinline__ void warp_taps(const PhxAugParam& p, int x, int y, int* sx, int* sy, float w[4]) {
    const long rd = 1024 / 32 / 2;
    const long adelta = (long)rint(p.iM[0] * (double)x * 1024.0), bdelta = (long)rint(p.iM[3] * (double)x * 1024.0);
    const long X0 = (long)rint((p.iM[1] * (double)y + p.iM[2]) * 1024.0) + rd;
    const long Y0 = (long)rint((p.iM[4] * (double)y + p.iM[5]) * 1024.0) + rd;
    const long XX = (X0 + adelta) >> 5, YY = (Y0 + bdelta) >> 5;
    *sx = (int)(XX >> 5);
    *sy = (int)(YY >> 5);
    const float wx1 = (float)(int)(XX & 31) * (1.0f / 32.0f), wy1 = (float)(int)(YY & 31) * (1.0f / 32.0f);
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    w[0] = wy0 * wx0; w[1] = wy0 * wx1; w[2] = wy1 * wx0; w[3] = wy1 * wx1;
}

// label of the INTERMEDIATE (rotated) map at (y, x): one-hot planes interpolated in double, arg-max (utils.py:24-27); without
// rotation the source label.  Recomputed where pass 2 needs it (four taps per output pixel) instead of being kept in LDS,
// which then holds the fp32 image alone: 192 x 192 fits.
__device__ __forceinline__ int mid_label(const PhxAugParam& p, const unsigned char* __restrict__ lbl, int X, int Y, int A, int nlabels,
                                         int y, int x) {
    if (!(p.flags & AUG_ROT)) return lbl[(size_t)(y * Y + x) * A];
    int sx, sy;
    float w[4];
    warp_taps(p, x, y, &sx, &sy, w);
    double cls[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int yy = sy + (t >> 1), xx = sx + (t & 1);
        if (yy >= 0 && yy < X && xx >= 0 && xx < Y) {
            const int l = lbl[(size_t)(yy * Y + xx) * A];
            if (l < 4) cls[l] += (double)w[t];
        }
    }
    int best = 0;
    for (int c = 1; c < nlabels; ++c)
        if (cls[c] > cls[best]) best = c;
    return best;
}

__global__ __launch_bounds__(256) void k_augment(const float* __restrict__ images, const unsigned char* __restrict__ labels,
                                                 const PhxAugParam* __restrict__ params, float* __restrict__ x_out,
                                                 unsigned char* __restrict__ s_out, int X, int Y, int A, int nlabels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* timg = reinterpret_cast<float*>(smem);                     // [X][Y] intermediate image
    const PhxAugParam p = params[blockIdx.x];
    const float* img = images + (size_t)p.src * X * Y;
    const unsigned char* lbl = labels + (size_t)p.src * X * Y * A + p.annot;      // element (y, x) at (y * Y + x) * A
    const int npix = X * Y;

    // ---- pass 1: rotation (cv2.warpAffine, INTER_LINEAR, BORDER_CONSTANT 0) or copy -> LDS
    if (p.flags & AUG_ROT) {
        for (int i = threadIdx.x; i < npix; i += blockDim.x) {
            const int y = i / Y, x = i - y * Y;
            int sx, sy;
            float w[4];
            warp_taps(p, x, y, &sx, &sy, w);
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int yy = sy + (t >> 1), xx = sx + (t & 1);
                const bool ok = yy >= 0 && yy < X && xx >= 0 && xx < Y;
                const float s = ok ? img[yy * Y + xx] : 0.f;
                v = t == 0 ? s * w[0] : v + s * w[t];
            }
            timg[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < npix; i += blockDim.x) timg[i] = img[i];
    }
    __syncthreads();

    // ---- pass 2: crop + cv2.resize INTER_LINEAR back to X x Y (or copy), flips on the way out
    float* xo = x_out + (size_t)blockIdx.x * npix;
    unsigned char* so = s_out + (size_t)blockIdx.x * npix;
    for (int i = threadIdx.x; i < npix; i += blockDim.x) {
        const int y = i / Y, x = i - y * Y;
        float v;
        int best;
        if (p.flags & AUG_SCALE) {
            int y0, y1, x0, x1;
            float b0, b1, a0, a1;
            resize_coeff(y, p.r_y, X, &y0, &y1, &b0, &b1);
            resize_coeff(x, p.r_y, Y, &x0, &x1, &a0, &a1);
            const int r0 = (p.p_y + y0) * Y + p.p_x, r1 = (p.p_y + y1) * Y + p.p_x;
            const float h0 = timg[r0 + x0] * a0 + timg[r0 + x1] * a1;
            const float h1 = timg[r1 + x0] * a0 + timg[r1 + x1] * a1;
            v = h0 * b0 + h1 * b1;
            const int l00 = mid_label(p, lbl, X, Y, A, nlabels, p.p_y + y0, p.p_x + x0), l01 = mid_label(p, lbl, X, Y, A, nlabels, p.p_y + y0, p.p_x + x1);
            const int l10 = mid_label(p, lbl, X, Y, A, nlabels, p.p_y + y1, p.p_x + x0), l11 = mid_label(p, lbl, X, Y, A, nlabels, p.p_y + y1, p.p_x + x1);
            double cls[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double g0 = (double)(l00 == c) * (double)a0 + (double)(l01 == c) * (double)a1;
                const double g1 = (double)(l10 == c) * (double)a0 + (double)(l11 == c) * (double)a1;
                cls[c] = g0 * (double)b0 + g1 * (double)b1;
            }
            best = 0;
            for (int c = 1; c < nlabels; ++c)
                if (cls[c] > cls[best]) best = c;
        } else {
            v = timg[i];
            best = mid_label(p, lbl, X, Y, A, nlabels, y, x);
        }
        const int oy = (p.flags & AUG_FLIPUD) ? X - 1 - y : y, ox = (p.flags & AUG_FLIPLR) ? Y - 1 - x : x;
        xo[oy * Y + ox] = v;
        so[oy * Y + ox] = (unsigned char)best;
    }
}

}  // namespace

extern "C" {

int phx_augment_param_bytes(void) { return (int)sizeof(PhxAugParam); }

// images [N][X][Y] f32, labels [N][X][Y][A] u8 (the HDF5 layout of data/lidc_data_loader.py:92-104), params: B records in
// DEVICE memory -> x_out [B][X][Y] (= [B,X,Y,1]) f32, s_out [B][X][Y] u8.  nlabels <= 4 (the one-hot interpolation branch of
// batch_provider.py:204-206,222-224).
int phx_augment_batch(const float* images, const unsigned char* labels, const void* params_dev, float* x_out,
                      unsigned char* s_out, int B, int X, int Y, int A, int nlabels, void* stream) {
    PHX_REQUIRE(images && labels && params_dev && x_out && s_out, PHX_E_INVAL, "augment_batch: null argument");
    PHX_REQUIRE(nlabels >= 1 && nlabels <= 4, PHX_E_SHAPE, "augment_batch: 1 <= nlabels <= 4 (one-hot interpolation)");
    PHX_REQUIRE(X > 0 && Y > 0 && A > 0 && (size_t)X * Y * 4 <= 160 * 1024, PHX_E_SHAPE,
                "augment_batch: the intermediate image has to fit LDS (X * Y <= 40960)");
    if (B <= 0) return PHX_OK;
    const size_t sh = (size_t)X * Y * 4;
    static bool attr = false;
    if (!attr) {
        PHX_CHECK_HIP(hipFuncSetAttribute((const void*)k_augment, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(k_augment, dim3(B), dim3(256), sh, (hipStream_t)stream, images, labels, (const PhxAugParam*)params_dev, x_out,
                       s_out, X, Y, A, nlabels);
    PHX_CHECK_LAUNCH();
    return PHX_OK;
}

}  // extern "C"
