/* phx.h -- C ABI of libphx.so: the MI355X (gfx950) kernels of the PHiSeg ELBO hot path.
 *
 * Drop-in boundary.  The reference (baumgach/PHiSeg-code) has no FFI: its
 * arithmetic sits behind TensorFlow 1.12's Python API.  Each entry point below therefore replaces the
 * TF op(s) invoked at the cited reference call site; the Python host in phiseg_code_amd/ (same function
 * names / arguments as the reference's tfwrapper.layers etc.) is the only caller.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PHX_E_* otherwise; phx_last_error() gives text;
 *   - the caller owns every device buffer (plain pointers + sizes, no hidden allocation, no torch types);
 *   - every launch takes the hipStream_t (passed as void*) it is enqueued on; nothing synchronises;
 *   - activations are NHWC contiguous; conv weights are TF's HWIO fp32 ([kh][kw][Cin][Cout]);
 *   - dtype codes: PHX_F32 = 0, PHX_BF16 = 1 (storage type; arithmetic is always fp32-accumulate);
 *   - act codes: 0 identity, 1 relu, 2 softplus.
 */
#ifndef PHX_H
#define PHX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PHX_F32 = 0, PHX_BF16 = 1 };
enum { PHX_ACT_ID = 0, PHX_ACT_RELU = 1, PHX_ACT_SOFTPLUS = 2 };
enum { PHX_OK = 0, PHX_E_INVAL = -1, PHX_E_SHAPE = -2, PHX_E_ALIGN = -3, PHX_E_LAUNCH = -4, PHX_E_RUNTIME = -5, PHX_E_COMM = -6 };

/* ---- runtime plumbing ------------------------------------------------------------------------------ */
int phx_abi_version(void);
int phx_last_error(char* buf, size_t n);
/* CRC-32C (Castagnoli) of a HOST buffer; *crc holds the running value (start with 0).  Host-side helper of the TensorFlow
 * checkpoint reader / writer (tfwrapper/tf_checkpoint.py: what tf.train.Saver's tensor bundles carry per block and tensor). */
int phx_crc32c(const void* data, size_t n, unsigned* crc);
int phx_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* name, size_t name_n);
int phx_stream_create(void** stream);
int phx_stream_destroy(void* stream);
int phx_stream_sync(void* stream);
int phx_event_create(void** ev);
int phx_event_destroy(void* ev);
int phx_event_record(void* ev, void* stream);
int phx_event_sync(void* ev);
int phx_event_elapsed_ms(void* start, void* stop, float* ms);
int phx_stream_wait_event(void* stream, void* ev);
/* hipGraph capture of a launch sequence (replaces tf.Session.run's per-step dispatch, phiseg_model.py:194) */
int phx_graph_begin_capture(void* stream);
int phx_graph_end_capture(void* stream, void** graph_exec);
int phx_graph_launch(void* graph_exec, void* stream);
int phx_graph_destroy(void* graph_exec);
int phx_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int phx_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int phx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int phx_memset(void* dst, int value, size_t bytes, void* stream);

/* ---- convolution: tf.nn.conv2d(x, W, [1,1,1,1], 'SAME') + bias_add + activation -------------------- */
/* tfwrapper/layers.py:122-135.  Generic fp32-math kernel: any odd k (1 or 3), any Cin / Cout, x and y in
 * f32 or bf16.  transpose_flip = 1 evaluates the data-gradient instead (x := dy [.., Cout], y := dx [.., Cin],
 * filter flipped + in/out swapped; `Cin`/`Cout` are always those of W).  stats (nullable) += per-channel
 * {sum, sum of squares} of the stored output, [channels][2] fp32 (= sums[1][C][2] of phx_norm_finalize). */
int phx_conv2d_direct(const void* x, int x_dt, const float* w_hwio, const float* bias, void* y, int y_dt,
                      int B, int H, int W, int Cin, int Cout, int ksize, int act, int transpose_flip,
                      float* stats, void* stream);
/* filter / bias gradient of the same op (what optimizer.minimize derives, phiseg_model.py:141):
 * dw[kh][kw][ci][co] += sum_{b,y,x} x[b,y+kh-p,x+kw-p,ci] * dy[b,y,x,co];  dbias[co] += sum dy. */
int phx_conv2d_direct_wgrad(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwio, float* dbias,
                            int B, int H, int W, int Cin, int Cout, int ksize, void* stream);
/* The same gradient with a FIXED summation order at full parallelism (what the engine launches under PHX_DETERMINISTIC=1, where
 * phx_conv2d_direct_wgrad falls back to one block per channel block): up to 64 pixel slices leave partial filters in `workspace`
 * (phx_conv2d_direct_wgrad_ordered_ws_bytes; 0 = one slice, no workspace needed) and a second launch adds them in slice order. */
size_t phx_conv2d_direct_wgrad_ordered_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize);
int phx_conv2d_direct_wgrad_ordered(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwio, float* dbias,
                                    void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int ksize,
                                    void* stream);

/* fp32 MFMA path (v_mfma_f32_32x32x2_f32: an fp32 fused-multiply-add chain on the matrix cores, the arithmetic class of
 * phx_conv2d_direct -- only the summation order differs), 3x3 only, f32 tensors, N % 32 == 0, any K.  What the fp32 parity plan
 * runs for tf.nn.conv2d (tfwrapper/layers.py:123) and the two gradients optimizer.minimize derives (phiseg_model.py:141).
 * wpk is the packed fp32 filter [K8 / 8][9][N][8], K8 = K rounded up to a multiple of 8, zero-filled (element (tap t, row n,
 * channel k) at (((k / 8) * 9 + t) * N + n) * 8 + k % 8), phx_conv3x3_f32_mfma_packed_floats(K, N) floats:
 *   forward:  N = Cout, K = Cin, tap t = kh*3+kw;   data gradient: N = Cin, K = Cout, t = (2-kh)*3+(2-kw), call with x := dy.
 * phx_pack_conv3x3_f32_multi: every filter of a step in one launch; descs_dev = device array of n records
 * { const float* w_hwio; float* wpk_fwd; float* wpk_dgrad (nullable); int32 Cin, Cout }  (32 bytes each). */
int phx_conv3x3_f32_mfma_supported(int B, int H, int W, int K, int N);
size_t phx_conv3x3_f32_mfma_packed_floats(int K, int N);
int phx_pack_conv3x3_f32_multi(const void* descs_dev, int n, void* stream);
int phx_conv3x3_f32_mfma(const float* x, const float* wpk, const float* bias, float* y, int B, int H, int W, int K, int N,
                         int act, void* stream);
/* Filter / bias gradient of the same op (dw += ..., dbias += ...; dbias nullable), Cout % 32 == 0, any Cin: persistent blocks leave
 * partial filters in `workspace` (phx_conv3x3_f32_mfma_wgrad_ws_bytes) and a second launch adds them in slice order -- a fixed
 * summation order in every mode. */
int phx_conv3x3_f32_mfma_wgrad_supported(int B, int H, int W, int Cin, int Cout);
size_t phx_conv3x3_f32_mfma_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int with_bias);
int phx_conv3x3_f32_mfma_wgrad(const float* x, const float* dy, float* dw_hwio, float* dbias, void* workspace,
                               size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream);

/* ---- bf16 MFMA path (v_mfma_f32_32x32x16_bf16): 3x3 SAME convolution, forward and data gradient -------------------------------
 * tf.nn.conv2d + bias_add + activation of tfwrapper/layers.py:122-135 and what optimizer.minimize derives for its input
 * (phiseg_model.py:141), Cin % 32 == 0, Cout % 32 == 0, bf16 NHWC tensors.
 * wpk is the packed bf16 filter [K / 32][9][N][32] (element (tap t, row n, channel k) at (((k / 32) * 9 + t) * N + n) * 32
 * + k % 32: the slab of one 32-channel chunk is contiguous) written by phx_pack_conv3x3_bf16:
 *   forward:  N = Cout, K = Cin, tap t = kh*3+kw          (wpk_fwd)
 *   dgrad:    N = Cin,  K = Cout, tap t = (2-kh)*3+(2-kw) (wpk_dgrad); launch with x := dy, K := Cout, N := Cin. */
int phx_pack_conv3x3_bf16(const float* w_hwio, void* wpk_fwd, void* wpk_dgrad, int Cin, int Cout, void* stream);
/* Narrow-input convolutions (image inputs Cin = 1 / 3, posteriors.py:87, priors.py:80; latent inputs Cin = zdim0 = 2,
 * likelihoods.py:197, posteriors.py:115): zero-pad the channel axis to Cin_pad = 32 so they run on the MFMA kernels: padded bf16 copy
 * of the input, padded packed filter, and the filter gradient of the padded problem folded back into dw_hwio[9][Cin][Cout]. */
int phx_pack_conv3x3_bf16_pad(const float* w_hwio, void* wpk_fwd, void* wpk_dgrad /* nullable */, int Cin, int Cin_pad,
                              int Cout, void* stream);
/* every filter of a step in one launch: descs_dev = device array of n records
 * { const float* w_hwio; void* wpk_fwd; void* wpk_dgrad (nullable); int32 Cin, Cin_pad, Cout, k1 }  (40 bytes each) */
int phx_pack_conv3x3_bf16_multi(const void* descs_dev, int n, void* stream);
int phx_pad_channels_bf16(const void* x, int dt, int C, void* out, int Cpad, size_t npix, void* stream);
int phx_unpad_channels_bf16(const void* src, void* dst, int dst_dt, int C, int Cpad, size_t npix, void* stream);
int phx_unpad_filter_grad_accumulate(const float* dw_pad, float* dw_hwio, int Cin, int Cin_pad, int Cout, void* stream);
/* 1x1 convolutions run as the centre tap of a 3x3 (descriptor field k1 of phx_pack_conv3x3_bf16_multi):
 * dw_1x1[ci][co] += dw_pad[tap 4][ci][co] */
int phx_unpad_filter_grad_center(const float* dw_pad, float* dw_1x1, int Cin, int Cin_pad, int Cout, void* stream);

/* ONE launch entry for the whole family (round 6: it replaces the eleven phx_conv3x3_mfma_bf16* entry points and their *_supported
 * probes of rounds 1-5).  Prologue, epilogue and statistics mode are fields of the descriptor; a field left 0 / NULL is off.
 *   input      x [B,H,W,K]; x2 != NULL: concat-free input (tf.concat([a, b], 3) -> conv2D: posteriors.py:87,120, priors.py:112,
 *              likelihoods.py:210) -- reduction channels [0, K1) from x (pixel stride K1), [K1, K) from x2 (stride K - K1), K1 % 32 == 0
 *   prologue   xscale / xshift != NULL: x is the PRE-normalisation tensor of the layer in front and the loader re-forms
 *              relu(x * xscale[k] + xshift[k]) (conv2d -> batch_norm -> relu -> conv2d without the activation tensor; plan.xf_ok:
 *              the 32 -> 32 layers of the large maps, HBM-bound, where the transform is free and the bytes are not); bit-identical
 *              to the launch on the materialised activation
 *   output     y [B,H,W,N] bf16; y2 != NULL: output channels [0, N1) to y (stride N1), [N1, N) to y2 (stride N - N1), N1 % 8 == 0
 *              (the data gradient of a concat-free layer); y_f32 != NULL (plan.f32out_ok, small maps): the fp32 accumulators reach
 *              y_f32[B*H*W][N] without a bf16 rounding, no epilogue; sum_slices = 0 then leaves the split-K slices ws[z][B*H*W][N] in
 *              the workspace for a consumer that sums them itself (phx_bn_wide_fwd) and y_f32 is not written
 *   epilogue   y = act(acc * oscale[n] + bias[n]) (oscale NULL: 1, bias NULL: 0) -- bias_add + activation, or inference-mode
 *              batch norm + activation folded into the convolution (normalisation.py:145-163 with is_training = False; scale / shift
 *              from phx_bn_infer_scale_shift(_multi))
 *   statistics stats_mode PHX_CONV_STATS_PARTIAL: per-channel {sum, sumsq} of the bf16-rounded output as rows stats[plan.tiles][2][N]
 *              (no atomics; plan.tiles_dual rows for a launch with x2 / y2 set); PHX_CONV_STATS_ATOMIC (plan.stats_atomic_ok: at most 64
 *              pixel tiles, never in deterministic mode): added into stats[N][2], the layout phx_norm_apply_fused takes
 *   split-K    workspace of plan.ws_bytes (0: not used for this shape): the K / 32 chunks are spread over plan.ksplit blocks per tile
 *              and a finishing kernel sums the fp32 slices and applies the epilogue (small maps; not with PARTIAL statistics)
 *   fused group / instance norm (gn_groups > 0, plan.fgn_block != 0: H, W in {2, 4, 8, 16}; gn_groups == N: instance norm): convolution
 *              + bias + normalisation + activation in ONE launch (layers.py:123-135 + normalisation.py:3-36) -- a block holds whole
 *              samples and whole groups; writes y = bf16(conv + bias), a_out = act((y - mean) * rstd * gamma + beta) and the per-sample
 *              mean_out / rstd_out [B][G], scale_out / shift_out [B][N] in phx_norm_small_bwd's layout. */
enum { PHX_CONV_STATS_NONE = 0, PHX_CONV_STATS_PARTIAL = 1, PHX_CONV_STATS_ATOMIC = 2 };
typedef struct phx_conv3x3_desc {
    const void* x;
    const void* x2;
    const float* xscale;
    const float* xshift;
    const void* wpk;
    void* y;
    void* y2;
    float* y_f32;
    const float* bias;
    const float* oscale;
    float* stats;
    void* workspace;
    size_t workspace_bytes;
    void* a_out;
    const float* gamma;
    const float* beta;
    float* mean_out;
    float* rstd_out;
    float* scale_out;
    float* shift_out;
    int K1, N1, act, stats_mode, sum_slices, gn_groups;
    float gn_eps;
    int B, H, W, K, N;
    int reserved[4];
} phx_conv3x3_desc;                 /* 224 bytes */
typedef struct phx_conv3x3_plan {
    int tiles, tiles_dual;          /* rows of a PARTIAL statistics buffer: plain launch / launch with x2 or y2 */
    int ksplit;                     /* fp32 slices a workspace launch leaves (1: no split) */
    int stats_atomic_ok, f32out_ok, xf_ok;
    int fgn_block;                  /* fused group norm with G groups: output channels per block (32 / 64), 0 = not supported */
    int reserved;
    size_t ws_bytes;
} phx_conv3x3_plan;                 /* 40 bytes */
int phx_conv3x3_desc_bytes(void);
int phx_conv3x3_bf16_plan(int B, int H, int W, int K, int N, int G, phx_conv3x3_plan* out);
int phx_conv3x3_bf16(const phx_conv3x3_desc* d, void* stream);

/* Filter gradients of the two input forms above.  _partial_xf: x is the pre-normalisation tensor (prologue as above; Cin == 32, 16 x 16
 * tiles, more than 1 024 of them, workspace of phx_conv3x3_wgrad_ws_bytes required; partial filters only, as
 * phx_conv3x3_wgrad_mfma_bf16_partial).  _dual: the second tensor explicitly (K1 % 32 == 0; a block's input channels lie in one tensor,
 * so K1 % 64 != 0 selects 32-channel tiles: use the ..._dual plan / workspace queries with the same K1). */
int phx_conv3x3_wgrad_xf_supported(int B, int H, int W, int Cin, int Cout);
int phx_conv3x3_wgrad_mfma_bf16_partial_xf(const void* x, const float* xscale, const float* xshift, const void* dy, float* dw_hwio,
                                           void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream);
size_t phx_conv3x3_wgrad_ws_bytes_dual(int B, int H, int W, int Cin, int Cout, int K1);
int phx_conv3x3_wgrad_reduce_plan_dual(int B, int H, int W, int Cin, int Cout, int K1, int* plan6);
int phx_conv3x3_wgrad_multi_job_dual(const void* x, const void* x2, int K1, const void* dy, float* dw_hwio, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int blocks_target, int blk0,
                                     void* job_out, int* info9);
int phx_conv3x3_wgrad_mfma_bf16_dual(const void* x, const void* x2, int K1, const void* dy, float* dw_hwio, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int reduce, void* stream);
/* debug: device buffer of >= 16 uint64 that receives shader-clock phase timestamps of block 0 (NULL disables) */
int phx_debug_set_trace(void* dev_buf);
/* debug: device buffer of 4 uint64 per block {start, end, HW_ID | XCC_ID << 32, realtime} written by the MFMA conv kernels */
int phx_debug_set_blocklog(void* dev_buf);
/* (the kernel-selection policy is a constant of libphx.so; the test build libphx_dbg.so can set it: include/phx_debug.h) */
/* dw_hwio[kh][kw][ci][co] += sum x * dy, Cin % 32 == 0, Cout % 32 == 0.  With a workspace (>= phx_conv3x3_wgrad_ws_bytes)
 * the per-block partial filters are stored with plain writes and summed by a second kernel; workspace == NULL falls back
 * to fp32 atomics straight into dw_hwio (a CU issues those at ~1 lane/clock: 46 us per block on MI355X). */
size_t phx_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout);
/* Deferred reduction: ..._partial launches the filter-gradient kernels only and leaves the partial filters in the workspace;
 * phx_conv3x3_wgrad_reduce_plan tells (host side) whether that launch uses the workspace at all (plan6[0]; maps of a few tiles
 * add straight into dw_hwio) and the slice / tile geometry; phx_wgrad_reduce_multi then sums the workspaces of many layers in
 * ONE launch.  jobs_dev: device array of {const float* ws; float* dw; int nslice, Cin, Cout, tci, tco, gx, gy, blk0;} with
 * gx, gy = plan6[4], plan6[5], blk0 = running sum of gx * gy (ascending), total_blocks = its final value. */
int phx_conv3x3_wgrad_reduce_plan(int B, int H, int W, int Cin, int Cout, int* plan6);
int phx_conv3x3_wgrad_mfma_bf16_partial(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes,
                                        int B, int H, int W, int Cin, int Cout, void* stream);
int phx_wgrad_reduce_multi(const void* jobs_dev, int njobs, int total_blocks, void* stream);
/* Deferred small-map filter gradients (maps narrower than 16 pixels: a few tiles, 9-36 blocks, latency-bound, and leaves of
 * the backward graph): phx_conv3x3_wgrad_multi_job fills ONE job record of phx_conv3x3_wgrad_multi_job_bytes() bytes in HOST
 * memory for the launch phx_conv3x3_wgrad_mfma_bf16_partial would make; info9 = {variant (1-8 register-staged small-map
 * kernels, 9-12 LDS-DMA kernels on 16x16 tiles with at most PHX_WGRAD_DEFER_TILES = 1024 tiles; 0: not deferred -- use the
 * per-layer launch), blocks, dynamic LDS bytes, uses_workspace, nslice, tci, tco, reduce grid x, reduce grid y} (9 ints;
 * the last five describe this launch's phx_wgrad_reduce_multi job).  blocks_target > 0 overrides the pixel-tile split of this job:
 * inside a multi-layer launch a layer needs far fewer partial filters than alone.  The caller concatenates the records of one variant (blk0 = running
 * sum of blocks), copies them to the device and calls phx_conv3x3_wgrad_multi once (lds_bytes = max over the jobs). */
int phx_conv3x3_wgrad_multi_job_bytes(void);
int phx_conv3x3_wgrad_multi_job(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes, int B,
                                int H, int W, int Cin, int Cout, int blocks_target, int blk0, void* job_out, int* info9);
int phx_conv3x3_wgrad_multi(const void* jobs_dev, int njobs, int total_blocks, int variant, size_t lds_bytes, void* stream);
int phx_conv3x3_wgrad_mfma_bf16(const void* x, const void* dy, float* dw_hwio, void* workspace, size_t workspace_bytes,
                                int B, int H, int W, int Cin, int Cout, void* stream);

/* 1x1 "head" convolutions with nout in {2,4,6,8} outputs (mu / sigma / y_lvl / pre_mu / prediction heads:
 * posteriors.py:125-127, priors.py:117-119, likelihoods.py:155,220): streaming kernels, fp32 outputs.
 * w = HWIO 1x1 filter [C][nout];  fwd: y = act(x.w + b);  dgrad: dx = dy.w^T (written);  wgrad: dw += x^T.dy, db += sum dy */
int phx_head1x1_fwd(const void* x, int x_dt, const float* w, const float* bias, float* y, size_t npix, int C, int nout,
                    int act, void* stream);
/* The two 1x1 latent heads of a level and the sample, fused (posteriors.py:125-128, priors.py:117-120):
 *   mu = x w_mu + b_mu;   sigma = softplus(x w_sigma + b_sigma);   z = mu + sigma * eps   (z may be NULL: heads only)
 * x [npix][C] (bf16 / fp32), w_* the HWIO 1x1 filters [C][zdim] fp32, mu / sigma / z [npix][zdim] fp32; eps = the Philox normal
 * phx_reparam_fwd draws for element (pixel-in-sample * zdim + channel) of sample (pixel / pix_per_sample + sample_offset).
 * _bwd: upstream gradients dz (NULL: none), dmu, dsigma (NULL: none) -> g_mu = dz + dmu, g_sigma = (dz * eps + dsigma) *
 * softplus'(pre-activation) (both [npix][zdim]: the dy operands of phx_head1x1_wgrad for the two heads) and
 * dx = g_mu w_mu^T + g_sigma w_sigma^T, written once. */
int phx_latent_heads_fwd(const void* x, int x_dt, const float* w_mu, const float* b_mu, const float* w_sigma, const float* b_sigma,
                         float* mu, float* sigma, float* z, size_t npix, int C, int zdim, int pix_per_sample, uint64_t seed,
                         const int32_t* step_dev, int stream_id, int sample_offset, void* stream);
int phx_latent_heads_bwd(const float* dz, const float* dmu, const float* dsigma, const float* sigma, const float* w_mu,
                         const float* w_sigma, void* dx, int dx_dt, float* g_mu, float* g_sigma, size_t npix, int C, int zdim,
                         int pix_per_sample, uint64_t seed, const int32_t* step_dev, int stream_id, int sample_offset, void* stream);
int phx_head1x1_dgrad(const float* dy, const float* w, void* dx, int dx_dt, size_t npix, int C, int nout, void* stream);
int phx_head1x1_wgrad(const void* x, int x_dt, const float* dy, float* dw, float* db, size_t npix, int C, int nout,
                      void* stream);
/* Deferred form: the head filter gradients are leaves of the backward graph, so all heads of a plan can share ONE launch.
 * phx_head1x1_wgrad_plan (host) gives plan4 = {PL, chunk, grid, dynamic LDS bytes} for C % 8 == 0; jobs_dev is a device array
 * of {const void* x; const float* dy; float* dw; float* db; uint64 npix; int C, PL, chunk, blk0; const float* xscale; const float*
 * xshift; int xact, pad;} (80 bytes) with blk0 = running sum of the grids (ascending), all jobs with the same x dtype and nout;
 * lds_bytes = max of the jobs' plan4[3].  xscale != NULL (round 5): x is the PRE-normalisation tensor of the layer whose only reader
 * is this head and a = act(x * xscale[c] + xshift[c]) is re-formed on load, rounded to bf16 as the stored tensor would have been --
 * the training plan then never writes a (phx_norm_apply_fused_head with y == NULL; batch norm: one scale / shift per channel). */
int phx_head1x1_wgrad_plan(size_t npix, int C, int nout, int* plan4);
int phx_head1x1_wgrad_multi(const void* jobs_dev, int njobs, int total_blocks, int x_dt, int nout, size_t lds_bytes,
                            void* stream);

/* ---- normalisation (tfwrapper/normalisation.py:3-36,145-163) --------------------------------------- */
/* One implementation for batch / group / instance norm.  A statistic is taken over P pixels x (C/G) channels
 * for each of NS sample-groups and G channel-groups:  batch norm: NS=1, P=B*H*W, G=C;  group norm: NS=B,
 * P=H*W, G=groups;  instance norm: NS=B, P=H*W, G=C.
 *   sums[NS][C][2]   per-channel {sum x, sum x^2}            (phx_norm_stats accumulates; zero it first)
 *   mean/rstd[NS][G], scale/shift[NS][C]: y = act(x*scale+shift), scale = gamma*rstd, shift = beta-mean*scale */
/* pivot (nullable) [NS][C]: when given, the kernel writes pivot = x[ns][pixel 0][c] and accumulates the sums of
 * (x - pivot), (x - pivot)^2 instead -- no catastrophic cancellation when var << mean^2 (few samples). */
int phx_norm_stats(const void* x, int dt, float* sums, float* pivot, int NS, int P, int C, void* stream);
/* partial[T][2][C] (conv epilogue rows) -> sums[1][C][2] */
int phx_norm_reduce_partials(const float* partial, int T, int C, float* sums, void* stream);
/* per-sample form for group / instance norm: the convolution's per-tile partial sums partial[ns * T + t][2][C] (T pixel tiles per
 * sample, tiles of one sample contiguous: maps of at least 16 x 16) -> sums[ns][c][2] (overwritten) */
int phx_norm_reduce_partials_ns(const float* partial, int T, int NS, int C, float* sums, void* stream);
int phx_norm_finalize(const float* sums, const float* pivot, const float* gamma, const float* beta, float eps, int NS,
                      int P, int C, int G, float* mean, float* rstd, float* scale, float* shift,
                      float* moving_mean, float* moving_var, float momentum /* 0 => no moving update */,
                      void* stream);
/* inference-mode batch norm: scale/shift from the moving statistics */
int phx_bn_infer_scale_shift(const float* gamma, const float* beta, const float* moving_mean,
                             const float* moving_var, float eps, int C, float* scale, float* shift, void* stream);
/* all inference-mode batch-norm layers of a plan in ONE launch: descs_dev = n records {const float* gamma, beta, moving_mean,
 * moving_var; float* scale, shift; int C; float eps} (48 bytes, device memory) */
/* out[b * n + k] = x[b] (k < n): a feature map repeated for the n Monte-Carlo samples drawn per image -- lets the sampling path
 * run the x-only part of the prior (priors.py:80-95, 6.2 of 49.2 GFLOP per sample at 192 x 192) ONCE per image where the
 * reference's np.tile(x, [n,1,1,1]) (phiseg_model.py:577-585) recomputes it n times */
int phx_repeat_batch(const void* x, void* out, int B, size_t bytes_per_sample, int n, void* stream);
int phx_bn_infer_scale_shift_multi(const void* descs_dev, int n, void* stream);
int phx_affine_act(const void* x, int x_dt, const float* scale, const float* shift, void* y, int y_dt,
                   int NS, int P, int C, int act, void* stream);
/* fused forms used by the engine: finalize + affine_act in one launch (every thread re-derives the statistics of its
 * own channels; block (0, ns) publishes mean/rstd/scale/shift and the moving-average update), and
 * bwd_finalize + bwd_apply in one launch (dgamma / dbeta accumulated atomically by block (0, ns)). */
int phx_norm_apply_fused(const void* x, int x_dt, const float* sums, const float* pivot, const float* gamma,
                         const float* beta, float eps, void* y, int y_dt, float* mean, float* rstd, float* scale,
                         float* shift, float* moving_mean, float* moving_var, float momentum, int NS, int P, int C,
                         int G, int act, void* stream);
/* ... and with the 2 x 2 average pool of its output fused in (round 5; averagepool2D of pre_z[i - 1], posteriors.py:80-82 / priors.py:76-78,
 * tfwrapper/layers.py:44-54): a thread takes a 2 x 2 pixel quad, writes the four activations and their average (of the values as stored,
 * in phx_avgpool2x2_fwd's order: bit-identical to the two launches).  bf16, C % 8 == 0, H and W even; P = pixels per statistic. */
int phx_norm_apply_pool_supported(int H, int W, int C);
int phx_norm_apply_pool(const void* x, const float* sums, const float* pivot, const float* gamma, const float* beta, float eps, void* y,
                        void* y_pool, float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                        float momentum, int NS, int P, int C, int G, int H, int W, int act, void* stream);
/* ... and with a 1x1 HEAD fused in (the likelihood's top layer feeding y_lvl0, likelihoods.py:220: the head is the only reader of
 * a = act(norm(x))): y receives a as usual and y_head[NS * P][nout] = b_head + a w_head (w_head the HWIO 1x1 filter [C][nout]), computed
 * from the values just produced instead of by a pass of its own over a.  bf16 in / out, C / 8 a power of two <= 64, nout in {2, 4}
 * (phx_norm_head_supported).  Backward: phx_norm_bwd_reduce_head / phx_norm_bwd_apply_fused_head form dA = dy_head w_head^T on the fly.
 * y == NULL: a is not written (its one other reader, the head's filter gradient, re-forms it: phx_head1x1_wgrad_multi, xscale). */
int phx_norm_head_supported(int C, int nout, int x_dt, int y_dt);
int phx_norm_apply_fused_head(const void* x, int x_dt, const float* sums, const float* pivot, const float* gamma,
                              const float* beta, float eps, void* y, int y_dt, float* mean, float* rstd, float* scale,
                              float* shift, float* moving_mean, float* moving_var, float momentum, int NS, int P, int C,
                              int G, int act, const float* w_head, const float* b_head, int nout, float* y_head, void* stream);
int phx_norm_bwd_reduce_head(const float* dy_head, const float* w_head, int nout, const void* x, const float* scale, const float* shift,
                             const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act, int nrep,
                             void* stream);
int phx_norm_bwd_apply_fused_head(const float* dy_head, const float* w_head, int nout, const void* x, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, const float* gamma, const float* sums2,
                                  void* dx, float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot, float* dbias,
                                  int NS, int P, int C, int G, int act, int nrep, void* stream);
int phx_norm_bwd_apply_fused(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                             const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx,
                             int dx_dt, float* dgamma, float* dbeta, int NS, int P, int C, int G, int act, int nrep,
                             void* stream);
/* the same with the gradient of the convolution bias that precedes the normalisation (group / instance norm layers keep their
 * bias, tfwrapper/layers.py:126-132): dbias[c] += sum over (ns, p) of dx, in closed form from the per-channel sums of the
 * backward reduction (sums2) and of the forward statistics pass (fwd_sums / fwd_pivot as given to phx_norm_apply_fused) --
 * no pass over dx.  dbias == NULL: identical to phx_norm_bwd_apply_fused. */
int phx_norm_bwd_apply_fused_bias(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx,
                                  int dx_dt, float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot,
                                  float* dbias, int NS, int P, int C, int G, int act, int nrep, void* stream);
/* Small-map batch norm, training mode, bf16 NHWC (tfwrapper/normalisation.py:16-45 batch_norm -> tf.layers.
 * batch_normalization(training=True), fused with the activation of layers.py:134-135): the whole layer in ONE launch when
 * P = B*H*W <= 4096 and C % 16 == 0 (a block owns 16 channels of all pixels; two-pass variance; no atomics).
 * fwd: y = act(gamma * (x - mean) * rstd + beta); publishes mean / rstd / scale / shift [C] and, if momentum > 0, the
 * moving-average update (moving -= (moving - batch) * momentum, unbiased variance).
 * bwd: dx from dA, the saved x and statistics; dgamma / dbeta are accumulated (+=). */
int phx_bn_small_supported(int P, int C, int dt);
/* x_dt = PHX_BF16, or PHX_F32 for P <= 1024 (round 5): the pre-normalisation tensor of the 2 x 2 / 4 x 4 levels stays in fp32
 * (phx_conv3x3_bf16 with y_f32) -- a channel is normalised from a few dozen to a few hundred values there, and bf16 rounding of x
 * (2^-9 of the channel MEAN) is blown up with the spread: the benchmarked precision's two coarsest KL terms trained 40 % high. */
int phx_bn_small_fwd(const void* x, int x_dt, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                     float* scale, float* shift, float* moving_mean, float* moving_var, float momentum, int P, int C,
                     int act, void* stream);
int phx_bn_small_bwd(const void* dA, const void* x, int x_dt, const float* scale, const float* shift, const float* mean,
                     const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, int P, int C, int act,
                     void* stream);
/* The WIDE form of the same layer for P <= 1024 (the 2 x 2 / 4 x 4 levels at batch 64; round 5): the fp32 pre-normalisation tensor is
 * given as the nz split-K slices xs[z][P][C] the convolution left in its workspace (phx_conv3x3_bf16 with y_f32 and sum_slices = 0;
 * nz = plan.ksplit; nz = 1: xs is the tensor itself) -- this launch is also the split-K finishing pass: it adds the slices
 * in slice order, writes their sum to xsum[P][C] (nz > 1; what the backward pass reads) and normalises.  A block owns four channels of
 * all pixels: C / 4 blocks.  bwd: dA as a bf16 tensor, or as the nzd fp32 slices the consumer's split-K data gradient left
 * (rounded to bf16 after the sum, as its finishing pass would have); x is the fp32 tensor. */
int phx_bn_wide_supported(int P, int C);
int phx_bn_wide_fwd(const float* xs, int nz, float* xsum, const float* gamma, const float* beta, float eps, void* y, float* mean,
                    float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, float momentum, int P, int C,
                    int act, void* stream);
int phx_bn_wide_bwd(const void* dA, const float* dA_slices, int nzd, const float* x, const float* scale, const float* shift,
                    const float* mean, const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, int P, int C,
                    int act, void* stream);
/* Group / instance norm (tfwrapper/normalisation.py:3-36), bf16 NHWC, the whole layer in ONE launch when a sample has
 * P = H*W <= 256 pixels (maps up to 16 x 16) and the statistic is per channel (G == C: instance norm) or per 16-channel
 * group (G * 16 == C: group_norm2D's default groups for C >= 32): a wave owns (sample, 16-channel slice) pairs, keeps the
 * slice in registers, two-pass variance, wave shuffles only; no atomics in the forward pass, one add per channel and block in
 * the backward pass.
 *   fwd: ws == NULL: x is the input.  ws != NULL: the convolution ran split-K (phx_conv3x3_bf16 with a workspace and y == NULL): the
 *        nz fp32 slices ws[z][NS*P][C] are summed, `bias` (the convolution bias, may be NULL) added, the bf16 result written
 *        to x and normalised.  mean / rstd: [NS][G]; scale / shift: [NS][C].
 *   bwd: dx, dgamma += , dbeta += , and (dbias != NULL) the gradient of the convolution bias in front of the layer,
 *        dbias[c] += sum over samples and pixels of dx -- closed form, no extra pass. */
int phx_norm_small_supported(int NS, int P, int C, int G, int dt);
int phx_norm_small_fwd(void* x, const float* ws, int nz, const float* bias, const float* gamma, const float* beta, float eps,
                       void* y, float* mean, float* rstd, float* scale, float* shift, int NS, int P, int C, int G, int act,
                       void* stream);
int phx_norm_small_bwd(const void* dA, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, float* dbias, int NS, int P,
                       int C, int G, int act, void* stream);
/* backward of y = act(norm(x)):  g = dA * act'(.);  sums2[nrep][NS][C][2] += {sum g, sum g*xhat}: block b adds into
 * replica b % nrep (same-address atomics serialise at ~45 ns each); phx_norm_bwd_apply_fused sums the replicas, the other
 * consumers take nrep = 1 */
int phx_norm_bwd_reduce(const void* dA, int da_dt, const void* x, int x_dt, const float* scale,
                        const float* shift, const float* mean, const float* rstd, float* sums2,
                        int NS, int P, int C, int G, int act, int nrep, void* stream);
/* per (ns,g): S[ns][g][2] = sum_{c in g} gamma_c * sums2[ns][c][*];  dgamma[c] += sum_ns sums2[..][1], dbeta += [..][0] */
/* One-pass batch-norm backward (round 6): what phx_norm_bwd_reduce + phx_norm_bwd_apply_fused compute for batch norm (NS = 1,
 * G = C), bf16 dA / x / dx, in ONE launch -- every thread keeps its (dA, x) values in registers across a grid barrier, so both
 * tensors are read once.  For tensors of at most 256 blocks x 256 threads x 16 pixels x 8 channels (phx_bn_bwd_onepass_supported; relu layers;
 * never in deterministic mode: the partial sums are added atomically).  sums2[nrep][C][2] and barrier[phx_bn_bwd_onepass_barrier_words()]
 * must be ZERO at launch (the engine carves both from its per-step zero arena); barrier[288] != 0 afterwards = the barrier timed out.
 * At most two such launches may run concurrently (one block per CU each: both stay resident). */
int phx_bn_bwd_onepass_supported(int P, int C, int act);
int phx_bn_bwd_onepass_barrier_words(void);
int phx_bn_bwd_onepass(const void* dA, const void* x, const float* scale, const float* shift, const float* mean, const float* rstd,
                       const float* gamma, float* sums2, unsigned* barrier, void* dx, float* dgamma, float* dbeta, int P, int C, int act,
                       int nrep, void* stream);
int phx_norm_bwd_finalize(const float* sums2, const float* gamma, float* S, float* dgamma, float* dbeta,
                          int NS, int C, int G, void* stream);
/* dx = rstd * (gamma*g - S0/m - xhat*S1/m),  m = P*C/G */
int phx_norm_bwd_apply(const void* dA, int da_dt, const void* x, int x_dt, const float* scale,
                       const float* shift, const float* mean, const float* rstd, const float* gamma,
                       const float* S, void* dx, int dx_dt, int NS, int P, int C, int G, int act, void* stream);
/* act backward without a norm: dpre = dy * act'(y) evaluated from the stored OUTPUT y */
int phx_act_bwd(const void* dy, int dy_dt, const void* y, int y_dt, void* dpre, int dpre_dt, size_t n, int act,
                void* stream);

/* ---- pooling / resize / concat (tfwrapper/layers.py:44-54, 336-345, 70-78; tf.concat) --------------- */
int phx_avgpool2x2_fwd(const void* x, int dt, void* y, int B, int H, int W, int C, void* stream);
int phx_avgpool2x2_bwd(const void* dy, int dt, void* dx, int B, int H, int W, int C, void* stream);
/* the same, ADDED to dx (a tensor with several readers: the later gradient contributions accumulate in place, no add pass) */
int phx_avgpool2x2_bwd_acc(const void* dy, int dt, void* dx, int B, int H, int W, int C, void* stream);
/* TF 1.12 ResizeBilinear(align_corners=False), legacy coordinates, factor 2 */
/* ---- bilinear_upsample2D -> conv2D 3x3 without the up-sampled tensor (tfwrapper/layers.py:336-345 into :123; likelihoods.py:200-204):
 * the elementwise half of the phase form (csrc/upconv.hip).  The matrix launches are the ordinary ones:
 *   y_packed [B][h][w][4 Cout] = phx_conv3x3_bf16(x [B][h][w][Cin], weff_fwd)           exact outside the frame (hi rows / columns 0, 2n-2, 2n-1)
 *   fr [1][6B][2w][Cout] = phx_conv3x3_bf16(f_rows, wpk_fwd of W),  fc [1][6B][2h][Cout] = phx_conv3x3_bf16(f_cols, wt_fwd)
 * y_packed[b][i][j][(a, b', co)] is hi-res pixel (2i + a, 2j + b'): a channels-last tensor of 4 B h w pixels for the norm kernels.
 * Backward: phx_upconv_frame_gather_dy moves the frame's gradient out of dy_packed (and zeroes it there), the three data gradients /
 * filter gradients are the ordinary launches on (dy_packed, weff_dgrad), (dfr, wpk_dgrad of W), (dfc, wt_dgrad);
 * phx_upconv_frame_scatter_dx adds the gathered rows' gradient into dx, phx_upconv_fold_wgrad folds dWeff [3][3][Cin][4 Cout] and
 * dWt [3][3][Cin][Cout] (fp32) into dw_hwio (+=). */
int phx_upconv_supported(int B, int h, int w, int Cin, int Cout);
/* (bias / bias4, nullable together: the convolution bias repeated for the 4 Cout packed columns -- group / instance norm layers keep it) */
int phx_upconv_pack(const float* w_hwio, void* weff_fwd, void* weff_dgrad /* nullable */, void* wt_fwd, void* wt_dgrad /* nullable */,
                    const float* bias, float* bias4, int Cin, int Cout, void* stream);
int phx_upconv_fold_wgrad(const float* dweff, const float* dwt /* nullable */, float* dw_hwio, int Cin, int Cout, void* stream);
int phx_upconv_frame_gather(const void* x, void* f_rows, void* f_cols, int B, int h, int w, int C, void* stream);
int phx_upconv_frame_scatter(const void* fr, const void* fc, void* y_packed, int B, int h, int w, int Cout, void* stream);
int phx_upconv_frame_gather_dy(void* dy_packed, void* dfr, void* dfc, int B, int h, int w, int Cout, void* stream);
int phx_upconv_frame_scatter_dx(const void* df_rows, const void* df_cols, void* dx, int B, int h, int w, int C, void* stream);
/* ... and the two permutations ride on the layer's own normalisation passes (NS * P = B * 4 h w pixels; a sample group is one image
 * -- group / instance norm -- or the whole batch): the apply pass reads the packed y and writes the hi-res activation, the two backward
 * passes read the hi-res dA and the packed y and write the packed dy (fwd_sums / fwd_pivot / dbias as phx_norm_bwd_apply_fused_bias) */
int phx_norm_apply_fused_d2s(const void* x, int x_dt, const float* sums, const float* pivot, const float* gamma, const float* beta, float eps,
                             void* y, int y_dt, float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                             float momentum, int NS, int P, int C, int G, int act, int h, int w, void* stream);
int phx_norm_bwd_reduce_s2d(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                            const float* mean, const float* rstd, float* sums2, int NS, int P, int C, int G, int act, int nrep, int h,
                            int w, void* stream);
int phx_norm_bwd_apply_fused_s2d(const void* dA, int da_dt, const void* x, int x_dt, const float* scale, const float* shift,
                                 const float* mean, const float* rstd, const float* gamma, const float* sums2, void* dx, int dx_dt,
                                 float* dgamma, float* dbeta, const float* fwd_sums, const float* fwd_pivot, float* dbias, int NS, int P,
                                 int C, int G, int act, int nrep, int h, int w, void* stream);
int phx_depth_to_space2(const void* packed, void* hi, int B, int h, int w, int C, void* stream);
int phx_space_to_depth2(const void* hi, void* packed, int B, int h, int w, int C, void* stream);
int phx_bilinear_up2x_fwd(const void* x, int dt, void* y, int B, int h, int w, int C, void* stream);
int phx_bilinear_up2x_bwd(const void* dy, int dt, void* dx, int B, int h, int w, int C, void* stream);
/* the same, ADDED to dx (a tensor with several readers: the later gradient contributions accumulate in place, no add pass) */
int phx_bilinear_up2x_bwd_acc(const void* dy, int dt, void* dx, int B, int h, int w, int C, void* stream);
int phx_concat2(const void* a, int Ca, const void* b, int Cb, void* out, size_t npix, int dt, void* stream);
int phx_split2(const void* in, void* a, int Ca, void* b, int Cb, size_t npix, int dt, void* stream);
int phx_add_inplace(void* dst, const void* src, size_t n, int dt, void* stream);
/* out[c] += sum over npix of x[p][c]  (bias gradient of the MFMA conv path) */
int phx_channel_sum_accumulate(const void* x, int dt, float* out, size_t npix, int C, void* stream);
int phx_cast(const void* src, int src_dt, void* dst, int dst_dt, size_t n, void* stream);
/* posterior input: concat[x, one_hot(s) - 0.5] (phiseg_model.py:29, posteriors.py:87) -> [npix][1+C] */
int phx_posterior_input(const float* x, const uint8_t* s, void* out, int out_dt, size_t npix, int nlabels,
                        void* stream);
/* global average pool [B,P,C] -> [B,C] and its adjoint; broadcast [B,C] -> [B,P,C] and its adjoint */
int phx_global_avgpool_fwd(const float* x, float* y, int B, int P, int C, void* stream);
int phx_global_avgpool_bwd(const float* dy, float* dx, int B, int P, int C, void* stream);
int phx_broadcast_pixels_fwd(const float* z, void* out, int out_dt, int B, int P, int C, void* stream);
int phx_broadcast_pixels_bwd(const void* dout, int dt, float* dz, int B, int P, int C, void* stream);

/* ---- reparameterisation: z = mu + sigma * N(0,1) (posteriors.py:108,128; priors.py:100,120) --------- */
/* Philox4x32-10, counter = (e/4, sample_offset + b, stream_id, *step), key = seed; see oracle/philox.py. */
int phx_reparam_fwd(const float* mu, const float* sigma, float* z, int B, int per_sample, uint64_t seed,
                    const int32_t* step_dev, int stream_id, int sample_offset, void* stream);
int phx_reparam_bwd(const float* dz, float* dsigma, int B, int per_sample, uint64_t seed,
                    const int32_t* step_dev, int stream_id, int sample_offset, void* stream);
int phx_philox_normal(float* out, int B, int per_sample, uint64_t seed, const int32_t* step_dev, int stream_id,
                      int sample_offset, void* stream);

/* ---- losses (phiseg_model.py:210-262) -------------------------------------------------------------- */
/* residual multinoulli loss over L levels: logits s[l] are [B, H>>shift[l], W>>shift[l], C] fp32 (the
 * NEAREST_NEIGHBOR resize of likelihoods.py:221 is folded into the read).  losses[l] = inv_batch * sum CE_l
 * (written; `losses` must hold 8 + 512 floats, the tail is scratch for the partial sums; nullable).
 * ds[l] (nullable; level 0 is written, coarser levels must be zeroed) += weight*inv_batch * d(sum_l CE_l)/d s[l].
 * labels may be NULL when only s_out (= sum_l s[l] at full resolution, nullable) / sm_out (= softmax(s_out),
 * nullable) are wanted (sampling path, phiseg_model.py:106-109). */
int phx_residual_ce(const float* const* s, float* const* ds, const int* shift, int L, const uint8_t* labels,
                    int B, int H, int W, int C, float weight, float inv_batch, float* losses, float* s_out,
                    float* sm_out, void* stream);
/* KL(N(mu0,s0^2) || N(mu1,s1^2)) summed over n = B*per_sample elements: loss += level_w*inv_batch*0.5*sum(...);
 * gradients (nullable) are WRITTEN (not accumulated), scaled by grad_scale*level_w*inv_batch. */
int phx_kl_diag_gauss(const float* mu0, const float* s0, const float* mu1, const float* s1, size_t n,
                      float level_w, float inv_batch, float grad_scale, float* loss, float* dmu0, float* ds0,
                      float* dmu1, float* ds1, void* stream);
/* All levels of the hierarchical KL term (phiseg_model.py:265-287) in ONE launch.  ptrs: 9 device pointers per level {mu0, sigma0,
 * mu1, sigma1, dmu0, dsigma0, dmu1, dsigma1, loss} (gradient pointers NULL: loss only); n[l] elements, level_w[l] = 4^l weights;
 * loss[l] is ACCUMULATED: zero it before the launch. */
int phx_kl_diag_gauss_multi(const void* const* ptrs, const size_t* n, const float* level_w, int L, float inv_batch, float grad_scale,
                            void* stream);

/* ---- optimiser: tf.train.AdamOptimizer (phiseg_model.py:137-141), TF 1.12 epsilon-hat form ---------- */
/* t = *step_dev + 1;  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m,v EMA;  p -= lr_t*m/(sqrt(v)+eps).  One launch over
 * the flat parameter arena.  grad_scale multiplies g first (1/world for averaged all-reduce). */
int phx_adam_tf1(float* p, const float* g, float* m, float* v, size_t n, const float* lr_dev, float beta1,
                 float beta2, float eps, const int32_t* step_dev, void* stream);
int phx_step_increment(int32_t* step_dev, void* stream);
/* diagnostic: *dst_u64 = the device's constant-rate (100 MHz) wall clock when the stream reaches this point; the engine
 * (PHX_STAMPS=1) brackets every operator with these to draw the per-lane timeline of a replayed plan without a profiler
 * (rocprofv3's kernel trace serialises the lanes).  No reference counterpart. */
int phx_stamp(void* dst_u64, void* stream);
int phx_sum_scalars(const float* in, int n, float* out, void* stream);
/* out = sum_i weights[i] * (*ptrs[i]), n <= 16; ptrs / weights are HOST arrays of device pointers / floats
 * (loss_tot of phiseg_model.py:118-130) */
int phx_weighted_sum(const float* const* ptrs, const float* weights, int n, float* out, void* stream);

/* weight decay (phiseg_model.py:290-299): out = scale * sum_i mask[i] p[i]^2 / 2 over the flat parameter arena (mask = 1 on the members of
 * the 'weight_variables' collection; work256: 256 floats of scratch), and its gradient g += alpha * mask * p */
int phx_l2_masked(const float* p, const float* mask, size_t n, float scale, float* work256, float* out, void* stream);
int phx_axpy_masked(float* g, const float* p, const float* mask, size_t n, float alpha, void* stream);

/* ---- validation metrics on the device ---------------------------------------------------
 * What phiseg_model._do_validation computes per validation image (phiseg_model.py:586-613 calling
 * utils.generalised_energy_distance utils.py:270-322, utils.variance_ncc_dist utils.py:326-370 and the Dice loop), for I
 * images at once:
 *   sm   [I][N][P][C] f32 soft-max of the N Monte-Carlo samples (P = X * Y pixels, 2 <= C <= 8)
 *   gt   [I][M][P] u8 annotations (M <= 8), sref [I][P] u8 the annotation the Dice is taken against
 *   out  [I][2 + 8] f32: GED over labels label0 .. C-1 (the reference passes label_range = 1 .. nlabels-1), NCC,
 *        Dice of arg-max(mean soft-max) vs sref for labels 0 .. C-1 (remaining slots unused)
 *   work scratch of phx_validation_metrics_ws_bytes bytes */
size_t phx_validation_metrics_ws_bytes(int I, int N, int M, int P, int C);
int phx_validation_metrics(const float* sm, const unsigned char* gt, const unsigned char* sref, void* work, size_t work_bytes,
                           int I, int N, int M, int P, int C, int label0, float* out, void* stream);

/* ---- transposed convolution (tfwrapper/layers.py:197-258, tf.nn.conv2d_transpose) --------------
 * x [B,H,W,Cin] -> y [B, H*sh, W*sw, Cout], filter w_hwoi [kh][kw][Cout][Cin] fp32 (TF's layout), SAME padding, optional bias and
 * activation.  dgrad: dy [B, H*sh, W*sw, Cout] -> dx [B,H,W,Cin].  wgrad ACCUMULATES into dw_hwoi.  Direct (untuned) kernels: no
 * shipped experiment calls this layer. */
int phx_tconv2d_fwd(const void* x, int x_dt, const float* w_hwoi, const float* bias, void* y, int y_dt, int B, int H, int W,
                    int Cin, int Cout, int kh, int kw, int sh, int sw, int act, void* stream);
int phx_tconv2d_dgrad(const void* dy, int dy_dt, const float* w_hwoi, void* dx, int dx_dt, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, void* stream);
int phx_tconv2d_wgrad(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwoi, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, void* stream);

/* ---- layers of tfwrapper/layers.py without a call site in the shipped experiments (csrc/gconv.hip; plain direct kernels) ----
 * General 2-D convolution, SAME padding, stride (sh, sw) and dilation (dh, dw), NHWC, HWIO fp32 filter [kh][kw][Cin][Cout]:
 * conv2D(strides=...) (layers.py:94-145) and dilated_conv2D = tf.nn.atrous_conv2d (layers.py:378-425).  Output size
 * ceil(H / sh) x ceil(W / sw) (phx_gconv2d_out_size); dgrad writes dx [B,H,W,Cin]; wgrad ACCUMULATES into dw_hwio. */
int phx_gconv2d_out_size(int H, int W, int sh, int sw, int* Ho, int* Wo);
int phx_gconv2d_fwd(const void* x, int x_dt, const float* w_hwio, const float* bias, void* y, int y_dt, int B, int H, int W,
                    int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw, int act, void* stream);
int phx_gconv2d_dgrad(const void* dy, int dy_dt, const float* w_hwio, void* dx, int dx_dt, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, int dh, int dw, void* stream);
int phx_gconv2d_wgrad(const void* x, int x_dt, const void* dy, int dy_dt, float* dw_hwio, int B, int H, int W, int Cin, int Cout,
                      int kh, int kw, int sh, int sw, int dh, int dw, void* stream);
/* maxpool2D (layers.py:18-28): tf.nn.max_pool 2x2 / stride 2 / SAME, y [B, ceil(H/2), ceil(W/2), C]; the gradient goes to the
 * first (row-major) maximum of each window */
int phx_maxpool2x2_fwd(const void* x, int dt, void* y, int B, int H, int W, int C, void* stream);
int phx_maxpool2x2_bwd(const void* x, const void* dy, int dt, void* dx, int B, int H, int W, int C, void* stream);
/* dst[b, y, x, :] = src[b, y + off_y, x + off_x, :] inside the source, 0 outside: pad_to_size (layers.py:625-650, negative
 * offsets) and the centre crop of crop_and_concat (layers.py:586-622, positive offsets); the gradient is the same call with the
 * sizes swapped and the offsets negated */
int phx_spatial_window(const void* src, void* dst, int dt, int B, int Hs, int Ws, int Hd, int Wd, int C, int off_y, int off_x,
                       void* stream);
/* strided window with a channel offset: dst[b, y, x, c] = src[b, y sy + off_y, x sx + off_x, c + off_c] inside the source, 0
 * outside -- the skip path of the residual units (layers.py:465-470: tf.pad along the channel axis, identity[:, ::2, ::2, :]);
 * _bwd writes the gradient with respect to src (every element, zeros where nothing maps) */
int phx_window4_fwd(const void* src, void* dst, int dt, int B, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int sy, int sx,
                    int off_y, int off_x, int off_c, void* stream);
int phx_window4_bwd(const void* ddst, void* dsrc, int dt, int B, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int sy, int sx,
                    int off_y, int off_x, int off_c, void* stream);
/* y = act(a + b): tf.add + activation at the end of a residual unit (layers.py:474-475); gradient: phx_act_bwd on y */
int phx_add_act(const void* a, const void* b, void* y, int dt, size_t n, int act, void* stream);
/* dropout (layers.py:653-668, tf.nn.dropout): y = x * keep / keep_prob with keep[b][e] = (u < keep_prob), u the 24-bit uniform of
 * word e % 4 of Philox block e / 4 under (seed, *step_dev, stream_id, sample_offset + b); the backward pass is the same call on dy */
int phx_dropout(const void* x, void* y, int dt, size_t per_sample, int B, float keep_prob, uint64_t seed, const int32_t* step_dev,
                int stream_id, int sample_offset, void* stream);

/* ---- mini-batch producer on the device ---------------------------------------------------
 * Replaces data/batch_provider.py:43-67 (next_batch), 131-137 (_select_random_label) and 140-272 (_augmentation_function with
 * the cv2 helpers of utils.py:18-38) for a data set resident in HBM: images [N][X][Y] f32, labels [N][X][Y][A] u8 (A annotators).
 * params_dev: B records of phx_augment_param_bytes() bytes in device memory,
 *   { int src, annot, flags (1 rotate | 2 crop-scale | 4 fliplr | 8 flipud), r_y, p_x, p_y; double iM[6] }
 * iM = inverse of cv2.getRotationMatrix2D((Y/2, X/2), angle, 1).  Output x_out [B][X][Y] f32, s_out [B][X][Y] u8 -- the plan's
 * x_input / s_input buffers can be written directly.  nlabels <= 4 (labels are interpolated as one-hot planes and arg-maxed). */
int phx_augment_param_bytes(void);
int phx_augment_batch(const float* images, const unsigned char* labels, const void* params_dev, float* x_out,
                      unsigned char* s_out, int B, int X, int Y, int A, int nlabels, void* stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI ------------------------------------------------
 * The reference is single-process, single-device (phiseg_model.py:151-157); the data-parallel design shards the batch over
 * ranks (one process per GPU) and exchanges ONE sum of the flat fp32 gradient arena per step.  RCCL is dlopen'ed on first use.
 *   phx_comm_unique_id : rank 0 creates the 128-byte rendezvous id and hands it to the other ranks out of band
 *                        (phiseg_code_amd/distributed.py broadcasts it through torch.distributed's store)
 *   phx_comm_init      : every rank, with its HIP device current; world == 1 is valid (the all-reduce is the identity)
 *   phx_comm_allreduce_sum_f32 : in-place sum of buf[0..n) over the ranks, split into bucket_elems-sized pieces (0: one piece)
 *                        inside one RCCL group, ENQUEUED on `stream` (ordered after / before the work around it on that stream;
 *                        nothing blocks the host)
 * Errors: PHX_E_COMM with the RCCL error text in phx_last_error. */
/* dlopen librccl and bind its entry points (idempotent; the other phx_comm_* calls do it implicitly): a per-rank check that can run
 * before any collective bootstrap step */
int phx_comm_load_api(void);
int phx_comm_unique_id(void* id128);
int phx_comm_init(void** comm, int world, int rank, const void* id128);
int phx_comm_allreduce_sum_f32(void* comm, float* buf, size_t n, size_t bucket_elems, void* stream);
int phx_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PHX_H */
