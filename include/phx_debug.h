/* phx_debug.h -- entry points that exist ONLY in the test build of the library (phiseg_code_amd/libphx_dbg.so: the sources of libphx.so
 * compiled with -DPHX_DEBUG_BUILD by csrc/build.sh).  libphx.so exports none of them and has no process-global mutable kernel policy:
 * there the policy below is the constant 1 and the pair kernel's grid is one work-group per CU.
 *
 * The kernel tests (tests/test_kernels_gpu.py, fixture `Ld`) use this build to force every forward / data-gradient kernel family of
 * tf.nn.conv2d 3x3 SAME (tfwrapper/layers.py:123) onto shapes small enough for the CPU oracle, and dev tools (tools/bench_pp.py,
 * tools/trace_pp.py) to time one family against another on the same shape.
 * The policy is process-wide in that build and must not change while a Plan built under another policy is alive: a plan sizes its
 * statistics rows from phx_conv3x3_mfma_bf16_tiles, which follows the policy. */
#ifndef PHX_DEBUG_H
#define PHX_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* kernel-selection policy of the forward / data-gradient launches: 1 = the measured policy (libphx.so's constant), 0 = never,
 * 2 = whenever the shape is eligible -- large_maps: the 16 x 32-tile large-map kernels (k_conv3x3_pp, k_conv3x3_c32); big_tiles: the
 * 16 x 32-tile instantiations of the 256-pixel kernel */
int phx_debug_conv_policy(int large_maps, int big_tiles);
/* persistent grid of the pair kernel (0: one work-group per CU) */
int phx_debug_pair_kernel_grid(int blocks);
#ifdef __cplusplus
}
#endif
#endif
