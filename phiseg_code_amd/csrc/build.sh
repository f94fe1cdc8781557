#!/bin/bash
# Build libphx.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libphx.so
SRCS="runtime.hip elementwise.hip losses_opt.hip conv_direct.hip conv_f32_mfma.hip conv_mfma.hip conv_wgrad.hip conv_pp.hip conv_c32.hip heads.hip metrics.hip comm.hip augment.hip tconv.hip gconv.hip upconv.hip"
OBJS=""
DOBJS=""
for s in $SRCS; do
  o="build_${s%.hip}.o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ phx_common.h -nt "$o" ] || [ conv_common.h -nt "$o" ] || [ philox.h -nt "$o" ] || [ ../../include/phx.h -nt "$o" ]; then
    rm -f "$o"                       # (a failed compile must not leave the previous object behind for the link)
    # conv_pp.hip: no SLP vectorisation -- packed fp32 VALU (v_pk_fma_f32 / v_pk_add_f32 formed from adjacent scalar operations) costs
    # issue slots beside the partner wave's MFMA stream, and a v_pk_fma_f32 issued straight behind a buffer_store_dwordx4 whose data
    # registers it overwrites corrupted the second dword of the store for the last lanes (MI355X, ROCm 7.2: hipcc inserts no wait state)
    EXTRA=""
    [ "$s" = "conv_pp.hip" ] && EXTRA="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $EXTRA -c "$s" -o "$o" &
  fi
  OBJS="$OBJS $o"
  # the test build libphx_dbg.so: the two translation units that hold the kernel-selection policy compiled with -DPHX_DEBUG_BUILD
  # (phx_debug_conv_policy / phx_debug_pair_kernel_grid, include/phx_debug.h), every other object shared with libphx.so
  if [ "$s" = "conv_mfma.hip" ] || [ "$s" = "conv_pp.hip" ]; then
    d="build_${s%.hip}_dbg.o"
    if [ ! -f "$d" ] || [ "$s" -nt "$d" ] || [ phx_common.h -nt "$d" ] || [ conv_common.h -nt "$d" ] || [ ../../include/phx.h -nt "$d" ] || [ ../../include/phx_debug.h -nt "$d" ]; then
      rm -f "$d"
      EXTRA=""
      [ "$s" = "conv_pp.hip" ] && EXTRA="-fno-slp-vectorize"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DPHX_DEBUG_BUILD $EXTRA -c "$s" -o "$d" &
    fi
    DOBJS="$DOBJS $d"
  else
    DOBJS="$DOBJS $o"
  fi
done
wait
for o in $OBJS $DOBJS; do
  [ -f "$o" ] || { echo "build.sh: compiling ${o#build_} failed" >&2; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o $OUT
echo "built $(realpath $OUT)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $DOBJS -ldl -o ../libphx_dbg.so
echo "built $(realpath ../libphx_dbg.so) (test build: settable kernel policy, include/phx_debug.h)"
