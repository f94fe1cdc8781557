#!/bin/bash
# Build libphx.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libphx.so
SRCS="runtime.hip elementwise.hip losses_opt.hip conv_direct.hip conv_mfma.hip conv_wgrad.hip conv_pp.hip conv_c32.hip heads.hip metrics.hip comm.hip augment.hip tconv.hip gconv.hip"
OBJS=""
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
done
wait
for o in $OBJS; do
  [ -f "$o" ] || { echo "build.sh: compiling ${o#build_} failed" >&2; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o $OUT
echo "built $(realpath $OUT)"
