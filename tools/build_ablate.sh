#!/bin/bash
# dev: libphx variants with parts of the MFMA conv kernels disabled (PHX_ABLATE bit mask: fwd 1 no global loads, 2 no LDS
# stores, 4 no MFMAs, 8 no output stores) -> ab/libphx_ab<N>.so (select with PHX_LIB)
set -e
cd "$(dirname "$0")/../phiseg_code_amd/csrc"
bash build.sh > /dev/null
mkdir -p ../../ab
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DPHX_ABLATE=$n -c conv_mfma.hip -o /tmp/ab_conv_$n.o 2>/dev/null &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DPHX_ABLATE=$n -c conv_wgrad.hip -o /tmp/ab_wgrad_$n.o 2>/dev/null &
done
wait
OBJS=$(ls build_*.o | grep -v -e build_conv_mfma.o -e build_conv_wgrad.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/ab_conv_$n.o /tmp/ab_wgrad_$n.o -ldl -o ../../ab/libphx_ab$n.so
  echo built ab/libphx_ab$n.so
done
