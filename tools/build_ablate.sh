#!/bin/bash
# dev: libphx variants with parts of the MFMA forward conv kernel disabled (PHX_ABLATE bit mask) -> phiseg_code_amd/libphx_ab<N>.so
set -e
cd "$(dirname "$0")/../phiseg_code_amd/csrc"
bash build.sh > /dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DPHX_ABLATE=$n -c conv_mfma.hip -o /tmp/ab_conv_$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_runtime.o build_elementwise.o build_losses_opt.o build_conv_direct.o /tmp/ab_conv_$n.o build_heads.o build_metrics.o -o ../libphx_ab$n.so
  echo built libphx_ab$n.so
done
