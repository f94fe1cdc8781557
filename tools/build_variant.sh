#!/bin/bash
# Dev A/B builds of the library: tools/build_variant.sh NAME -DFLAG=V [...]  ->  phiseg_code_amd/libphx_NAME.so (all translation units
# recompiled with the flags into a scratch directory; the product objects are untouched).  Select it with PHX_LIB=<path> (runtime.py).
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/phiseg_code_amd/csrc
out=$src/build_variant_$name
mkdir -p "$out"
objs=""
for s in runtime.hip elementwise.hip losses_opt.hip conv_direct.hip conv_f32_mfma.hip conv_mfma.hip conv_wgrad.hip conv_pp.hip conv_c32.hip heads.hip metrics.hip comm.hip augment.hip tconv.hip gconv.hip upconv.hip; do
  extra=""; [ "$s" = "conv_pp.hip" ] && extra="-fno-slp-vectorize"
  (cd "$src" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $extra "$@" -c "$s" -o "$out/${s%.hip}.o") &
  objs="$objs $out/${s%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o "$here/phiseg_code_amd/libphx_$name.so"
rm -rf "$out"
echo "built $here/phiseg_code_amd/libphx_$name.so ($*)"
