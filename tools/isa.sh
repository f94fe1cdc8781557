#!/bin/bash
# dev helper: device ISA of conv_mfma.hip -> /tmp/cm2.s; extract one kernel by mangled-name prefix ($1) -> /tmp/k.s
cd /root/repo/phiseg_code_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S -o /tmp/cm2.s conv_mfma.hip --cuda-device-only 2>&1 | grep -i "error" -A3 | head -20
awk -v pat="^$1" '$0 ~ pat && /:/ {f=1} f {print} f && /^\.Lfunc_end/ {exit}' /tmp/cm2.s > /tmp/k.s
wc -l /tmp/k.s
grep -n "\.name:.*$1" -A12 /tmp/cm2.s | grep "vgpr\|agpr"
