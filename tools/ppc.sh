#!/bin/bash
# dev: compile conv_pp.hip alone, print the register / spill table; -S: also dump ISA to /tmp/pp.s
cd /root/repo/phiseg_code_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -c conv_pp.hip -o build_conv_pp.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning: |Function Name|VGPRs:|Spill|ScratchSize|SGPRs:" | sed 's/conv_pp.hip:[0-9]*:1: remark: //; s/\[-Rpass.*//' | paste - - - - - - | sed 's/_ZN12_GLOBAL__N_112k_conv3x3_pp//; s/EEvPKt.*Dual//'
if [ "$1" == "-S" ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -S -o /tmp/pp.s conv_pp.hip --cuda-device-only 2>/dev/null; fi
