#!/bin/bash
# dev helper (GPU box): kernel traces of bench.py under two builds -> gpurun_out/trace_{a,b}.csv + comparison.  usage: trace_ab.sh libA libB [filter]
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  lib=$1; [ $v = b ] && lib=$2
  rm -rf /tmp/tl_$v
  PHX_LIB=/root/repo/$lib rocprofv3 --kernel-trace -d /tmp/tl_$v -o p --output-format csv -- python /root/repo/bench.py --steps 4 --warmup 2 > /dev/null 2>&1
  cp $(find /tmp/tl_$v -name "*kernel_trace.csv" | head -1) /root/repo/gpurun_out/trace_$v.csv
done
cd /root/repo && python tools/compare_traces.py gpurun_out/trace_a.csv gpurun_out/trace_b.csv "$3"
