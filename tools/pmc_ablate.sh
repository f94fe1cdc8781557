#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ablate
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $out/p1 -- python tools/bench_fwd_ablate.py ${1:-pp} > $out/p1.log 2>&1)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_ablate/p1/**/*counter_collection.csv", recursive=True)[0]
dur = collections.defaultdict(list)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r["Grid_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg):
    if "conv3x3" not in k[0]: continue
    a = {c: sum(v) / len(v) for c, v in agg[k].items()}
    d = sum(dur[k]) / len(dur[k]) / 1e3
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    print("%-46s grid %8s  %7.1f us  %7.0f kcyc  %.2f GHz  mfma util %.2f  wait_any %.2f wait_inst %.2f active %.2f" % (
        k[0][-46:], k[1], d, cyc / 1e3, cyc / d / 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
        a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
PY
