#!/bin/bash
# GPU box: clock / MFMA utilisation / wait split of k_conv3x3_pp and k_conv3x3_fwd_dma128 per shape (one rocprofv3 pass per shape)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_pp
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for sh in ${@:-0 1 2}; do
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $out/s$sh -- python tools/bench_pp.py shape=$sh > $out/s$sh.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_pp/s*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    print("##", open(d.rstrip("/") + ".log").read().strip().split("\n")[-1][:24])
    dur = collections.defaultdict(list)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(agg):
        if "conv3x3" not in k: continue
        a = {c: sum(v) / len(v) for c, v in agg[k].items()}
        dd = sum(dur[k]) / len(dur[k]) / 1e3
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        print("%-40s n=%3d %7.1f us  %7.0f kcyc  %.2f GHz  mfma util %.2f  wait_any %.2f wait_inst %.2f active %.2f lds_active/cyc %.2f" % (
            k[-40:], len(dur[k]), dd, cyc / 1e3, cyc / dd / 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
            a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"],
            a["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)))
PY
