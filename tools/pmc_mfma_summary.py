"""Per-kernel MFMA / LDS utilisation from one rocprofv3 --pmc pass (csv) of bench.py.

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs): the busy counter sums the
matrix-pipe cycles of all SIMDs (32 per v_mfma_f32_32x32x16_bf16), GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs.
LDS busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256 CUs); conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections, csv, sys
f, steps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
rows = []
for k, c in agg.items():
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    if g <= 0:
        continue
    cu_cycles = g / 8.0 * 256.0
    rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), k, cnt[k] / steps, g / 8.0 / steps, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cu_cycles * 4.0),
                 c.get("SQ_LDS_IDX_ACTIVE", 0.0) / cu_cycles, c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)))
rows.sort(reverse=True)
lines = ["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- python bench.py --steps 2 --warmup 2",
         "# phiseg_7_5 128x128 bf16 B=64.  MFMA util = matrix-pipe busy cycles / (active cycles x 1024 SIMDs), per kernel over all its launches",
         "%-64s %9s %14s %10s %9s %13s" % ("kernel", "calls/st", "XCD cyc/step", "MFMA util", "LDS busy", "LDS conflict")]
for r in rows[:24]:
    lines.append("%-64s %9.1f %14.0f %9.1f%% %8.1f%% %12.1f%%" % (r[1][:64], r[2], r[3], 100 * r[4], 100 * r[5], 100 * r[6]))
tot_m = sum(r[0] for r in rows)
tot_g = sum(agg[k].get("GRBM_GUI_ACTIVE", 0.0) for k in agg)
lines.append("ALL KERNELS: MFMA util %.1f%% of the active cycles" % (100 * tot_m / (tot_g / 8.0 * 1024.0)))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
