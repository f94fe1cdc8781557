"""Summarise the PMC passes of tools/pmc_conv.sh: per (kernel, grid) mean counter values."""
import collections, csv, glob, sys
d = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""))
        a = agg[(k, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in agg.items():
        tab[k][c] = v / n
for k in sorted(tab):
    if "conv3x3" not in k[0]:
        continue
    print(k)
    for c, v in tab[k].items():
        print("    %-32s %16.0f" % (c, v))
