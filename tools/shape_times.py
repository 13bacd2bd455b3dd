import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if not n.startswith("rs_tree"): continue
    key = (n[8:], r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    agg[key][0] += 1; agg[key][1] += d
nb = int(sys.argv[2])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s vgpr %4s lds %6s calls %4d ms/batch %.3f" % (k[0], k[1], k[2], v[0], v[1] / nb))
