"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv)."""
import csv, glob, re, sys, json
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|_GLOBAL__N_1", "", r["Kernel_Name"])
    if pat and not re.search(pat, name): continue
    k = (name[:70], r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (name, c), (n, tot) in sorted(agg.items()):
    print("%-70s %-12s launches %6d  avg %14.1f  total %16.1f" % (name, c, n, tot / n, tot))
