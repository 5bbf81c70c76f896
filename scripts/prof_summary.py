"""Summarise a rocprofv3 kernel_stats CSV (per-kernel calls/avg/total/%)."""
import csv, sys, glob, re
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = re.sub(r"\(anonymous namespace\)::|_GLOBAL__N_1", "", r["Name"])[:100]
    print("%-100s calls %6s avg %9.1f us  tot %8.2f ms %5.1f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
