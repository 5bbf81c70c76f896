#!/usr/bin/env python3
"""Launches that leave CUs idle: workgroups < 400 and duration > 8 us, from a rocprofv3 kernel_trace.csv."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    blocks = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // wg
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if blocks < 400 and dur > 8:
        n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name']); n = re.sub(r"\(.*", "", n)[-52:]
        a = agg[(n, blocks)]; a[0] += 1; a[1] += dur
for (n, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-52s blocks %4d n/step %5.1f ms/step %6.3f avg %6.1f us" % (n, b, c / steps, t / steps / 1e3, t / c))
print("sum ms/step %.3f" % (sum(t for c, t in agg.values()) / steps / 1e3))
