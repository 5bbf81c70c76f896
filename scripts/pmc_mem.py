"""Summarise memory-side PMC passes of scripts/kbench.py: per kernel family the counters averaged per launch.
usage: python scripts/pmc_mem.py <rocprof out dir> [...]"""
import csv, glob, sys, collections
FAM = (("conv_gemm_kernel", "conv_gemm"), ("materialize_kernel", "materialize"), ("dw2_fwd", "dw_fwd"), ("bn_bwd_kernel", "bn_bwd"))
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[0])):
        name = next((v for k, v in FAM if k in r["Kernel_Name"]), None)
        if name is None: continue
        a = agg[name][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for name, cs in sorted(agg.items()):
        print("%-12s" % name + "  ".join("%s=%.4g" % (c.replace("_sum", ""), v[1] / v[0]) for c, v in sorted(cs.items())))
