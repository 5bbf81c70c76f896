#!/usr/bin/env python3
"""Group a rocprofv3 kernel_stats CSV by C-ABI entry point (kernel family): calls, total, share, average.
usage: python scripts/prof_families.py <rocprof out dir> [steps in the profiled process]"""
import csv, glob, re, sys

FAMILIES = [("conv_gemm_kernel", "isa_conv_gemm"), ("conv_gemm_tiled_kernel", "isa_conv_gemm"), ("conv3x3_tiled_kernel", "isa_conv_gemm"),
            ("conv3x3_wgrad", "isa_conv_wgrad"), ("conv_wgrad", "isa_conv_wgrad"), ("wgrad_reduce", "isa_conv_wgrad (immediate fold)"),
            ("wgrad_fold", "isa_slab_arena_flush"), ("pw_bn_bwd_kernel", "isa_conv1x1_bn_backward"),
            ("dw_bn_bwd_kernel", "isa_dwconv3x3_bn_backward"), ("dw2_fwd_kernel", "isa_dwconv3x3"), ("dw_fwd_kernel", "isa_dwconv3x3"),
            ("dw2_wgrad", "isa_dwconv3x3_dgrad/_wgrad"), ("dw_wgrad", "isa_dwconv3x3_dgrad/_wgrad"), ("dwpw_eval", "isa_dwpw_eval"),
            ("bn_bwd_kernel", "isa_bn_bwd_reduce/apply"), ("materialize_kernel", "isa_affine_act_res"), ("bn_finalize", "isa_bn_finalize"),
            ("bn_running_update", "isa_bn_running_update"), ("axpy", "isa_axpy"), ("pack_kernel", "isa_pack (param packer)"),
            ("adadelta", "isa_adadelta"), ("scale_bc", "isa_scale_bc"), ("colsum", "isa_colsum")]
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)[0]
agg, tot, ncalls = {}, 0.0, 0
for r in csv.DictReader(open(f)):
    fam = next((v for k, v in FAMILIES if k in r["Name"]), "other (attention head, losses, collate, torch fills, ...)")
    a = agg.setdefault(fam, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    tot += float(r["TotalDurationNs"]); ncalls += int(r["Calls"])
print("%-52s %8s %12s %8s %10s" % ("family", "calls", "total_ms", "share", "avg_us"))
for fam, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %8d %12.3f %7.1f%% %10.2f" % (fam, c, t / 1e6, 100 * t / tot, t / c / 1e3))
print("total kernel time %.1f ms over %d launches" % (tot / 1e6, ncalls) + (" (%.2f ms and %.0f launches per step over %g steps)" % (
    tot / 1e6 / steps, ncalls / steps, steps) if steps else ""))
