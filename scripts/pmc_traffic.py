#!/usr/bin/env python3
"""HBM bytes per launch by kernel family from two rocprofv3 counter passes (csv output):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d A --output-format csv -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d B --output-format csv -- python bench.py ...
    python scripts/pmc_traffic.py A B profiles/rNN_traffic.json
bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section); calibration on a known-size launch: profiles/r01_pmc_calibration_conv1x1.txt."""
import csv, glob, json, re, sys

FAMILIES = [("conv_gemm_kernel", "isa_conv_gemm"), ("conv_gemm_tiled_kernel", "isa_conv_gemm"), ("conv3x3_tiled_kernel", "isa_conv_gemm"), ("conv_wgrad", "isa_conv_wgrad"), ("wgrad_reduce", "isa_conv_wgrad(reduce)"), ("wgrad_fold", "isa_wgrad_defer_flush"),
            ("pw_bn_bwd_kernel", "isa_conv1x1_bn_backward"), ("dw_bn_bwd_kernel", "isa_dwconv3x3_bn_backward"),
            ("dw2_fwd_kernel", "isa_dwconv3x3"), ("dw_fwd_kernel", "isa_dwconv3x3"), ("bn_bwd_kernel", "isa_bn_bwd"),
            ("materialize_kernel", "isa_affine_act_res"), ("axpy", "isa_axpy")]


def load(path, counter):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        fam = next((v for k, v in FAMILIES if k in r["Kernel_Name"]), None)
        if fam is None:
            continue
        a = agg.setdefault(fam, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in sorted(set(fetch) & set(write)):
    n = min(fetch[fam][0], write[fam][0])
    fa, wa = fetch[fam][1] / fetch[fam][0], write[fam][1] / write[fam][0]
    out[fam] = {"launches_profiled": n, "fetch_size_kb_raw_avg": round(fa, 1), "write_size_kb_avg": round(wa, 1),
                "hbm_bytes_per_launch": int((2 * fa + wa) * 1024),
                "note": "2*FETCH_SIZE + WRITE_SIZE (KB): gfx950 FETCH_SIZE counts half of wide coalesced reads "
                        "(MI355X_MICROARCH.md, HBM); calibrated 1.002x/1.009x on a known-size conv_gemm launch"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print("%-32s launches %6d  %10.1f MB/launch" % (k, v["launches_profiled"], v["hbm_bytes_per_launch"] / 1e6))
