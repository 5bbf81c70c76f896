#!/usr/bin/env python3
"""Print per-kernel register/LDS/occupancy usage of a .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    if flt and not re.search(flt, r["name"]): continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:90]
    print("%-90s VGPR %4s AGPR %3s spill %3s scratch %5s occ %2s LDS %6s" % (
        name, r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
        r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
