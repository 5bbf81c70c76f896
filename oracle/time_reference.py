"""Time the UPSTREAM reference network itself on this container's CPU cores: `reseg.ReSeg` imported
through oracle/ref_shim.py, fp32, torch CPU, synthetic inputs, >= 5 repetitions, median.  The entry scripts cannot run
as shipped, so the baseline is defined at the ReSeg.forward boundary.  Test / measurement infrastructure:
needs /root/reference, never travels to the GPU box.    python oracle/time_reference.py [out.json]"""
import contextlib
import io
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim          # noqa: E402
import reseg_ref as R    # noqa: E402
from gen_golden import build  # noqa: E402


def med(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts), max(ts)


def main():
    threads = int(os.environ.get("REF_THREADS", "8"))
    torch.set_num_threads(threads)
    reseg, config = ref_shim.install()
    out = dict(threads=threads, torch=torch.__version__, size=256, reps=5, rows=[])
    for bs in (1, 16):
        m = build(reseg, config, 256, False, torch.float32).eval()
        x, _, _, _ = R.synth_batch(bs, 256, 256, seed=1)

        def infer():
            with torch.no_grad():
                m(False, x)
        t, lo, hi = med(infer, 5)
        out["rows"].append(dict(workload="ReSeg(False, x) sem-only inference", batch=bs, median_s=t, min_s=lo, max_s=hi,
                                images_per_s=bs / t))
        print(out["rows"][-1], flush=True)
    for bs in (4, 8):
        m = build(reseg, config, 256, True, torch.float32).train()
        x, sem, ins, n = R.synth_batch(bs, 256, 256, seed=1)
        order = [list(range(int(k))) for k in n.view(-1)]
        m.decoder.getRandomIdx = lambda n_ins: [list(s) for s in order]

        def step():
            with contextlib.redirect_stdout(io.StringIO()):
                o = m(True, x, sem, ins, n)
                ce, dice = R.sem_losses(o[0], sem)
                m.zero_grad()
                (o[2] + ce + dice).backward()
        t, lo, hi = med(step, 5)
        out["rows"].append(dict(workload="train step forward+backward (2 decoder iterations, no optimizer)", batch=bs,
                                median_s=t, min_s=lo, max_s=hi, images_per_s=bs / t))
        print(out["rows"][-1], flush=True)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), "profiles", "r02_reference_cpu_timing.json")
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
