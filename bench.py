#!/usr/bin/env python3
"""Headline benchmark: images/sec at 256x256, bs=16 per GPU, synthetic data, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` = whole-job images/s with
inputs resident in HBM.  Workloads:
  train_step : forward + backward + grad all-reduce (N>1) + fused Adadelta   [default when built]
  train_fwd  : ReSeg.forward(True, x, sem, ins, N) — conv backbone + attention mask head, 2 iterations
  infer      : ReSeg(False, x) sem-only inference (configs[2])
Extra objects: `roofline` (dominant kernel family, HIP-event timed in an instrumented extra step on
the launch stream; algorithmic bytes = input read once + output written once per conv)
and `cpu_baseline` (the CPU oracle = port of the reference path, timed on the host cores on a
bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT]

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured achievable)
# SURVEY.md §8(d) contract figures, bytes per image at 256^2 in bf16 (x2 for fp32 storage)
ALG_BYTES_PER_IMAGE_BF16 = {"infer": 161.7e6, "train_fwd": 1.071e9, "train_step": 3.2e9}


def load_traffic(entry_point):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_traffic.json,
    collected with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs and the gfx950 x2 FETCH
    correction of MI355X_MICROARCH.md).  null when no committed measurement covers that kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if entry_point in d:
                return d[entry_point]["hbm_bytes_per_launch"]
        except Exception:
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="auto", choices=["auto", "train_step", "train_fwd", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    def log(msg):
        print("[bench] " + msg, file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    import isa_amd  # noqa: F401
    from isa_amd.reseg import ReSeg
    from isa_amd.data import synth_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    # ISA_DIST_REHEARSAL=1: all ranks share the visible GPU(s) and exchange over gloo - the only way to execute the
    # world > 1 branch (graph for forward+backward, eager exchange + update) on a one-GPU box; never a measurement
    rehearsal = os.environ.get("ISA_DIST_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    workload = args.workload
    try:
        from isa_amd.trainer import Trainer
        have_trainer = True
    except ImportError:
        have_trainer = False
    if workload == "auto":
        workload = "train_step" if have_trainer else "train_fwd"

    use_ins = workload != "infer"
    model = ReSeg(2, use_ins, dtype=dtype)
    model.reset_parameters(seed=23)                              # random-init weights (deterministic)
    B, S = args.batch, args.size
    x, sem, ins, n = synth_batch(B, S, S, seed=100 + rank)       # per-rank shard of the global batch
    x, sem, ins = x.cuda(), sem.cuda(), ins.cuda()
    sel = [list(range(int(k))) for k in n.view(-1)]

    trainer = None
    if workload == "train_step":
        model.train()
        trainer = Trainer(model, world_size=world)

        def eager_step():
            trainer.train_step(x, sem, ins, n, selected_idx=sel)

        def step():
            if args.no_graph:
                eager_step()
            else:               # hipGraph replay of the same launches (trainer.py: train_step_graphed)
                trainer.train_step_graphed(x, sem, ins, n, selected_idx=sel)
    elif workload == "train_fwd":
        model.train()
        trainer = Trainer(model, world_size=world)

        def step():
            if args.no_graph:
                model(True, x, sem, ins, n, selected_idx=sel)
            else:               # the training-mode forward alone (batch statistics, sampling, Dropout2d, losses), graph-replayed
                trainer.train_step_graphed(x, sem, ins, n, selected_idx=sel, forward_only=True)
    else:
        model.eval()

        def step():
            if args.no_graph:
                model(False, x)
            else:
                model.infer_graphed(x)      # hipGraph replay of the same launches (reseg.py)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if workload in ("train_step", "train_fwd", "infer") and not args.no_graph:
        log("graph setup: one eager step + one capture step (untimed, before the warmup steps)")
        step()
        step()
        sync_all()
        if workload in ("train_step", "train_fwd") and trainer.static_inputs():
            # the synthetic batch is resident in HBM: place it in the graph's own input buffers once, so that a step
            # is exactly the replay (a real pipeline writes each new batch into these buffers)
            gx_, gsem_, gins_ = trainer.static_inputs()[0]
            gx_.copy_(x); gsem_.copy_(sem); gins_.copy_(ins)
            x, sem, ins = gx_, gsem_, gins_
            sync_all()
    if workload == "train_step" and model.engine.fold_stats[0]:
        log("weight-gradient folds postponed to the end of the backward pass: %d per step, %.0f MB of slab arena"
            % (model.engine.fold_stats[0], model.engine.fold_stats[1] * 4 / 1e6))
    log("workload=%s dtype=%s world=%d: warmup" % (workload, args.dtype, world))
    for _ in range(args.warmup):
        step()
    sync_all()
    log("timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    log("%.2f ms/step, %.1f img/s" % (ms, value))

    out = {
        "metric": "images/sec @%dx%d bs=%d per GPU" % (S, S, B),
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s %dx%d bs=%d/GPU (%s)" % (
            {"train_step": "train.py step fwd+bwd+update", "train_fwd": "ReSeg.forward(training) conv+attention head",
             "infer": "pred_list batched inference"}[workload], S, S, B, args.dtype),
            "global_batch": world * B, "image": [S, S], "parallelism": "dp%d" % world,
            "launch": "eager" if args.no_graph else "hipGraph replay"},
    }

    if rank == 0:
        # ---- roofline: one extra instrumented step, events on the launch stream -----------------
        E = model.engine
        E.profile = True
        # one stream for this step: with the decoder iterations overlapped on two streams an event pair around a launch
        # also times whatever the other stream runs meanwhile (26.3 us per conv_gemm launch against 20.3 us in the rocprofv3
        # trace of the same command)
        head = getattr(model, "head", None)
        saved_streams = getattr(head, "streams", None)
        if saved_streams is not None:
            head.streams = 1
        if workload == "train_step":                              # per-launch events need the eager launch loop
            # forward + backward only: this block runs on rank 0 alone, so it must stay collective-free (the update's
            # all-reduce would wait forever for the other ranks)
            trainer.forward_backward(x, sem, ins, n, selected_idx=sel)
        elif workload == "infer":
            model(False, x)
        else:
            trainer.forward_backward(x, sem, ins, n, selected_idx=sel, backward=False)
        prof = E.profile_summary()
        E.profile = False
        if saved_streams is not None:
            head.streams = saved_streams
        fam = {k: v for k, v in prof.items() if v[2] > 0}
        if fam:
            dom = max(fam.items(), key=lambda kv: kv[1][1])
            name, (calls, tot_ms, nbytes) = dom
            achieved = nbytes / (tot_ms * 1e-3) / 1e9
            esz = 2 if args.dtype == "bf16" else 4
            step_bytes = ALG_BYTES_PER_IMAGE_BF16[workload] * (esz / 2) * (S * S) / 65536.0
            out["roofline"] = {
                "bound": "hbm", "kernel": name, "launches_per_step": calls,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(name),
                "avg_launch_us": round(tot_ms * 1e3 / calls, 2),
                "alg_bytes_per_launch": int(nbytes / calls),
                "step_frac": round(value / world * step_bytes / 1e9 / HBM_PEAK_GBS, 4),
                "launches_per_step_all": sum(v[0] for v in prof.values()),
                "families_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]},
            }
        # ---- CPU baseline: the oracle (port of the reference path) on a bounded sample ----------
        if world == 1 and not args.no_cpu_baseline:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            # the ONLY place the oracle is used: the CPU port of the reference path, timed as a baseline
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import reseg_ref as R
            torch.set_num_threads(max(1, min(ncpu, 16)))      # the GPU box gives one GPU a 16-core share
            log("cpu baseline on %d threads" % torch.get_num_threads())
            # bounded sample: the oracle's train step costs ~2 s per image on 16 host threads, so the benchmark batch
            # (16 images) would take half a minute per repetition; bs=2 and at least 5 repetitions (median) keep the
            # default run within minutes and the number stable.  The per-image CPU rate is flat in the batch size (the
            # reference itself, profiles/r02_reference_cpu_timing.json: bs=4 vs bs=8)
            cb = 2 if workload != "infer" else 16
            cx, csem, cins, cn = R.synth_batch(cb, S, S, seed=7)
            sd = R.synth_state_dict(23, True)
            csel = [list(range(int(k))) for k in cn.view(-1)]
            pick = (lambda a: a.argmax(1))

            def cpu_step():
                if workload == "infer":
                    with torch.no_grad():
                        R.reseg_forward(sd, cx, use_instance_seg=False)
                elif workload == "train_fwd":
                    with torch.no_grad():
                        R.reseg_forward(sd, cx, csem, cins, cn, ctx=R.Ctx(bn_train=True, training=True, drop_rate=0.0),
                                        state=R.HeadState(), selected_idx=csel, sample_fn=pick)
                else:
                    P = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
                         for k, v in sd.items()}
                    o = R.reseg_forward(P, cx, csem, cins, cn, ctx=R.Ctx(bn_train=True, training=True, drop_rate=0.0),
                                        state=R.HeadState(), selected_idx=csel, sample_fn=pick)
                    ce, dice = R.sem_losses(o["sem_out"], csem)
                    (o["ins_cost_finite"] + ce + dice).backward()

            cpu_step()
            log("cpu baseline warm rep done")
            times, t1 = [], time.perf_counter()
            while len(times) < 5 or (time.perf_counter() - t1 < 20.0 and len(times) < 20):
                t2 = time.perf_counter()
                cpu_step()
                times.append(time.perf_counter() - t2)
            times.sort()
            cdt, reps = times[len(times) // 2], len(times)
            out["cpu_baseline"] = {"value": round(cb / cdt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
                                   "kind": "port", "sample": "median of %d reps of the same workload at bs=%d %dx%d fp32 (oracle; "
                                   "fastest %.3f, slowest %.3f images/s)" % (reps, cb, S, S, cb / times[0], cb / times[-1])}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
