#!/usr/bin/env python3
"""bench.py -- throughput of the MVP-raymarch training hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, through the operator API the
autoencoder calls:  compute_raydirs -> (AABB build + march forward, grad mode) -> march backward
(grad zero-fill included, as in the reference's MVPRaymarch.backward).  Inputs are resident in HBM before the
timed region.  Workload at N=1 (and per GPU for N>1, weak scaling): BASELINE.json configs[1] =
"C2": 1 subject, 80 cameras, 512x512, K=4096 primitives, 8^3 RGBA slabs, fp32.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run
(one rank per GPU, RCCL); rank 0 prints ONE JSON line.  Cameras are sharded across ranks with no data-path
collective (render/march units are independent: SURVEY.md section 8e); the only collectives are the barriers
and the MAX-reduce of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (N cams per GPU, H, W, K, slab)
    "C2": (80, 512, 512, 4096, 8),
    "C3": (4, 512, 512, 16384, 8),
    "C1": (4, 128, 128, 512, 8),
    "C4": (4, 1024, 1024, 8192, 8),
}


def algorithmic_bytes(N, H, W, K, V=512):
    """SURVEY.md section 8(d) / BASELINE.md section 3: every tensor touched once, fp32."""
    R = N * H * W
    fwd = 60 * R + N * K * (V * 16 + 60) + 24 * N * (2 * K - 1)
    bwd = 60 * R + 2 * N * K * V * 16 + 120 * N * K + 24 * N * (2 * K - 1)
    return fwd, bwd


def cpu_baseline(N, H, W, K, slab, budget_s=20.0):
    """Time the fp32 CPU port (oracle/, OpenMP over rays) on a bounded sample of the same workload."""
    import numpy as np
    from ava256_amd.scene import make_scene
    from oracle.mvp_oracle import Oracle

    cores = os.cpu_count() or 1
    o = Oracle("f32")
    n_img = 1
    s = make_scene(n_img, H, W, K, device="cpu", seed=1112, slab=slab)
    npv = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in s.items()}

    def one_pass(npv, n_img):
        t0 = time.perf_counter()
        rp, rd, tm = o.raydirs(npv["campos"], npv["camrot"], npv["focal"], npv["princpt"], npv["pixelcoords"],
                               npv["volradius"])
        a = (rp, rd, npv["stepsize"], tm, npv["primpos"], npv["primrot"], npv["primscale"], npv["template"])
        rgba, sat, st = o.march_forward(*a)
        o.march_backward(*a, sat, np.ones_like(rgba))
        return time.perf_counter() - t0

    t1 = one_pass(npv, 1)
    reps = int(max(1, min(16, budget_s / max(t1, 1e-3))))
    tt = t1
    for _ in range(reps - 1):
        tt += one_pass(npv, 1)
    rays = reps * H * W
    return {"value": rays / tt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "%d x (1 camera %dx%d, K=%d: raydirs+aabb+fwd+bwd) in %.1f s, OpenMP over rays, fp32" % (
                reps, H, W, K, tt)}


def train_mode(args, rank, local_rank, world, dev, dist):
    """Reference-shaped training iterations (ava-256_amd/trainloop.py): stand-in decoder -> rays -> march fwd/bwd ->
    L1 + primvolsum -> NaN mask -> clip -> Adam, DDP all-reduce of parameter gradients when world > 1."""
    from ava256_amd import _hooks as mm
    from ava256_amd.trainloop import RaymarchTrainModel, SlabDecoderStandIn, Trainer, make_training_batch
    N, H, W, K, slab = WORKLOADS[args.workload]
    batch, volradius = make_training_batch(N, H, W, K, dev, seed=1112 + rank,
                                           target_decoder=SlabDecoderStandIn(K, slab, seed=9))
    model = RaymarchTrainModel(SlabDecoderStandIn(K, slab, seed=1), volradius).to(dev)
    nparams = sum(p.numel() for p in model.parameters())
    tr = Trainer(model, ddp=world > 1, device_ids=[local_rank] if world > 1 else None)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(batch)
    sync()
    events = []
    mm.set_event_sink(events)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    mm.set_event_sink(None)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kt = {}
    for name, a, b in events:
        kt.setdefault(name, []).append(a.elapsed_time(b))
    kavg = {k: sum(v) / len(v) for k, v in kt.items()}
    if rank == 0:
        print(json.dumps({
            "metric": "train iters/sec, raymarch training path with a stand-in decoder (NOT ava-256's conv stacks)",
            "value": args.steps / elapsed, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d frames/GPU, %dx%d, K=%d, %d^3 slabs" % (args.workload, N, H, W, K, slab),
                       "parallelism": "DDP over %d rank(s), gradients only, one %0.1f MB bucket" % (world, nparams * 4e-6)},
            "frames_per_s": N * world * args.steps / elapsed, "rays_per_s": N * H * W * world * args.steps / elapsed,
            "allreduce_mb": nparams * 4e-6, "kernel_ms": kavg, "final_loss": float(loss)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--alpha-gain", type=float, default=1.0, help="1.0 = random-init opacity (nothing saturates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="march", choices=["march", "train"],
                    help="march (default, the contract metric): the raymarch hot path; train: the reference-shaped "
                         "optimisation loop (stand-in decoder, DDP gradient all-reduce over RCCL) -> iterations/s")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ava256_amd as ops
    from ava256_amd import _hooks as mm
    from ava256_amd.scene import make_scene

    if args.mode == "train":
        train_mode(args, rank, local_rank, world, dev, dist)
        return

    N, H, W, K, slab = WORKLOADS[args.workload]
    s = make_scene(N, H, W, K, device=dev, seed=1112 + rank, alpha_gain=args.alpha_gain, slab=slab)
    prim_names = ("primpos", "primrot", "primscale", "template")
    for k in prim_names:
        s[k].requires_grad_(True)
    torch.manual_seed(5 + rank)
    gout = torch.randn(N, H, W, 4, device=dev)
    volradius, stepsize = s["volradius"], s["stepsize"]

    def step():
        for k in prim_names:
            s[k].grad = None
        raypos, raydir, tminmax = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"],
                                                      s["pixelcoords"], volradius)
        rgba = ops.mvpraymarch(raypos, raydir, stepsize, tminmax, (s["primpos"], s["primrot"], s["primscale"]),
                               s["template"], None)
        rgba.backward(gout)
        return rgba

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    events = []
    mm.set_event_sink(events)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    mm.set_event_sink(None)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel average launch durations from the HIP events recorded inside the timed region
    kt = {}
    for name, a, b in events:
        kt.setdefault(name, []).append(a.elapsed_time(b))
    kavg = {k: sum(v) / len(v) for k, v in kt.items()}

    if rank == 0:
        rays_per_step = N * H * W * world
        bf, bb = algorithmic_bytes(N, H, W, K, slab ** 3)
        dom = "march_backward" if kavg.get("march_backward", 0) >= kavg.get("march_forward", 0) else "march_forward"
        dom_bytes = bb if dom == "march_backward" else bf
        dom_ms = kavg.get(dom, float("nan"))
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch HBM bytes from rocprofv3 PMC passes
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.workload, {}).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "rendered rays/sec, MVP raymarch training hot path (raydirs + AABB + march fwd + march bwd), "
                      "80-cam 512x512 K=4096 per GPU",
            "value": rays_per_step * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded shell scene, random-init slab statistics; SURVEY.md 8d)",
            "config": {"workload": "%s: %d cams/GPU, %dx%d, K=%d primitives, %d^3 RGBA slabs, fadeexp=8, dt=1/256" % (
                args.workload, N, H, W, K, slab), "alpha_gain": args.alpha_gain,
                "parallelism": "cameras sharded over %d rank(s), no data-path collective" % world},
            "iters_per_s": args.steps / elapsed,
            "fwd_rays_per_s": (N * H * W) / (kavg["march_forward"] * 1e-3) if "march_forward" in kavg else None,
            "kernel_ms": kavg,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                         "fwd": {"achieved": bf / (kavg.get("march_forward", float("nan")) * 1e-3) / 1e9,
                                 "algorithmic_bytes_per_launch": bf,
                                 "avg_launch_ms": kavg.get("march_forward")}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, H, W, K, slab)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
