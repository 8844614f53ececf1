#!/usr/bin/env python3
"""bench.py -- throughput of the MVP-raymarch training hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, through the operator API the
autoencoder calls:  compute_raydirs -> (AABB build + march forward, grad mode) -> march backward
(grad zero-fill included, as in the reference's MVPRaymarch.backward).  Inputs are resident in HBM before the
timed region.  Workload at N=1 (and per GPU for N>1, weak scaling): BASELINE.json configs[1] =
"C2": 1 subject, 80 cameras, 512x512, K=4096 primitives, 8^3 RGBA slabs, fp32.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 either launched by torch.distributed.run
(one rank per GPU, RCCL) or, when no launcher set WORLD_SIZE, starting its own N ranks (`self_launch`); rank 0
prints ONE JSON line.  Every rank renders its own 80-camera scene (weak
scaling; `--scaling strong` shards ONE scene's cameras over the ranks instead); there is no data-path
collective (render/march units are independent: SURVEY.md section 8e) -- the only collectives of the march
leg are the barriers and the MAX-reduce of the elapsed time.

The same line carries a `train` object: iterations/s of the reference-shaped optimisation loop
(ava-256_amd/trainloop.py: VAE bottleneck -> stand-in decoder with a geometry branch (guide mesh, adaptwarps running
average, the schedule of the first 100 iterations) -> rays -> march -> colour calibration -> bf16 background MLP ->
matting -> irgbl1 + vertl1 + primvolsum + kldiv -> backward -> NaN mask -> clip -> Adam; DDP all-reduce of parameter
gradients over RCCL when N>1), measured after the march leg on the same process group.

The control flow (init, warm-up, barrier + synchronize, timed steps, MAX over ranks, rank-0 line) lives in
`run_timed` / `main` and takes the process-group backend and the step factory as arguments, so that
tests/test_bench_multirank.py can drive exactly this code with 2 gloo ranks on CPU (the kernels replaced by the
CPU checker THERE, never here).
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

# The reference's own code timed on CPU.  Its Python may not travel to the GPU box, so these are measurements taken where it
# is mounted (the build container) by tools/measure_reference_cpu.py, recorded with the commit, date and core count in
# profiles/reference_cpu.json -- carried beside the live `cpu_baseline` (this build's CPU port on the GPU box's cores).
def reference_cpu():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "reference_cpu.json")))
    except Exception as e:
        return {"error": "profiles/reference_cpu.json not readable: %s" % e}


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


def render_bytes(N, H, W, K, V=512, voxel_bytes=16):
    """Algorithmic bytes of one no-grad render with the rays made inside the march (no ray tensors, rgba out only):
    16 B per ray written, every slab voxel (16 B fp32 RGBA, 8 B fp16 RGBA) and pose record read once, the node boxes once."""
    return 16 * N * H * W + N * K * (V * voxel_bytes + 60) + 24 * N * (2 * K - 1)


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo: SMT siblings counted once (SURVEY.md 8d asks for the physical count
    next to the CPU timing); None when the file does not say."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(N, H, W, K, slab, budget_s=20.0, cams=1):
    """Time the fp32 CPU port (oracle/, OpenMP over rays) on a bounded sample of the same workload (`cams` cameras of it per
    pass)."""
    import numpy as np
    from ava256_amd.scene import make_scene
    from oracle.mvp_oracle import Oracle

    o = Oracle("f32")
    threads = o.max_threads()                 # what the OpenMP loops of the port run on (OMP_NUM_THREADS or every logical CPU)
    logical, physical = os.cpu_count() or 1, physical_cores()
    s = make_scene(cams, H, W, K, device="cpu", seed=1112, slab=slab)
    npv = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in s.items()}

    def one_pass():
        t0 = time.perf_counter()
        rp, rd, tm = o.raydirs(npv["campos"], npv["camrot"], npv["focal"], npv["princpt"], npv["pixelcoords"],
                               npv["volradius"])
        a = (rp, rd, npv["stepsize"], tm, npv["primpos"], npv["primrot"], npv["primscale"], npv["template"])
        rgba, sat, st = o.march_forward(*a)
        o.march_backward(*a, sat, np.ones_like(rgba))
        return time.perf_counter() - t0

    t1 = one_pass()
    reps = int(max(1, min(16, budget_s / max(t1, 1e-3))))
    tt = t1
    for _ in range(reps - 1):
        tt += one_pass()
    rays = reps * cams * H * W
    return {"value": rays / tt, "unit": "rays/s", "cores": threads, "kind": "port",
            "omp_max_threads": threads, "logical_cpus": logical, "physical_cores": physical,
            "sample": "%d x (%d camera(s) %dx%d, K=%d: raydirs+aabb+fwd+bwd) in %.1f s, OpenMP over rays (%d threads on %s "
                      "physical cores / %d logical CPUs), fp32" % (reps, cams, H, W, K, tt, threads, physical or "?", logical)}


# ---------------------------------------------------------------------------------------------------------------
# control flow shared by every leg (and by the 2-rank gloo test)
# ---------------------------------------------------------------------------------------------------------------
def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, rank, world, dev, force=False):
    """One process per GPU (backend "nccl" = RCCL) or per CPU rank (backend "gloo", tests).  None when world == 1 --
    unless `force` (--dist-smoke): then a one-rank group is made, so that every collective of the N > 1 path (barrier,
    MAX-reduce, DDP's bucket all-reduce, the adaptwarps MAX) runs through RCCL on a single GPU."""
    if world == 1 and not force:
        return None
    import torch.distributed as dist
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def run_timed(step, steps, warmup, dist, dev):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device synchronize on both sides;
    returns the elapsed wall time, MAX over ranks."""
    def sync():
        if dist is not None:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def camera_shard(args, rank, world):
    """Cameras this rank renders: its own N-camera scene (weak scaling, the default: per-GPU work fixed) or a
    contiguous shard of ONE N-camera scene (strong scaling: total work fixed)."""
    from ava256_amd.dist_util import shard_range
    N = WORKLOADS[args.workload][0] if args.cams is None else args.cams
    if args.scaling == "strong":
        lo, hi = shard_range(N, rank, world)
        return N, lo, hi, 1112          # same seed everywhere: one scene, sharded
    return N, 0, N, 1112 + rank          # one scene per rank


# ---------------------------------------------------------------------------------------------------------------
# the two GPU legs
# ---------------------------------------------------------------------------------------------------------------
def l1_matting_gradient(rgba, bg=60.0, shift=3):
    """d loss / d rayrgba of an L1 image loss behind the matting of the decode tail (models/autoencoder.py:254-269: irgbrec =
    rayrgb + (1 - rayalpha) * bg; ddp-train.py:408 / losses.py:12-14: mean |irgbrec - image|), against a target that is the
    same image shifted by `shift` pixels: what a TRAINING iteration hands the march's backward -- sign-valued (+-c per colour
    channel, exactly 0 where prediction and target agree: the flat background), an alpha channel that is a signed sum of those,
    nothing like the Gaussian of the headline step."""
    x = rgba.detach().clone().requires_grad_(True)       # [N,H,W,4]
    pred = x[..., :3] + (1.0 - x[..., 3:4]) * bg
    with torch.no_grad():
        target = torch.roll(pred, shifts=(shift, shift), dims=(1, 2))
    (pred - target).abs().mean().backward()
    return x.grad.detach()


def make_march_step_gpu(args, rank, world, dev):
    """The hot path through the operator API on `dev`.  Returns (step, info)."""
    import ava256_amd as ops
    from ava256_amd.scene import make_scene
    _, H, W, K, slab = WORKLOADS[args.workload]
    N, lo, hi, seed = camera_shard(args, rank, world)
    s = make_scene(N, H, W, K, device=dev, seed=seed, alpha_gain=args.alpha_gain, slab=slab)
    prim_names = ("primpos", "primrot", "primscale", "template")
    cam_names = ("campos", "camrot", "focal", "princpt", "pixelcoords")
    for k in prim_names + cam_names:
        s[k] = s[k][lo:hi].contiguous()
    for k in prim_names:
        s[k].requires_grad_(True)
    n_local = hi - lo
    torch.manual_seed(5 + rank)
    gout = torch.randn(n_local, H, W, 4, device=dev)
    volradius, stepsize = s["volradius"], s["stepsize"]
    if getattr(args, "gout", "randn") == "l1_matting":   # the upstream gradient of a training iteration (see l1_matting_gradient)
        import ava256_amd as ops_
        with torch.no_grad():
            rp_, rd_, tm_ = ops_.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], volradius)
            rgba0 = ops_.mvpraymarch(rp_, rd_, stepsize, tm_, (s["primpos"], s["primrot"], s["primscale"]), s["template"], None)
        gout = l1_matting_gradient(rgba0)
        del rp_, rd_, tm_, rgba0

    def step():
        for k in prim_names:
            s[k].grad = None
        raypos, raydir, tminmax = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"],
                                                      s["pixelcoords"], volradius)
        rgba = ops.mvpraymarch(raypos, raydir, stepsize, tminmax, (s["primpos"], s["primrot"], s["primscale"]),
                               s["template"], None)
        rgba.backward(gout)
        return rgba

    def fused_step():  # the same training step with the rays made inside the forward march (row N1): no raydirs launch
        for k in prim_names:
            s[k].grad = None
        rgba = ops.mvpraymarch_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"],
                                            volradius, stepsize, (s["primpos"], s["primrot"], s["primscale"]),
                                            s["template"])
        rgba.backward(gout)
        return rgba

    def render():  # inference: no gradients, rays made inside the march (row N1), nothing handed to a backward
        with torch.no_grad():
            return ops.mvpraymarch_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"],
                                                volradius, stepsize, (s["primpos"], s["primrot"], s["primscale"]),
                                                s["template"])

    half = {}

    def render_half():  # the opt-in render path over fp16 slabs (ava-256_amd/halfslab.py); conversion outside the timing
        from ava256_amd import halfslab
        if "t" not in half:
            with torch.no_grad():
                half["t"] = halfslab.template_to_half(s["template"])
        return halfslab.render_half_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"],
                                                 volradius, stepsize, (s["primpos"], s["primrot"], s["primscale"]),
                                                 half["t"])

    def to_half():
        from ava256_amd import halfslab
        with torch.no_grad():
            return halfslab.template_to_half(s["template"])

    def hit_packets():  # 8x8 ray packets with a non-empty hit list (kernel diagnostics of one untimed forward)
        from ava256_amd import _hooks
        diag = torch.zeros(8, dtype=torch.int32, device=dev)
        _hooks.set_diag_buffer(diag)
        try:
            render()
            torch.cuda.synchronize(dev)
            return int(_hooks.read_diag()["packets_hit"])
        finally:
            _hooks.set_diag_buffer(None)

    def marks():  # what the last backward left to its other kernels: primitives on the two-pass / on the ray-centric kernel
        from ava256_amd import _hooks
        keep = _hooks.keep_raysat
        _hooks.keep_raysat = True
        try:
            step()
            torch.cuda.synchronize(dev)
            c = _hooks.last_pl_count[: n_local * K]
            return {"two_pass_primitives": int(((c >> 30) & 1).sum()), "ray_centric_primitives": int(((c >> 31) & 1).sum()),
                    "primitives": n_local * K}
        finally:
            _hooks.keep_raysat, _hooks.last_raysat, _hooks.last_pl_count = keep, None, None

    return step, {"n_local": n_local, "H": H, "W": W, "K": K, "slab": slab, "N": N, "render": render, "marks": marks,
                  "fused_step": fused_step, "hit_packets_fn": hit_packets, "render_half": render_half, "to_half": to_half}


def kernel_averages(events):
    kt = {}
    for name, a, b in events:
        kt.setdefault(name, []).append(a.elapsed_time(b))
    return {k: sum(v) / len(v) for k, v in kt.items()}


def time_calls(fn, dev, reps=5, warm=2):
    """Average wall time (ms, device events on the current stream) of `reps` calls after `warm` untimed ones."""
    for _ in range(warm):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize(dev)
    return ev0.elapsed_time(ev1) / reps


def recorded_traffic(key):
    """Per-launch HBM bytes of the march kernels of one workload from the RECORDED counter passes (profiles/traffic.json,
    tools/make_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes), with the commit they were measured at -- or (None, None)."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return doc.get(key), doc.get("_by_workload", {}).get(key, {}).get("measured_at_commit", doc.get("_measured_at_commit"))
    except Exception:
        return None, None


def roofline_of(kavg, n_local, H, W, K, slab, traffic_key=None):
    """Both march kernels of one workload against the HBM peak: algorithmic bytes per launch / HIP-event launch time; `traffic`
    = the recorded PMC bytes of that workload (traffic_key), when a pass exists."""
    bf, bb = algorithmic_bytes(n_local, H, W, K, slab ** 3)
    rec, at = recorded_traffic(traffic_key) if traffic_key else (None, None)
    out = {}
    for name, nbytes in (("march_forward", bf), ("march_backward", bb)):
        ms = kavg.get(name)
        if ms:
            ach = nbytes / (ms * 1e-3) / 1e9
            tr = (rec or {}).get(name)
            out[name] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "traffic": tr,
                         "traffic_ratio": (tr / nbytes) if tr else None, "traffic_measured_at_commit": at if tr else None}
    return out


def march_leg(args, rank, world, dev, workload, alpha_gain, steps=10, warmup=3, gout="randn", traffic_key=None):
    """One more march workload on this rank, timed like the headline (same step through the operator API, HIP events per
    launch): SURVEY.md 8(d)'s secondary runs -- C3 / C4 at their per-GPU batch, and the "trained-like" scene (opacity x 40:
    about half the rays saturate, early termination primaccum.h:63-79, mvpraymarch_subset_kernel.h:76-97)."""
    from ava256_amd import _hooks as mm
    a = argparse.Namespace(**vars(args))
    a.workload, a.alpha_gain, a.cams, a.scaling, a.gout = workload, alpha_gain, None, "weak", gout
    step, info = make_march_step_gpu(a, rank, world, dev)
    for _ in range(warmup):
        step()
    events = []
    mm.set_event_sink(events)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        out = step()
    ev1.record()
    torch.cuda.synchronize(dev)
    mm.set_event_sink(None)
    ms = ev0.elapsed_time(ev1) / steps
    kavg = kernel_averages(events)
    N, H, W, K, slab = info["n_local"], info["H"], info["W"], info["K"], info["slab"]
    res = {"workload": "%s: %d cams/GPU, %dx%d, K=%d, alpha_gain %g" % (workload, N, H, W, K, alpha_gain),
           "ms_per_step": ms, "rays_per_s": N * H * W / (ms * 1e-3), "steps": steps, "kernel_ms": kavg,
           "saturated_ray_fraction": float((out.detach()[..., 3] >= 1.0 - 1e-6).float().mean()),
           "roofline": roofline_of(kavg, N, H, W, K, slab, traffic_key)}
    if gout != "randn":
        res["upstream_gradient"] = ("d(mean L1 of rgb + (1 - alpha) * bg against the image shifted by 3 pixels) / d rayrgba: "
                                    "sign-valued, zero on the flat background -- bench.l1_matting_gradient")
        res["backward_marks"] = info["marks"]()
        res["kernel_ms_note"] = ("march_backward = the whole mvp_march_backward call: bound prologue + bwd_prim_kernel + the "
                                 "two-pass kernel + the ray-centric kernel (HIP events around the call)")
    del step, info, out
    torch.cuda.empty_cache()
    return res


def collective_identity(dist, dev, world):
    """What the first multi-GPU record should answer from the line itself: did the process group
    see N ranks, which device is each rank on, and what bus bandwidth does ONE flat all-reduce of ava-256's gradient size
    (46.87 M fp32 parameters = 187.5 MB, ddp-train.py:312: DDP over the whole model) reach: busbw = 2 (n-1)/n * bytes / t."""
    name, bus = "cpu", None
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        name = pr.name
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    me = {"rank": dist.get_rank(), "device": str(dev), "name": name, "pci": bus, "pid": os.getpid()}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    numel = 46_870_000 if dev.type == "cuda" else 1 << 18      # (the CPU test keeps it small)
    buf = torch.ones(numel, device=dev, dtype=torch.float32)
    reps = 10 if dev.type == "cuda" else 2
    for _ in range(2):
        dist.all_reduce(buf)
    buf.fill_(1.0)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t = torch.tensor([(time.perf_counter() - t0) / reps], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sec = float(t.item())
    nbytes = numel * 4
    ok = bool(abs(float(buf[0].item()) - float(world) ** reps) <= 1e-3 * float(world) ** reps)
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks,
            "allreduce_bytes": nbytes, "allreduce_ms": sec * 1e3, "allreduce_algbw_gbs": nbytes / sec / 1e9,
            "allreduce_busbw_gbs": 2.0 * (world - 1) / max(world, 1) * nbytes / sec / 1e9, "allreduce_sums_ok": ok,
            "what": "one flat fp32 all_reduce of %d elements (ava-256's 187.5 MB gradient set when on GPUs), %d timed "
                    "repetitions, MAX over ranks; busbw = 2 (n-1)/n * bytes / t" % (numel, reps)}


def train_leg(workload, steps, warmup, rank, local_rank, world, dev, dist, with_bg, ddp=None, config=None, graph=False,
              bucket_mb=None):
    """Reference-shaped training iterations (ava-256_amd/trainloop.py) -> dict for the `train` object.  `config` =
    config.load_train_config(...) of one of the reference's YAML files: its batch size per GPU and hyper-parameters."""
    from ava256_amd import _hooks as mm
    from ava256_amd.trainloop import (BackgroundMLPStandIn, CodeEncoderStandIn, ColorCalStandIn, RaymarchTrainModel,
                                      SlabDecoderStandIn, Trainer, make_training_batch)
    N, H, W, K, slab = WORKLOADS[workload]
    if config is not None:
        N = config["batchsize"]                     # train.batchsize: frames per GPU (ddp-train.py:321)
    ncams, nident = 80, 4
    batch, volradius = make_training_batch(N, H, W, K, dev, seed=1112 + rank, ncams=ncams, nident=nident,
                                           target_decoder=SlabDecoderStandIn(K, slab, seed=9))
    model = RaymarchTrainModel(SlabDecoderStandIn(K, slab, seed=1), volradius, colorcal=ColorCalStandIn(ncams, nident),
                               bgmodel=BackgroundMLPStandIn(ncams, nident) if with_bg else None,
                               encoder=CodeEncoderStandIn()).to(dev)
    nparams = sum(p.numel() for p in model.parameters())
    ddp = (world > 1) if ddp is None else ddp
    kw = {} if bucket_mb is None else {"bucket_cap_mb": int(bucket_mb)}   # DDP bucket size (default: one flat 256 MB bucket)
    if config is not None:
        tr = Trainer.from_config(model, config, ddp=ddp, device_ids=[local_rank] if ddp else None, graph=graph, **kw)
    else:
        tr = Trainer(model, ddp=ddp, device_ids=[local_rank] if ddp else None, graph=graph, **kw)
    state = {}

    def step():
        state["loss"], _ = tr.step(batch)

    events = []
    for _ in range(warmup):
        step()
    mm.set_event_sink(events)
    elapsed = run_timed(step, steps, 0, dist, dev)
    mm.set_event_sink(None)
    px = N * H * W
    out = {"workload": "%s: %d frames/GPU, %dx%d, K=%d" % (workload, N, H, W, K), "iters_per_s": steps / elapsed,
           "ms_per_iter": 1e3 * elapsed / steps, "steps": steps, "frames_per_s": N * world * steps / elapsed,
           "allreduce_mb": nparams * 4e-6 if ddp else 0.0, "param_mb": nparams * 4e-6,
           "ddp_bucket_cap_mb": (int(bucket_mb) if bucket_mb is not None else 256) if ddp else None,
           "kernel_ms": kernel_averages(events), "final_loss": float(state["loss"]),
           "launch": ("one hipGraph replay per iteration (Trainer(graph=True): %d of the %d timed iterations)"
                      % (min(tr.graph_replays, steps), steps)) if tr.graph else "eager (one launch per kernel)",
           "hyper_parameters": {"lr": tr._lr0, "clip": tr.clip,
                                "loss_weights": tr.loss_weights, "from": "config file" if config else "configs/config.yaml values"},
           "background_mlp": ("fused MFMA kernels (csrc/bgmlp.hip: bf16 operands, fp32 accumulation), %.1f GFLOP fwd per "
                              "iteration" % (px * 2 * (120 * 256 + 4 * 256 * 256 + 256 * 3) * 1e-9))
           if with_bg else "off (matting over a constant background)"}
    if with_bg:  # MFMA utilisation of the dense kernels of this leg: a RECORDED counter pass (tools/make_mfma.py), not this run
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            rec = doc.get("mfma", {}).get("C2_bg" if workload == "C2" else workload)
            if rec:
                out["mfma_frac"] = dict(rec, _what="SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles of the kernel, against the nominal "
                                                   "peak; recorded pass of tools/evidence.sh", _source=rec.get("_source", doc.get("_mfma_source")))
        except Exception:
            pass
    del tr, model, batch
    torch.cuda.empty_cache()
    return out


def _spawned_rank(rank, world, port, argv, backend, make_step, device, errdir):
    """Entry point of one self-launched rank (see self_launch): the environment torch.distributed.run would have set.
    stderr (file descriptor 2, so that RCCL's own messages are included) goes to a per-rank file the parent reads."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    os.environ.setdefault("NCCL_DEBUG", "WARN")               # a failed rendezvous / transport says why
    if errdir:
        fd = os.open(os.path.join(errdir, "rank%d.err" % rank), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        sys.stderr.flush()
        os.dup2(fd, 2)
        os.close(fd)
    try:
        main(argv, backend=backend, make_step=make_step, device=device)
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        raise


def self_launch(argv, world, backend, make_step, device, timeout_s=1800.0):
    """`python bench.py --gpus N` without a launcher: spawn one process per GPU on this node (rendezvous on 127.0.0.1,
    a free port), like the reference's own `mp.spawn(run, nprocs=world_size)` (ddp-train.py:612-625).  Every rank then
    runs main() exactly as under `python -m torch.distributed.run`; rank 0 prints the one JSON line to the inherited
    stdout.  `make_step` (tests) must be a module-level function: the ranks are started with the spawn method.

    A rank that dies, raises or hangs must not leave the caller with a bare traceback (or nothing): the ranks are polled,
    killed at `timeout_s`, and on any failure ONE JSON line carrying "error", the failing rank and the tail of its stderr
    (NCCL_DEBUG=WARN output included) is printed in the bench's own format; the exit status is then 1."""
    import shutil
    import socket
    import tempfile
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    sys.stdout.flush()
    errdir = tempfile.mkdtemp(prefix="bench_ranks_")
    failure = None
    t0 = time.monotonic()
    ctx = mp.spawn(_spawned_rank, args=(world, port, argv, backend, make_step, device, errdir), nprocs=world, join=False)
    try:
        while not ctx.join(timeout=1.0):
            if time.monotonic() - t0 > timeout_s:
                alive = [i for i, p in enumerate(ctx.processes) if p.is_alive()]
                for p in ctx.processes:
                    if p.is_alive():
                        p.kill()
                failure = {"kind": "timeout", "rank": alive[0] if alive else None,
                           "what": "rank(s) %s still running after %.0f s: killed" % (alive, timeout_s)}
                break
    except Exception as e:  # ProcessRaisedException / ProcessExitedException: mp has terminated the other ranks
        failure = {"kind": type(e).__name__, "rank": getattr(e, "error_index", None),
                   "what": str(e).strip().splitlines()[-1][:300] if str(e).strip() else repr(e)}

    def tail(rank, n=25):
        try:
            return open(os.path.join(errdir, "rank%d.err" % rank), errors="replace").read().splitlines()[-n:]
        except OSError:
            return []

    rc = 0
    if failure is None:
        for r in range(world):  # nothing a rank said is lost
            for line in tail(r, 10 ** 6):
                print("[rank %d] %s" % (r, line), file=sys.stderr)
    else:
        r = failure["rank"] if failure["rank"] is not None else 0
        print(json.dumps({"metric": "bench.py --gpus %d (self-launched ranks)" % world, "value": None, "n_gpus": world,
                          "error": failure["what"], "error_kind": failure["kind"], "failed_rank": failure["rank"],
                          "stderr_tail": tail(r), "other_ranks_stderr_tail": {str(o): tail(o, 5) for o in range(world) if o != r},
                          "elapsed_s": time.monotonic() - t0}), flush=True)
        rc = 1
    shutil.rmtree(errdir, ignore_errors=True)
    return rc


def main(argv=None, backend="nccl", make_step=None, device=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--cams", type=int, default=None, help="override the camera count of the workload")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank renders its own scene; strong: ONE scene's cameras are sharded")
    ap.add_argument("--alpha-gain", type=float, default=1.0, help="1.0 = random-init opacity (nothing saturates)")
    ap.add_argument("--gout", default="randn", choices=["randn", "l1_matting"],
                    help="upstream gradient of the march step: Gaussian (the headline) or the gradient of an L1 matting loss against "
                         "a shifted target (what a training iteration hands the backward; the `train_like` leg of the default line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the train leg (profiling runs of the march kernels)")
    ap.add_argument("--no-render", action="store_true", help="skip the no-grad render timing (profiling runs: keeps the "
                                                             "per-kernel averages those of the training-path launches)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the secondary march workloads (saturated scene, C3, C4)")
    ap.add_argument("--no-collective-check", action="store_true",
                    help="N > 1: skip the timed flat all-reduce / rank identities (`collectives` object)")
    ap.add_argument("--dist-smoke", action="store_true",
                    help="with ONE rank: still create the process group and wrap the train model in DDP, so that the "
                         "collectives of the N > 1 path run through RCCL on a single GPU (a test of the plumbing, not a "
                         "measurement)")
    ap.add_argument("--config", default=None, help="--mode train: one of the reference's YAML files (configs/config*.yaml): "
                                                   "its train.batchsize, learning rate, schedule, clip and loss weights")
    ap.add_argument("--opts", default=[], nargs="+", help="key value overrides of the config (ddp-train.py:596)")
    ap.add_argument("--launch-timeout", type=float, default=1800.0,
                    help="self-launched ranks (--gpus N without a launcher) are killed after this many seconds")
    ap.add_argument("--bg", default="auto", choices=["auto", "on", "off"],
                    help="--mode train: the background MLP (auto: on except at C2, whose 80-frame batch holds 2 x 54 GB of bf16 "
                         "activations with it)")
    ap.add_argument("--bucket-mb", type=int, default=None,
                    help="--mode train with N > 1: DDP bucket_cap_mb (default 256 = ONE flat bucket, all-reduced after the backward; "
                         "25 / 64 overlap the all-reduce of early buckets with the rest of the backward -- ddp-train.py:312 uses "
                         "torch's default 25).  The first multi-GPU record can answer which wins: run once per value")
    ap.add_argument("--mode", default="march", choices=["march", "train"],
                    help="march (default, the contract metric + a `train` object); train: only the training loop, as "
                         "the headline value (iterations/s)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks here, as the reference starts its own (ddp-train.py:612-625)
        rc = self_launch(list(sys.argv[1:] if argv is None else argv), args.gpus, backend, make_step, device,
                         timeout_s=args.launch_timeout)
        if rc:
            raise SystemExit(rc)
        return
    rank, local_rank, world = env_ranks()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s) (WORLD_SIZE=%d)" % (args.gpus, world, world))
    gpu = device is None
    if gpu:
        assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device(device)  # tests only
    dist = init_process_group(backend, rank, world, dev, force=args.dist_smoke)

    if args.mode == "train":
        cfg = None
        if args.config:
            from ava256_amd.config import load_train_config
            cfg = load_train_config(args.config, args.opts)
        t = train_leg(args.workload, args.steps, args.warmup, rank, local_rank, world, dev, dist,
                      with_bg=(args.workload != "C2") if args.bg == "auto" else (args.bg == "on"),
                      ddp=(world > 1 or args.dist_smoke), config=cfg, bucket_mb=args.bucket_mb)
        if rank == 0:
            print(json.dumps({
                "metric": "train iters/sec, raymarch training path with a stand-in decoder (NOT ava-256's conv stacks)",
                "value": t["iters_per_s"], "unit": "iters/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": t["ms_per_iter"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32 (march) + bf16 autocast (background MLP)", "data": "synthetic",
                "config": {"workload": t["workload"], "parallelism": "DDP over %d rank(s), gradients only" % world},
                "train": t}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    step, info = (make_step or make_march_step_gpu)(args, rank, world, dev)
    events = []
    if gpu:
        from ava256_amd import _hooks as mm
        for _ in range(args.warmup):
            step()
        mm.set_event_sink(events)
        elapsed = run_timed(step, args.steps, 0, dist, dev)
        mm.set_event_sink(None)
    else:
        elapsed = run_timed(step, args.steps, args.warmup, dist, dev)
    kavg = kernel_averages(events)
    render_ms = render_half_ms = to_half_ms = None
    if gpu and "render" in info and not args.no_render:  # the forward alone as a renderer would call it (extra field, not the contract value)
        render_ms = time_calls(info["render"], dev)
    if gpu and "render_half" in info and not args.no_render:  # ... and over fp16 slabs (opt-in; never the contract value)
        with torch.no_grad():
            render_half_ms = time_calls(info["render_half"], dev)
        to_half_ms = time_calls(info["to_half"], dev)
    if gpu and "hit_packets_fn" in info and not args.no_render:
        info["hit_packets"] = info["hit_packets_fn"]()
    fused_ms = None
    if gpu and "fused_step" in info and not args.no_render:  # the training step with row N1's fusion (extra field as well)
        fused_ms = time_calls(info["fused_step"], dev)
    # SURVEY.md 8(d)'s secondary march runs, this rank: the trained-like scene and the other single-GPU configurations
    legs = {}
    if gpu and not args.no_render and not args.no_workloads and args.workload == "C2":
        legs["saturated"] = march_leg(args, rank, world, dev, "C2", 40.0, traffic_key="C2_saturated")
        legs["train_like"] = march_leg(args, rank, world, dev, "C2", 1.0, gout="l1_matting", traffic_key="C2")
        legs["C3"] = march_leg(args, rank, world, dev, "C3", 1.0, steps=20, traffic_key="C3")
        legs["C4"] = march_leg(args, rank, world, dev, "C4", 1.0, steps=20, traffic_key="C4")
    ident = collective_identity(dist, dev, world) if (dist is not None and not args.no_collective_check) else None

    # rays of all ranks per step: every rank contributes its own shard (gathered, so that uneven strong-scaling
    # shards are counted exactly)
    n_local = torch.tensor([info["n_local"]], device=dev, dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(n_local)
    cams_total = int(n_local.item())
    H, W, K, slab = info["H"], info["W"], info["K"], info["slab"]

    train = None
    if gpu and not args.no_train:
        # the reference's other logged number (ddp-train.py:446,512): iterations/s of the loop, here on the raymarch
        # training path with a stand-in decoder.  C3 = the reference's per-GPU batch shape (4 frames, K=16384) with the
        # background MLP (fused MFMA kernels); C2 = the 80-frame render batch, background off and (C2_bg) on.
        train = {"note": "stand-in decoder (per-primitive slab parameters), NOT ava-256's conv stacks",
                 "C3": train_leg("C3", 16, 5, rank, local_rank, world, dev, dist, with_bg=True),
                 "C2": train_leg("C2", 20, 5, rank, local_rank, world, dev, dist, with_bg=False)}
        if world > 1:
            # N > 1: the same C3 leg with DDP's default bucket size (25 MB: the all-reduce of early buckets overlaps the rest of the
            # backward, ddp-train.py:312) next to the ONE flat 256 MB bucket of the rows above -- the first multi-GPU record says
            # which wins on xGMI without anybody having to pass --bucket-mb
            b25 = train_leg("C3", 16, 5, rank, local_rank, world, dev, dist, with_bg=True, bucket_mb=25)
            train["C3"]["ddp_bucket_25mb"] = {k: b25[k] for k in ("iters_per_s", "ms_per_iter", "frames_per_s", "steps",
                                                                   "ddp_bucket_cap_mb", "final_loss")}
        if world == 1:
            # single process: the same iterations as ONE hipGraph replay each (Trainer(graph=True)); the eager rows above keep
            # the per-kernel HIP-event averages, which a replay cannot record
            for key, (st, wu, bg) in (("C3", (16, 7, True)), ("C2", (20, 7, False))):
                try:
                    g = train_leg(key, st, wu, rank, local_rank, world, dev, dist, with_bg=bg, graph=True)
                    train[key]["graph"] = {k: g[k] for k in ("iters_per_s", "ms_per_iter", "frames_per_s", "steps", "launch",
                                                             "final_loss")}
                except Exception as e:  # a capture that fails must not cost the bench line
                    train[key]["graph"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the 80-frame batch WITH the background MLP: its bf16 activations and their gradients for the backward are
        # 2 x 54 GB (80 x 512 x 512 pixels x 5 layers x 256 channels) -- run when the device has the room
        # (the decision is taken by ALL ranks together -- a rank that skipped the leg would leave the others in its barrier)
        free = torch.tensor([float(torch.cuda.mem_get_info(dev)[0])], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(free, op=dist.ReduceOp.MIN)
        if float(free.item()) > 160 * (1 << 30):
            train["C2_bg"] = train_leg("C2", 20, 3, rank, local_rank, world, dev, dist, with_bg=True)

    if rank == 0:
        rays_per_step = cams_total * H * W
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded shell scene, random-init slab statistics; SURVEY.md 8d)",
            "config": {"workload": "%s: %d cams/GPU, %dx%d, K=%d primitives, %d^3 RGBA slabs, fadeexp=8, dt=1/256" % (
                args.workload, info["N"] if args.scaling == "weak" else info["n_local"], H, W, K, slab),
                "alpha_gain": args.alpha_gain,
                "parallelism": ("every rank renders its own %d-camera scene" % info["N"] if args.scaling == "weak" else
                                "%d cameras of one scene sharded" % cams_total) + " over %d rank(s), no data-path collective" % world},
            "iters_per_s": args.steps / elapsed,
        }
        if kavg:
            bf, bb = algorithmic_bytes(info["n_local"], H, W, K, slab ** 3)
            dom = "march_backward" if kavg.get("march_backward", 0) >= kavg.get("march_forward", 0) else "march_forward"
            dom_bytes = bb if dom == "march_backward" else bf
            dom_ms = kavg.get(dom, float("nan"))
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
            # per-launch HBM bytes from separate rocprofv3 PMC passes (tools/make_traffic.py): a RECORDED measurement,
            # stamped with the commit it was taken at -- this run did not collect counters
            traffic = traffic_at = None
            traffic_fwd = valu = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf):
                try:
                    doc = json.load(open(tf))
                    traffic = doc.get(args.workload, {}).get(dom)
                    traffic_fwd = doc.get(args.workload, {}).get("march_forward")
                    traffic_at = doc.get("_measured_at_commit")
                    valu = doc.get("valu", {}).get(args.workload)
                except Exception:
                    traffic = None
            out["fwd_rays_per_s"] = (info["n_local"] * H * W) / (kavg["march_forward"] * 1e-3) if "march_forward" in kavg else None
            out["kernel_ms"] = kavg
            if render_ms is not None:
                out["render"] = {"what": "no-grad forward with rays made inside the march (mvp_march_forward_cams), this rank",
                                 "ms": render_ms, "rays_per_s": info["n_local"] * H * W / (render_ms * 1e-3)}
            if render_half_ms is not None:
                hb = render_bytes(info["n_local"], H, W, K, slab ** 3, 8)
                fb = render_bytes(info["n_local"], H, W, K, slab ** 3, 16)
                out["render_fp16"] = {
                    "what": "OPT-IN render path over fp16 RGBA slabs (mvp_march_render_half: 4 x 16-byte gathers per sample, fp32 "
                            "weights / interpolation / compositing), rays made inside the march, this rank; parity: kernel vs the "
                            "float64 oracle on the same rounded slabs within 2e-4 (tests/test_gpu_half.py); NOT the headline",
                    "dtype": "f16 slab storage, f32 arithmetic", "ms": render_half_ms,
                    "rays_per_s": info["n_local"] * H * W / (render_half_ms * 1e-3),
                    "speedup_vs_render_f32": (render_ms / render_half_ms) if render_ms else None,
                    "template_to_half_ms": to_half_ms,
                    "roofline": {"bound": "hbm", "achieved": hb / (render_half_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": hb / (render_half_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": hb, "traffic": None},
                    "render_f32_roofline": {"achieved": fb / (render_ms * 1e-3) / 1e9 if render_ms else None,
                                            "algorithmic_bytes_per_launch": fb}}
            if fused_ms is not None:
                out["fused_rays_step"] = {"what": "the same training step with the rays made inside the forward march "
                                                  "(mvpraymarch_from_cameras + backward: no raydirs launch), this rank",
                                          "ms": fused_ms, "rays_per_s": info["n_local"] * H * W / (fused_ms * 1e-3)}
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_measured_at_commit": traffic_at,
                               "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                               "traffic_ratio": (traffic / dom_bytes) if traffic else None,
                               "fwd": {"achieved": bf / (kavg.get("march_forward", float("nan")) * 1e-3) / 1e9,
                                       "algorithmic_bytes_per_launch": bf,
                                       "avg_launch_ms": kavg.get("march_forward"), "traffic": traffic_fwd,
                                       "traffic_ratio": (traffic_fwd / bf) if traffic_fwd else None}}
            if valu:
                # What binds these kernels is not HBM (frac above) but VALU issue + latency: recorded SQ counters of the
                # same evidence run as `traffic` (tools/make_traffic.py; same commit stamp), per launch of each kernel.
                hp = info.get("hit_packets")
                out["roofline"]["valu"] = {
                    "what": "busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
                            "wave_insts = SQ_INSTS_VALU per launch; recorded counter passes (traffic_measured_at_commit)",
                    "kernels": {k: dict(v, wave_insts_per_hit_packet=(v["wave_insts"] / hp) if hp else None)
                                for k, v in valu.items()},
                    "hit_packets_per_launch": hp}
        if legs:
            out["saturated"] = legs.pop("saturated")
            out["train_like"] = legs.pop("train_like")
            out["workloads"] = legs
        if ident is not None:
            out["collectives"] = ident
        if train is not None:
            out["train"] = train
        if gpu and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(info["N"], H, W, K, slab)
            # SURVEY.md 8(d) names C1 (the reference's own CPU-runnable configuration: 4 cameras, 128x128, K=512) as the
            # second CPU timing: all four cameras, a few seconds
            c1 = WORKLOADS["C1"]
            out["cpu_baseline"]["C1"] = cpu_baseline(c1[0], c1[1], c1[2], c1[3], c1[4], budget_s=4.0, cams=c1[0])
            out["reference_cpu"] = reference_cpu()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
