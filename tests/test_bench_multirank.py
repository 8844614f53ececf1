"""bench.py's N>1 control flow on CPU: two gloo ranks run bench.main() itself -- process-group init, warm-up, barrier +
MAX-reduce of the elapsed time, per-rank camera shards, rank-0 JSON line -- with the step factory replaced by one that
renders with the CPU checker (in this test only; bench.py's own factory has no CPU path)."""
import json
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_step_factory(args, rank, world, dev):
    """Same contract as bench.make_march_step_gpu: (step, info); the shard comes from bench.camera_shard."""
    import bench
    from ava256_amd.scene import make_scene
    from oracle.mvp_oracle import Oracle
    H = W = 16
    K = 32
    N, lo, hi, seed = bench.camera_shard(args, rank, world)
    s = make_scene(N, H, W, K, device="cpu", seed=seed, alpha_gain=5.0)
    o = Oracle("f32")
    sl = slice(lo, hi)
    calls = {"n": 0}

    def step():
        rp, rd, tm = o.raydirs(s["campos"][sl].numpy(), s["camrot"][sl].numpy(), s["focal"][sl].numpy(),
                               s["princpt"][sl].numpy(), s["pixelcoords"][sl].numpy(), s["volradius"])
        o.march_forward(rp, rd, s["stepsize"], tm, s["primpos"][sl].numpy(), s["primrot"][sl].numpy(),
                        s["primscale"][sl].numpy(), s["template"][sl].numpy())
        calls["n"] += 1

    step.calls = calls
    return step, {"n_local": hi - lo, "H": H, "W": W, "K": K, "slab": 8, "N": N}


def _worker(rank, world, port, outdir, scaling):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    import bench
    from test_bench_multirank import _oracle_step_factory
    out = open(os.path.join(outdir, "rank%d.out" % rank), "w")
    sys.stdout = out
    try:
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "C1", "--cams", "5",
                    "--scaling", scaling], backend="gloo", make_step=_oracle_step_factory, device="cpu")
    finally:
        sys.stdout = sys.__stdout__
        out.close()


def _run(tmp_path, scaling):
    port = _free_port()
    d = tmp_path / scaling
    d.mkdir()
    mp.spawn(_worker, args=(2, port, str(d), scaling), nprocs=2, join=True)
    r0 = open(d / "rank0.out").read().strip().splitlines()
    assert open(d / "rank1.out").read().strip() == ""          # only rank 0 prints
    assert len(r0) == 1                                        # ONE JSON line
    return json.loads(r0[0])


def test_bench_two_ranks_weak_scaling(tmp_path):
    j = _run(tmp_path, "weak")
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["unit"] == "rays/s" and j["higher_is_better"] is True and j["vs_baseline"] is None
    # whole-job aggregate: every rank renders its own 5 cameras -> 10 cameras of 16x16 rays per step
    rays = 2 * 5 * 16 * 16
    assert abs(j["value"] - rays * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) <= 1e-6 * j["value"]
    assert "cpu_baseline" not in j and "train" not in j        # GPU-only legs
    # what the first multi-GPU record has to answer from the line itself: did the process group see
    # N ranks, on which devices, and the bus bandwidth of one flat gradient-sized all-reduce
    c = j["collectives"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["allreduce_sums_ok"] is True
    assert [r["rank"] for r in c["ranks"]] == [0, 1] and len({r["pid"] for r in c["ranks"]}) == 2
    assert all(set(r) >= {"rank", "device", "name", "pci", "pid"} for r in c["ranks"])
    assert c["allreduce_bytes"] > 0 and c["allreduce_ms"] > 0
    assert abs(c["allreduce_busbw_gbs"] - 2.0 * (2 - 1) / 2 * c["allreduce_bytes"] / (c["allreduce_ms"] * 1e-3) / 1e9) <= 1e-6 * c["allreduce_busbw_gbs"]
    assert abs(c["allreduce_busbw_gbs"] - c["allreduce_algbw_gbs"]) <= 1e-9 + 1e-6 * c["allreduce_algbw_gbs"]   # n = 2: 2 (n-1)/n = 1


def test_bench_two_ranks_strong_scaling(tmp_path):
    from ava256_amd.dist_util import shard_range
    j = _run(tmp_path, "strong")
    assert j["scaling"] == "strong" and j["n_gpus"] == 2
    assert [shard_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]     # uneven shards are counted exactly:
    rays = 5 * 16 * 16                                                     # 5 cameras in total, not 2 x ceil(5/2)
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - rays) <= 1e-6 * rays


def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path, monkeypatch, capfd):
    """`python bench.py --gpus 2` with no torch.distributed.run around it (the reference starts its ranks itself,
    ddp-train.py:612-625): main() spawns the two ranks, and exactly one JSON line with n_gpus == 2 comes out."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("OMP_NUM_THREADS", "2")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    import bench
    capfd.readouterr()
    bench.main(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C1", "--cams", "3"], backend="gloo",
               make_step=_oracle_step_factory, device="cpu")
    lines = [l for l in capfd.readouterr().out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak"
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 2 * 3 * 16 * 16) <= 1e-6 * 2 * 3 * 16 * 16


def _failing_step_factory(args, rank, world, dev):
    """Rank 1 dies while it builds its step (what a rank that cannot open its GPU, or runs out of memory, looks like)."""
    if rank == 1:
        print("RCCL-like diagnostic line of rank 1", file=sys.stderr)
        raise RuntimeError("rank 1 cannot make its step")
    return _oracle_step_factory(args, rank, world, dev)


def _hanging_step_factory(args, rank, world, dev):
    import time
    if rank == 0:
        time.sleep(3600)
    return _oracle_step_factory(args, rank, world, dev)


def _launch(monkeypatch, capfd, factory, extra=()):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("OMP_NUM_THREADS", "2")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    import bench
    capfd.readouterr()
    rc = 0
    try:
        bench.main(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C1", "--cams", "3"] + list(extra),
                   backend="gloo", make_step=factory, device="cpu")
    except SystemExit as e:
        rc = e.code
    lines = [l for l in capfd.readouterr().out.strip().splitlines() if l.startswith("{")]
    return rc, lines


def test_self_launch_reports_a_failed_rank_in_its_own_format(monkeypatch, capfd):
    """The first multi-GPU run this repository is ever offered must fail loudly and legibly: a rank that raises gives ONE
    JSON line with "error", the failing rank and the tail of ITS stderr, and a non-zero exit status -- not a traceback of
    the parent and no line."""
    rc, lines = _launch(monkeypatch, capfd, _failing_step_factory)
    assert rc == 1 and len(lines) == 1, (rc, lines)
    j = json.loads(lines[0])
    assert j["value"] is None and j["n_gpus"] == 2 and j["failed_rank"] == 1
    assert "cannot make its step" in j["error"] or any("cannot make its step" in l for l in j["stderr_tail"])
    assert any("RCCL-like diagnostic line" in l for l in j["stderr_tail"])      # what the rank wrote to fd 2 is there


def test_self_launch_kills_hung_ranks(monkeypatch, capfd):
    rc, lines = _launch(monkeypatch, capfd, _hanging_step_factory, extra=["--launch-timeout", "20"])
    assert rc == 1 and len(lines) == 1, (rc, lines)
    j = json.loads(lines[0])
    assert j["error_kind"] == "timeout" and j["value"] is None and "killed" in j["error"]


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["march", "train", "train_bucket25"])
def test_collectives_of_the_multi_gpu_path_run_through_rccl_on_one_gpu(mode):
    """No multi-GPU node is available to these tests, and RCCL refuses two ranks on one device -- but a ONE-rank group
    still takes every collective of the N > 1 path through RCCL: `bench.py --dist-smoke` initialises the "nccl" group with
    `device_id`, runs the barriers and the MAX all-reduce of the elapsed time around the timed steps, and (train mode)
    wraps the model in DistributedDataParallel with the one flat gradient bucket and runs the decoder's adaptwarps MAX
    all-reduce.  The line must come out, with finite numbers."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dist-smoke", "--steps", "3", "--warmup", "1", "--workload", "C1",
           "--no-cpu-baseline", "--no-train", "--no-render"]
    if mode.startswith("train"):
        cmd += ["--mode", "train"]
    if mode == "train_bucket25":   # several DDP buckets instead of the one flat one (--bucket-mb, ddp-train.py:312's default)
        cmd += ["--bucket-mb", "25"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["ms_per_step"] > 0
    if mode == "march":   # the flat gradient-sized all-reduce and the rank identities, through RCCL (one rank: busbw = 0)
        c = j["collectives"]
        assert c["world_size"] == 1 and c["backend"] == "nccl" and c["allreduce_sums_ok"] is True
        assert c["allreduce_bytes"] == 46_870_000 * 4 and c["allreduce_ms"] > 0 and c["ranks"][0]["pci"] is not None
    if mode.startswith("train"):
        t = j["train"]
        assert t["allreduce_mb"] > 0 and t["final_loss"] == t["final_loss"]      # DDP was on; the loss is finite
        assert t["ddp_bucket_cap_mb"] == (25 if mode == "train_bucket25" else 256)
