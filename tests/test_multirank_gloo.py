"""N>1 path on CPU: two gloo ranks shard the cameras exactly as bench.py does (no data-path collective), each
renders its shard (with the CPU checker standing in for the device kernels), and the gathered result must equal
the single-process render; the elapsed-time MAX-reduce and barrier run over the real process group."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ava256_amd.dist_util import barrier, max_over_ranks, shard_range
    from ava256_amd.scene import make_scene
    from oracle.mvp_oracle import Oracle
    s = make_scene(5, 24, 24, 64, device="cpu", seed=3, alpha_gain=10.0)   # every rank builds the same scene
    lo, hi = shard_range(5, rank, world)
    o = Oracle("f32")
    sl = slice(lo, hi)
    rp, rd, tm = o.raydirs(s["campos"][sl].numpy(), s["camrot"][sl].numpy(), s["focal"][sl].numpy(),
                           s["princpt"][sl].numpy(), s["pixelcoords"][sl].numpy(), s["volradius"])
    rgba, _, _ = o.march_forward(rp, rd, s["stepsize"], tm, s["primpos"][sl].numpy(), s["primrot"][sl].numpy(),
                                 s["primscale"][sl].numpy(), s["template"][sl].numpy())
    np.save(os.path.join(outdir, "shard%d.npy" % rank), rgba)
    barrier()
    m = max_over_ranks(1.0 + rank)
    assert m == float(world), m
    dist.destroy_process_group()


def test_camera_sharding_two_ranks(tmp_path):
    from ava256_amd.dist_util import shard_range
    from ava256_amd.scene import make_scene
    from oracle.mvp_oracle import Oracle
    assert [shard_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_range(80, r, 8) for r in range(8)][-1] == (70, 80)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s = make_scene(5, 24, 24, 64, device="cpu", seed=3, alpha_gain=10.0)
    o = Oracle("f32")
    rp, rd, tm = o.raydirs(s["campos"].numpy(), s["camrot"].numpy(), s["focal"].numpy(), s["princpt"].numpy(),
                           s["pixelcoords"].numpy(), s["volradius"])
    full, _, _ = o.march_forward(rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(),
                                 s["primscale"].numpy(), s["template"].numpy())
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "shard%d.npy" % r)) for r in range(2)], 0)
    assert got.shape == full.shape and np.array_equal(got, full)
