"""tools/profile_train_ops.py [workload] [with_bg] -- torch.profiler over a few training iterations of the bench's train leg:
GPU time per OPERATOR (with input shapes), forward and backward, to see which eager statements of the stand-in
decoder / loss / optimizer the small kernels of `*_train_C3_kernel_stats.csv` belong to."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ava256_amd.trainloop import (BackgroundMLPStandIn, CodeEncoderStandIn, ColorCalStandIn, RaymarchTrainModel,  # noqa: E402
                                  SlabDecoderStandIn, Trainer, make_training_batch)

workload = sys.argv[1] if len(sys.argv) > 1 else "C3"
with_bg = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
N, H, W, K, slab = bench.WORKLOADS[workload]
batch, volradius = make_training_batch(N, H, W, K, dev, seed=1112, ncams=80, nident=4, target_decoder=SlabDecoderStandIn(K, slab, seed=9))
model = RaymarchTrainModel(SlabDecoderStandIn(K, slab, seed=1), volradius, colorcal=ColorCalStandIn(80, 4),
                           bgmodel=BackgroundMLPStandIn(80, 4) if with_bg else None, encoder=CodeEncoderStandIn()).to(dev)
tr = Trainer(model, ddp=False)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile, record_function  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(4):
        with record_function("ITER"):
            tr.step(batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48,
                                                         max_shapes_column_width=70))
