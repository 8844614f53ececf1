#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc $?"
tail -2 $O/bench_torchrun.err; cut -c1-300 $O/bench_torchrun.json
