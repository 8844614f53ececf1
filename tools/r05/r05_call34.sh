#!/bin/bash
# the 80-frame training leg with the background MLP at 4 / 8 / 32 images per kernel pair (16 = product default: r05_call33.sh)
O=gpurun_out/r05grp; mkdir -p $O
for n in 8 4 32; do
  timeout 14 python -c "
import sys, json, io, contextlib
import ava256_amd.bgmlp as b; b.IMAGES_PER_CALL = $n
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(['--mode', 'train', '--workload', 'C2', '--bg', 'on'])
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('images_per_call', $n, 'iters_per_s', round(d['value'], 3), 'ms', round(d['ms_per_step'], 2))
" 2>/dev/null < /dev/null | tee -a $O/groups.txt
done
