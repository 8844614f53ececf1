#!/bin/bash
# round-6 GPU call 8: bgmlp backward with the next tile's first inputs requested ahead of the last plane's store burst:
# MLP tests, fused-MLP microbenchmark A/B against the previous library, train C3 / C2_bg.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06h; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
for i in 1 2 3; do
  timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 2>/dev/null < /dev/null | cut -c1-300 | tee -a $O/bgmlp_new.txt
  timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 build_variants/libmvp_r06cut.so 2>/dev/null < /dev/null | cut -c1-300 | tee -a $O/bgmlp_old.txt
done
for i in 1 2; do
  timeout 300 python bench.py --mode train --workload C3 --steps 16 --warmup 5 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('C3', d['value'], {k: round(v,3) for k,v in d['train']['kernel_ms'].items() if 'bgmlp' in k})" | tee -a $O/train.txt
done
timeout 400 python bench.py --mode train --workload C2 --bg on --steps 10 --warmup 3 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('C2_bg', d['value'], {k: round(v,3) for k,v in d['train']['kernel_ms'].items() if 'bgmlp' in k or 'march' in k})" | tee -a $O/train.txt
