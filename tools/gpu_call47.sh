#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_final.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_final.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()}, d['render']['ms'])"
