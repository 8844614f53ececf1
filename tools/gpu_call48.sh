#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'], sorted(d['train'].keys()), d['train']['C2_bg']['iters_per_s'])"
