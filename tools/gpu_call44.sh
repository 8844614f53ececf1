#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_img.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_img.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
for w in C2 C3 C4; do echo -n "$w: "; timeout 300 $B --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()}, d.get('render',{}).get('ms'))"; done
echo -n "C2 a20: "; timeout 300 $B --alpha-gain 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()})"
echo -n "C2 cams=12: "; timeout 300 $B --cams 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()})"
