#!/bin/bash
# round-2 call 1: LDS-atomic microbenchmark (kept under profiles/), then the whole GPU suite with the new config tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02a
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomic.hip -o /tmp/lds_atomic 2> gpurun_out/r02a/ubench_build.log
timeout 120 /tmp/lds_atomic > gpurun_out/r02a/ubench_lds_atomic.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_cost.hip -o /tmp/gather_cost 2>> gpurun_out/r02a/ubench_build.log
timeout 120 /tmp/gather_cost > gpurun_out/r02a/ubench_gather_cost.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc $?"
tail -5 gpurun_out/r02a/pytest.log
cat gpurun_out/r02a/ubench_lds_atomic.txt | head -30
