#!/bin/bash
# training batches walked in groups of 16 images (gradient planes of one group at a time): MLP + train-loop tests, the 80-frame leg
O=gpurun_out/r05grp; mkdir -p $O
timeout 35 python -m pytest tests/test_bgmlp.py tests/test_trainloop.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -2 $O/tests.log
timeout 40 python bench.py --mode train --workload C2 --bg on > $O/c2bg.json 2>$O/c2bg.err < /dev/null; echo "bench rc $?"; cut -c1-400 $O/c2bg.json
