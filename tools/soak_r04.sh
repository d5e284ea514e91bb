#!/bin/bash
# Round 4's kernels, long runs against the oracle (identical: True / False per line).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_soak.txt
: > $out
run() { timeout 600 python tests/soak.py "$@" 2>&1 | grep -v Yippee | tail -1 | cut -c1-600 >> $out; }
run 40000 demc 256 zlag=4
run 40000 demc 256 zlag=1
run 20000 demc 2048 zlag=4
run 40000 snooker 256 zlag=4
run 40000 demc 256 fepsilon=0.01
run 40000 snooker 256 fepsilon=0.01
run 10000 demc 256 thinning=4 fx=1
run 30000 demc 256 poly=8
run 30000 snooker 256 poly=5
run 20000 demc 256 lin=10
run 20000 snooker 256 lin=10 fx=1
run 20000 demc 256 lin=5 zlag=2 thinning=2
cat $out
