#!/bin/bash
# Round 4's kernels, long runs against the oracle (identical: True / False per line).
cd $GRAFT_REPO_ROOT
part=${1:-all}
out=gpurun_out/r04_soak_$part.txt
: > $out
run() { timeout 600 python tests/soak.py "$@" 2>&1 | grep -v Yippee | tail -1 | cut -c1-600 >> $out; }
if [ $part = wavelet ]; then
# the pipelined wavelet kernel (kernels_wavelet.h wavelet_pipe_kernel): one chain per workgroup
# (speculation on reject), two and four
run 3000 snooker 256 wavelet 16384
run 1500 snooker 512 wavelet 16384
run 800 snooker 1024 wavelet 16384
run 1500 mrw 512 wavelet 16384
run 3000 mrw 256 wavelet 8192
run 2000 snooker 300 wavelet 8192
cat $out
exit 0
fi
if [ $part != rows ]; then
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
fi

# the rows layout (kernels_rows.h): large ensembles, the chains of a wave side by side
run 2000 demc 8192
run 3000 demc 4096
run 3000 demc 6000
run 2000 snooker 8192
run 3000 snooker 4096
run 3000 mrw 4096
run 2000 mrw 8192

cat $out
