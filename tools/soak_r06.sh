#!/bin/bash
# Round 6 (the same list on the final library of round 6)'s kernels, long runs against the oracle (identical: True / False per line), and the
# fuzz generators on seeds the suite does not use.   bash tools/soak_r06.sh [soak|hunt|all]
cd $GRAFT_REPO_ROOT
part=${1:-all}
mkdir -p gpurun_out/r06
if [ $part != hunt ]; then
out=gpurun_out/r06/soak_round5_kernels.txt
: > $out
run() { timeout 900 python tests/soak.py "$@" 2>&1 | grep -v Yippee | tail -1 | cut -c1-600 >> $out; }
# the pipelined wavelet kernel's chain wave for one chain per workgroup (pipe_chain_wave_one)
run 4000 snooker 256 wavelet 16384
run 6000 snooker 256 wavelet 8192
run 3000 snooker 100 wavelet 16384
# the one-wave wavelet chain (wavelet_one_kernel)
run 12000 snooker 256 wavelet 2048
run 20000 snooker 256 wavelet 256
run 20000 snooker 64 wavelet 64
# ... for the random walk and DEMC (wavelet_one_other_kernel), with the support distribution,
# and on four waves (4 096 points)
run 12000 mrw 256 wavelet 2048
run 12000 demc 256 wavelet 1024
run 12000 demc 100 wavelet 256 fepsilon=0.01
run 12000 snooker 256 wavelet 512 fepsilon=0.01
run 8000 snooker 256 wavelet 4096
run 8000 demc 256 wavelet 4096
run 8000 mrw 128 wavelet 4096
# the tiles in the registers of several waves (lean_*_wide_kernel)
run 20000 demc 256 ndata=2048
run 20000 demc 256 ndata=4096
run 12000 demc 256 ndata=8192
run 8000 demc 256 ndata=16384
run 20000 snooker 256 ndata=4096
run 12000 snooker 256 ndata=8192
run 6000 snooker 256 ndata=16384
run 20000 demc 512 ndata=3000
# the resident tiled plan with nobody in step (resident_async_kernel): C5, and a ragged shape
run 300 demc 512 lin=10 ndata=100000 fx=1
MC3HIP_RS_ASYNC=1 run 1500 demc 83 lin=7 ndata=30000 fx=1
MC3HIP_RS_ASYNC=1 run 1000 demc 200 lin=11 ndata=20000 fepsilon=0.01
MC3HIP_RS_ASYNC=1 run 1000 snooker 96 lin=10 ndata=40000 fx=1
run 200 snooker 512 lin=10 ndata=100000 fx=1
# the random walk / the support distribution beyond four coefficients; the wide loops
# specialised at run time
run 20000 mrw 256 poly=8
run 20000 demc 256 lin=10 fepsilon=0.01
run 20000 snooker 256 poly=6 fepsilon=0.01
run 20000 mrw 256 ndata=8192
run 20000 snooker 256 ndata=4096 fepsilon=0.01
run 20000 demc 256 ndata=2048 fepsilon=0.01
# the headline loop and its neighbours on the final library
run 100000 demc 256
run 60000 snooker 256
# ... beyond 1 024 chains (lean_demc_kernel BIG: the wait laid out by hand), two sizes
run 40000 demc 2048
run 40000 demc 1500
cat $out
fi
if [ $part != soak ]; then
out=gpurun_out/r06/fuzz_hunt.txt
: > $out
echo "== python tools/fuzz_hunt.py 400000 1600 (all generators)" >> $out
timeout 1500 python tools/fuzz_hunt.py 400000 1600 >> $out 2>&1
echo "== FUZZ_KIND=5 python tools/fuzz_hunt.py 410000 400 (the wide loops)" >> $out
FUZZ_KIND=5 timeout 900 python tools/fuzz_hunt.py 410000 400 >> $out 2>&1
echo "== FUZZ_KIND=2 python tools/fuzz_hunt.py 420000 600 (wavelet likelihood, either model kind)" >> $out
FUZZ_KIND=2 timeout 900 python tools/fuzz_hunt.py 420000 600 >> $out 2>&1
tail -20 $out
fi
