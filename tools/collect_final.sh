#!/bin/bash
# The measurements that must match the FINAL library of the round (run last):
#   - the bench line, the kernel trace and the two HBM counter passes of the headline kernel
#     (same commands as tools/collect_profiles.sh steps 1-3);
#   - bench lines and kernel traces of the one-launch tiled plans (C5, polynomial q5) and of
#     the three-launch plan they replace (MC3HIP_NO_RESIDENT=1).
# Usage: bash tools/collect_final.sh   -> gpurun_out/r02f/, reduced by tools/reduce_final.py
out=gpurun_out/r02f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $out/c2_bench.json 2> $out/c2_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- \
    python3 bench.py --no-cpu-baseline > $out/kt_c2.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_c2_$ctr -o c2 -- \
        python3 bench.py --no-cpu-baseline > $out/pmc_c2_$ctr.log 2>&1
done
timeout 600 python tools/bench_configs.py c2snooker c3one c4 c5 q5 > $out/configs.jsonl 2> $out/configs.err
MC3HIP_NO_RESIDENT=1 timeout 600 python tools/bench_configs.py c5 q5 > $out/configs_three_launch.jsonl 2> $out/configs3.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_q5 -o q5 -- \
    python3 tools/bench_configs.py q5 > $out/kt_q5.log 2>&1
timeout 300 python tools/e2e_sample.py 1e7 256 1 > $out/e2e_sample_1e7.txt 2>&1
timeout 300 python tools/e2e_sample.py 1024000 512 1 c5 > $out/e2e_sample_c5.txt 2>&1
ls $out
