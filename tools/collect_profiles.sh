#!/bin/bash
# Collect the round's measurements on the MI355X box into gpurun_out/<tag>/ (copied to
# profiles/ afterwards).  Usage: bash tools/collect_profiles.sh r02
# rocprofv3 is always given python3 itself (never a wrapper), PMC passes are separate runs
# with nothing but --pmc (+ --kernel-trace for the kernel names).
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PORT=29731

# 1. the bench line, as the driver runs it
timeout 600 python bench.py > $out/c2_bench.json 2> $out/c2_bench.err

# 2. kernel trace of the same workload
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- \
    python3 bench.py --no-cpu-baseline > $out/kt_c2.log 2>&1

# 3. HBM traffic: one counter per pass
for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_c2_$ctr -o c2 -- \
        python3 bench.py --no-cpu-baseline > $out/pmc_c2_$ctr.log 2>&1
done

# 4. the other configurations on one GPU
timeout 900 python tools/bench_configs.py c2 c2snooker c3one c4 c5 > $out/configs.jsonl 2> $out/configs.err

# 5. C4 / C5: kernel trace and instruction-mix / stall counters
for cfg in c4 c5; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$cfg -o $cfg -- \
        python3 tools/bench_configs.py $cfg > $out/kt_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv \
        -d $out/pmc1_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc1_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
        SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv \
        -d $out/pmc2_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc2_$cfg.log 2>&1
done

# 6. multi-rank paths on the one GPU: RCCL exchange (one rank, captured in a hipGraph),
#    peer-to-peer hand-off with 2 and 8 ranks
MC3HIP_BENCH_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --refreshes 4096 \
    > $out/c2_bench_rccl_graph_1rank.json 2> $out/c2_bench_rccl_graph_1rank.err
for n in 2 8; do
    MC3HIP_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
        --master-addr 127.0.0.1 --master-port $((PORT + n)) bench.py --gpus $n --steps 4 --warmup 1 \
        > $out/c2_bench_${n}ranks_on_1gpu.json 2> $out/c2_bench_${n}ranks_on_1gpu.err
done

# 7. end to end through sample(), and the soak runs
timeout 300 python tools/e2e_sample.py 1e7 256 1 > $out/e2e_sample_1e7.txt 2>&1
timeout 900 python tests/soak.py 1000000 demc > $out/soak_demc_1e6.txt 2>&1
ls $out
