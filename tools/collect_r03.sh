#!/bin/bash
# Round 3: collect the measurements of the FINAL library on the MI355X box into
# gpurun_out/r03/ (reduced to profiles/r03_* by tools/reduce_r03.py).
# rocprofv3 is always given python3 itself (never a wrapper); PMC passes are separate runs with
# nothing but --pmc (+ --kernel-trace for the kernel names).
#   bash tools/collect_r03.sh [part ...]      parts: bench kt pmc configs c4 c5 ranks loops soak
out=gpurun_out/r03
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
parts=${*:-bench kt pmc configs c4 c5 ranks soak}
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }

# the bench line as the driver runs it (with the CPU baseline and the end-to-end timings)
has bench && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
# kernel trace of the same workload
has kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- \
    python3 bench.py --no-cpu-baseline --no-e2e > $out/kt_c2.log 2>&1
# HBM traffic of the headline kernel: one counter per pass
if has pmc; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_c2_$ctr -o c2 -- \
            python3 bench.py --no-cpu-baseline --no-e2e > $out/pmc_c2_$ctr.log 2>&1
    done
fi
# the other configurations on one GPU (each in a fresh process)
if has configs; then
    timeout 1200 python tools/bench_configs.py c2 c2snooker c2mrw c2thin10demc c2thin10snooker c2thin10mrw c2twodemc c2twosnooker c2twomrw c3one c3onesnooker c3x2 c3x4 c3x2snooker c3x2mrw c4 c4wide c4demc c4mrw c4x2 c4x4 c5 q5 p6 p6matrix \
        > $out/configs.jsonl 2> $out/configs.err
    MC3HIP_NO_RESIDENT=1 timeout 600 python tools/bench_configs.py c5 q5 > $out/configs_three_launch.jsonl 2> $out/configs3.err
    MC3HIP_NO_CHAINWAVE=1 timeout 600 python tools/bench_configs.py c4 c4wide > $out/configs_no_chainwave.jsonl 2>> $out/configs3.err
    # the general chain (run_chain) instead of the hand-structured chain wave / chain owner
    MC3HIP_NO_CW_LEAN=1 MC3HIP_NO_RS_LEAN=1 timeout 600 python tools/bench_configs.py c4 c4wide c5 q5 \
        > $out/configs_general_chain.jsonl 2>> $out/configs3.err
fi
# C4 / C5: kernel trace, instruction mix and stall counters of the dominant kernel
for cfg in c4 c5; do
    has $cfg || continue
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$cfg -o $cfg -- \
        python3 tools/bench_configs.py $cfg > $out/kt_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv \
        -d $out/pmc1_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc1_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
        SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv \
        -d $out/pmc2_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc2_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC \
        SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace \
        --output-format csv -d $out/pmc3_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc3_$cfg.log 2>&1
done
# several ranks on the one GPU: the self-launching bench (peer-to-peer hand-off; the RCCL
# segment cannot run with several ranks on one device and says so), and the RCCL-graph path
# with one rank
if has ranks; then
    MC3HIP_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --refreshes 16384 \
        > $out/c2_bench_2ranks_on_1gpu.json 2> $out/c2_bench_2ranks_on_1gpu.err
    MC3HIP_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --workload c3 --steps 3 --warmup 1 --refreshes 4096 \
        > $out/c3_bench_8ranks_on_1gpu.json 2> $out/c3_bench_8ranks_on_1gpu.err
    MC3HIP_BENCH_DEVICE=0 MC3HIP_RESIDENT_MAXCOL=1 timeout 900 python bench.py --gpus 4 --workload c5 --steps 2 --warmup 1 \
        > $out/c5_bench_4ranks_on_1gpu.json 2> $out/c5_bench_4ranks_on_1gpu.err
    MC3HIP_BENCH_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --refreshes 4096 \
        > $out/c2_bench_rccl_graph_1rank.json 2> $out/c2_bench_rccl_graph_1rank.err
    timeout 300 sh tools/probes/run_xgmi_handoff.sh > $out/handoff_probe.json 2> $out/handoff_probe.err
fi
# kernel statistics of the other hand-laid-out loops
if has loops; then
    for cfg in c2snooker c2mrw c3x2snooker c2thin10demc; do
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$cfg -o $cfg -- \
            python3 tools/bench_configs.py $cfg > $out/kt_$cfg.log 2>&1
    done
fi
has soak && timeout 900 python tests/soak.py 1000000 demc > $out/soak_demc_1e6.txt 2>&1
ls $out
