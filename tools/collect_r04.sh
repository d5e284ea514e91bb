#!/bin/bash
# Round 4: collect measurements on the MI355X box into gpurun_out/r04/ (reduced to
# profiles/r04_* by tools/reduce_r04.py).  rocprofv3 is always given python3 itself; PMC passes
# are separate runs with nothing but --pmc (+ --kernel-trace for the kernel names).
#   bash tools/collect_r04.sh [part ...]
# parts: bench kt pmc configs offlean lag sweep sweep8 ranks big c4 c4x2 c5 c4family
out=gpurun_out/r04
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
parts=${*:-bench kt pmc configs offlean lag sweep ranks}
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }

# the bench line as the driver runs it
has bench && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err
# kernel trace of the same workload (shorter: the trace of one step is the same kernel)
has kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- \
    python3 bench.py --no-cpu-baseline --no-e2e --steps 4 --warmup 1 > $out/kt_c2.log 2>&1
if has pmc; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_c2_$ctr -o c2 -- \
            python3 bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 --refreshes 98304 > $out/pmc_c2_$ctr.log 2>&1
    done
fi
if has configs; then
    timeout 1500 python tools/bench_configs.py c2 c2snooker c2mrw c2thin10demc c2thin10snooker c2thin10mrw c2twodemc c2twosnooker c2twomrw c3one c3onesnooker c3x2 c3x4 c3x2snooker c3x2mrw c3x4snooker c3x4mrw c4 c4wide c4demc c4mrw c4x2 c4x4 c5 q5 p6matrix \
        > $out/configs.jsonl 2> $out/configs.err
fi
# the shapes off the hand-laid-out loops: as they run now, and behind the switch (the general /
# lean run_chain instances they ran on before round 4)
if has offlean; then
    OFF="off_shared off_eps off_eps_snooker off_poly5 off_poly8 off_lin5 off_lin10 off_prior_thin10 off_prior_snooker off_prior off_shared_thin10 off_poly5_snooker off_lin10_snooker"
    timeout 1200 python tools/bench_configs.py $OFF > $out/offlean_configs.jsonl 2> $out/offlean.err
    MC3HIP_NO_LEAN=1 timeout 1200 python tools/bench_configs.py $OFF > $out/offlean_configs_no_lean.jsonl 2>> $out/offlean.err
fi
# the declared multi-GPU cadence on one GPU
has lag && timeout 900 python tools/bench_configs.py c2lag1 c2lag4 c2lag4snooker c3onelag4 c3x2lag4 c3x4lag4 > $out/lag_configs.jsonl 2> $out/lag.err
# injected xGMI latency: ranks on ONE GPU
has sweep && timeout 3000 python tools/delay_sweep.py $out/delay_sweep.jsonl --ranks 2,4 --zlags 0,1,2,4 --delays 0,1000,2000,3000 > $out/sweep.log 2>&1
has sweep8 && timeout 3000 python tools/delay_sweep.py $out/delay_sweep.jsonl --ranks 8 --zlags 0,1,2,4,8 --delays 0,1000,2000,3000 > $out/sweep8.log 2>&1
has sweepc3 && timeout 3000 python tools/delay_sweep.py $out/delay_sweep_c3.jsonl --workload c3 --ranks 2,4,8 --zlags 0,4 --delays 0,2000 --refreshes 8192 > $out/sweepc3.log 2>&1
if has ranks; then
    MC3HIP_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --refreshes 16384 \
        > $out/c2_bench_2ranks_on_1gpu.json 2> $out/c2_bench_2ranks_on_1gpu.err
    # (eight ranks of 256 chains fill the one device to its last wave slot: now and then a launch
    # of such a set-up gives up and the run falls back to the host-staged exchange -- tried again)
    for try in 1 2 3; do
        MC3HIP_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --workload c3 --steps 3 --warmup 1 --refreshes 4096 \
            > $out/c3_bench_8ranks_on_1gpu.json 2> $out/c3_bench_8ranks_on_1gpu.err
        grep -q '"recoveries": 0' $out/c3_bench_8ranks_on_1gpu.json && break
    done
    MC3HIP_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --workload c3 --steps 3 --warmup 1 --refreshes 4096 --zlag 0 \
        > $out/c3_bench_8ranks_on_1gpu_strict.json 2> $out/c3_bench_8ranks_on_1gpu_strict.err
    MC3HIP_BENCH_DEVICE=0 MC3HIP_RESIDENT_MAXCOL=1 timeout 900 python bench.py --gpus 4 --workload c5 --steps 2 --warmup 1 \
        > $out/c5_bench_4ranks_on_1gpu.json 2> $out/c5_bench_4ranks_on_1gpu.err
fi
# large ensembles: instruction mix and stall counters of the several-chains-per-wave loops
for cfg in ${BIGCFGS:-c3x4 c3x2mrw c3one}; do
    has big || continue
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$cfg -o $cfg -- \
        python3 tools/bench_configs.py $cfg > $out/kt_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-trace --output-format csv \
        -d $out/pmc1_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc1_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
        SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv \
        -d $out/pmc2_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc2_$cfg.log 2>&1
    timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_MISC \
        SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace \
        --output-format csv -d $out/pmc3_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc3_$cfg.log 2>&1
done
for cfg in c4 c4x2 c5; do
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
if has c4family; then
    # the wavelet configurations after the pipelined kernel (tools/reduce_r04.py patches their
    # lines in profiles/r04_configs.jsonl), and where a slot of it goes
    timeout 900 python tools/bench_configs.py c4 c4wide c4demc c4mrw c4x2 c4x4 c4x2wide c4x2mrw \
        > $out/c4family_configs.jsonl 2> $out/c4family_configs.err
    for cfg in c4 c4wide c4x2 c4x2wide c4x4; do
        echo "== $cfg" >> $out/c4_pipe_diag.txt
        PIPE_DIAG_ITERS=2000 timeout 300 python tools/pipe_diag.py $cfg 2>&1 | grep -v "^{" | tail -4 >> $out/c4_pipe_diag.txt
    done
fi
ls $out
