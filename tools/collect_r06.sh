#!/bin/bash
# Round 6: collect measurements on the MI355X box into gpurun_out/r06/ (reduced to
# profiles/r06_* by tools/reduce_r06.py).  rocprofv3 is always given python3 itself; PMC passes
# are separate runs with nothing but --pmc (+ --kernel-trace for the kernel names).
#   bash tools/collect_r06.sh [part ...]
# parts: tests bench ranks kt pmc configs c4 c5 c3one e2e c4family midrange
out=gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
parts=${*:-bench}
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }

has tests && { timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log; }
# the bench line as the driver runs it
has bench && { timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/c2_bench.json 2> $out/c2_bench.err; tail -c 600 $out/c2_bench.err; }
has probe && tools/probes/_bin/hbm_read_probe 4 8 > $out/hbm_read_probe.json 2>&1
if has ranks; then
    MC3HIP_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --refreshes 16384 \
        > $out/c2_bench_2ranks_on_1gpu.json 2> $out/c2_bench_2ranks_on_1gpu.err
    tail -c 400 $out/c2_bench_2ranks_on_1gpu.err
fi
# kernel trace of the same workload (shorter: the trace of one step is the same kernel)
has kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- \
    python3 bench.py --no-cpu-baseline --no-e2e --no-extras --steps 4 --warmup 1 > $out/kt_c2.log 2>&1
if has pmc; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_c2_$ctr -o c2 -- \
            python3 bench.py --no-cpu-baseline --no-e2e --no-extras --steps 3 --warmup 1 --refreshes 98304 > $out/pmc_c2_$ctr.log 2>&1
    done
fi
if has configs; then
    timeout 1500 python tools/bench_configs.py ${CONFIGS:-c2 c2snooker c2mrw c3one c3onesnooker c3x2 c3x4 c4 c4wide c4demc c4mrw c4x2 c4x4 c5 q5 p6matrix} \
        > $out/configs.jsonl 2> $out/configs.err
fi
# (round 6: the PMC passes and the kernel trace of C3-in-one-context, C4 and C5 profile the very
# child process of the bench line -- bench.py --config-child: for C5 ONE launch of 8 192
# refreshes behind a warm-up of 1 024, the regime `configs.c5` is measured in, not the
# 300-refresh launch of tools/bench_configs.py that round 5's passes described)
for cfg in c4 c5 c3one; do
    has $cfg || continue
    run="bench.py --config-child $cfg"
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$cfg -o $cfg -- \
        python3 $run > $out/kt_$cfg.log 2>&1
    timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv \
        -d $out/pmc1_$cfg -o $cfg -- python3 $run > $out/pmc1_$cfg.log 2>&1
    timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
        SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv \
        -d $out/pmc2_$cfg -o $cfg -- python3 $run > $out/pmc2_$cfg.log 2>&1
    timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC \
        SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace \
        --output-format csv -d $out/pmc3_$cfg -o $cfg -- python3 $run > $out/pmc3_$cfg.log 2>&1
done
if has e2e; then
    for a in "1e5 512 1 c5" "1e6 512 1 c5" "1e7 256 1 c2"; do
        echo "== tools/e2e_sample.py $a" >> $out/e2e_sample.txt
        timeout 600 python tools/e2e_sample.py $a 2>&1 | head -40 >> $out/e2e_sample.txt
    done
fi
for cfg in c4x2; do
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
        SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace \
        --output-format csv -d $out/pmc3_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg > $out/pmc3_$cfg.log 2>&1
done
if has c4family; then
    timeout 900 python tools/bench_configs.py c4 c4wide c4demc c4mrw c4x2 c4x4 c4x2wide c4x2mrw \
        > $out/c4family_configs.jsonl 2> $out/c4family_configs.err
    for cfg in c4 c4wide; do
        echo "== $cfg" >> $out/c4_pipe_diag.txt
        PIPE_DIAG_ITERS=2000 timeout 300 python tools/pipe_diag.py $cfg 2>&1 | grep -v "^{" | tail -4 >> $out/c4_pipe_diag.txt
    done
fi
has midrange && timeout 1200 python tools/mid_range.py > $out/mid_range.jsonl 2> $out/mid_range.err
ls $out
