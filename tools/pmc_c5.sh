out=gpurun_out/r02b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c5 -o c5 -- python3 tools/bench_configs.py c5 > $out/kt_c5.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc1_c5 -o c5 -- python3 tools/bench_configs.py c5 > $out/pmc1_c5.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $out/pmc2_c5 -o c5 -- python3 tools/bench_configs.py c5 > $out/pmc2_c5.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d $out/pmc3_c5 -o c5 -- python3 tools/bench_configs.py c5 > $out/pmc3_c5.log 2>&1
tail -2 $out/pmc3_c5.log
ls $out/*
timeout 120 python tools/bench_configs.py c5 > $out/c5_bench.jsonl 2>/dev/null
timeout 200 python tools/ticks.py c5 > $out/c5_ticks.txt 2>&1
[ -x build/exp/mfma_f64_issue_probe ] && timeout 120 ./build/exp/mfma_f64_issue_probe > $out/mfma_f64_issue_probe.txt 2>&1
