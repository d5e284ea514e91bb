mkdir -p gpurun_out/r05d
for L in "" $(ls _variants/*.so) ""; do echo "lib=$L $(MC3HIP_LIBRARY=$L python bench.py --config-child c5 2>/dev/null | tail -1 | cut -c1-400)"; done | tee gpurun_out/r05d/steady.log
