#!/bin/bash
# Development aid: C5 in the bench line's long launch (bench.py --config-child c5: one launch of
# 8 192 refreshes, repeats to +-0.02 us on a box) for the product library and for every variant
# library in _variants/ (k_resident.hip compiled with an experiment's macro and linked with the
# product's other objects), the product first and last.   gpurun -- 'bash tools/run_variants.sh'
mkdir -p gpurun_out/r05d
for L in "" $(ls _variants/*.so 2>/dev/null) ""; do
    echo "lib=$L $(MC3HIP_LIBRARY=$L python bench.py --config-child c5 2>/dev/null | tail -1 | cut -c1-400)"
done | tee gpurun_out/r05d/steady.log
