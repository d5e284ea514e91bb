#!/bin/bash
# Development aid: the headline loop's rate (tests/soak.py, exact vs the oracle) with variant
# libraries.   bash tools/c2_variants.sh "256 2048" LIB ...
sizes=$1; shift
for lib in "$@"; do
  for n in $sizes; do
    MC3HIP_LIBRARY=$lib timeout 250 python tests/soak.py ${SOAK_ITERS:-60000} demc $n 2>&1 | tail -2 | grep -o "x [0-9]* chains\|([0-9.]* M/s\|identical: [A-Za-z]*" | tr "\n" " "
  done
  echo " $lib"
done
