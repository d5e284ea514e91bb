#!/bin/bash
# Development aid: the C4 line of bench.py (--config-child c4) with variant libraries / switches.
#   bash tools/c4_variants.sh "LIB[:ENV=VALUE]" ...
for spec in "$@"; do
  lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env MC3HIP_LIBRARY=$lib $envs timeout 300 python bench.py --config-child c4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', d['us_per_iteration'], d['us_per_iteration_hip_events'], d['kernel_launched'][-40:])"
done
