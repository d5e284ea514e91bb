#!/bin/sh
# us per hop of the in-launch hand-off across two GPUs (and on one): the first thing to run
# on a multi-GPU box.  Usage: tools/probes/run_xgmi_handoff.sh [devA devB [round trips]]
set -e
here=$(dirname "$0")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/xgmi_handoff "$here/xgmi_handoff.hip"
exec /tmp/xgmi_handoff "$@"
