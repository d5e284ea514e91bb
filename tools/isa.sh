#!/bin/bash
# Development aid: the gfx950 assembly of one kernel of one translation unit.
#   tools/isa.sh k_wavelet_w8.hip 'wavelet_cw_kernelILi3ELi2E' > /tmp/k.s
set -e
unit=$1; pat=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cd "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden \
    --save-temps -c "$root/mc3_amd/csrc/$unit" -o /dev/null >/dev/null 2>&1
f=$(ls *gfx950.s)
start=$(grep -n "^_ZN6mc3hip[0-9]*${pat}.*:" "$f" | head -1 | cut -d: -f1)
awk -v s="$start" 'NR>=s' "$f" | awk '/^\.Lfunc_end/{print; exit} {print}'
rm -rf "$tmp"
