#!/bin/bash
# A/B of compile-time variants of the 16-bit-term GEMM on the GPU box: x3_ab.sh <mode> "<flags A>" "<flags B>" ...
mode=$1; shift
i=0
for f in "$@"; do
  bash az-net_amd/tools/ab_build.sh v$i "$f" > /dev/null 2>&1
  echo "== variant $i: $f"
  AZNET_HIP_LIB=/tmp/az_ab_v$i/libaznet_hip.so timeout 300 python tests/dev/x3_probe.py $mode fast 2>&1 | grep -E "search|rows 670|differ"
  i=$((i+1))
done
