#!/bin/bash
# round 5, second probe: the same 1024 resident rows as ONE 1024-row chain, two 512-row chains, four 256-row chains
export TMPDIR=/tmp
for spec in "1024 1" "512 2" "512 4" "256 4"; do
  set -- $spec
  timeout 300 python tools/sweep_gang.py --batch $1 --engines $2 --kernels --settings '[{}, {"NTTS_XCD_AFFINE": 0}]' 2>&1 | grep '^{' | sed "s/^/B=$1 E=$2 /"
done
