#!/bin/bash
# round 5, third probe: chains started out of phase; tall tiles under the fp8 Nano-sized gang (four 512-row chains); repeatability of the B = 256 winners
export TMPDIR=/tmp
timeout 300 python tools/sweep_gang.py --mid 60 --steps 16 --reps 2 --stagger 250 500 1000 --settings '[{"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0}]' 2>&1 | grep '^{' | sed "s/^/stagger /"
timeout 400 python tools/sweep_gang.py --config nano-fp8 --batch 512 --kernels --settings '[{}, {"NTTS_XCD_AFFINE": 0}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0, "NTTS_GU_TILE": 3}, {"NTTS_TALL": 2, "NTTS_XCD_AFFINE": 0}, {}]' 2>&1 | grep '^{' | sed "s/^/nano-fp8 B=512 /"
timeout 400 python tools/sweep_gang.py --settings '[{}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0}, {"NTTS_XCD_AFFINE": 0}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0, "NTTS_GU_TILE": 3}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0}, {}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0, "NTTS_HEAD_TILE": 2}]' 2>&1 | grep '^{' | sed "s/^/air B=256 /"
