#!/bin/bash
# kernel-trace summary of one bench run: bash tools/kstats.sh <config> <out.csv>   (from the repo root, on the GPU box)
set -e
CFG=$1; OUT=$2; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_k
rocprofv3 --kernel-trace -d /tmp/p_k -- python3 $R/bench.py --no-cpu-baseline --config $CFG --steps 20 --warmup 5 > /tmp/p_k.log 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/p_k -name "*.db" | head -1) > $R/$OUT
