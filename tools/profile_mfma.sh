#!/bin/bash
# counters of the opt-in matrix-pipe build variant (k_build_mfma) beside the vector form: one PMC pass each (SQ block only)
# usage: bash tools/profile_mfma.sh <tag>     (from the repo root, on the GPU box)
set -e
TAG=$1; R=$PWD
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for v in mfma vector; do
  rm -rf /tmp/p_pmc_$v
  if [ $v = mfma ]; then export MI_BA_MFMA=1; else unset MI_BA_MFMA; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/p_pmc_$v -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $R/gpurun_out/prof/${TAG}_pmc_${v}.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find /tmp/p_pmc_$v -name "*.db" | head -1) --pmc > $R/gpurun_out/prof/${TAG}_pmc_build_${v}_C3.csv
  head -3 $R/gpurun_out/prof/${TAG}_pmc_build_${v}_C3.csv
done
