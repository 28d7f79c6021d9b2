#!/bin/bash
# SQ counter passes over the bench (own runs, --pmc only: no trace options beside them).  usage: bash tools/profile_pmc.sh <tag> [bench args...]
# Pass a: where the waves' cycles go (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES); pass b: instruction counts and the LDS.
set -e
TAG=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_pmc_a /tmp/p_pmc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d /tmp/p_pmc_a -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $R/gpurun_out/prof/${TAG}_pmc_a.log 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/p_pmc_a -name "*.db" | head -1) --pmc > $R/gpurun_out/prof/${TAG}_pmc_cycles.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA -d /tmp/p_pmc_b -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $R/gpurun_out/prof/${TAG}_pmc_b.log 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/p_pmc_b -name "*.db" | head -1) --pmc > $R/gpurun_out/prof/${TAG}_pmc_insts.csv
head -4 $R/gpurun_out/prof/${TAG}_pmc_cycles.csv $R/gpurun_out/prof/${TAG}_pmc_insts.csv
