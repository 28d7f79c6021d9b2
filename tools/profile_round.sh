#!/bin/bash
# One GPU-box call that refreshes profiles/: kernel trace + the two PMC passes (separate runs: FETCH_SIZE and WRITE_SIZE cannot
# share a pass) for a config, then the summaries.  usage: bash tools/profile_round.sh <tag> <config>   (run from the repo root)
set -e
TAG=$1; CFG=$2; R=$PWD
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_trace /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace -d /tmp/p_trace -- python3 $R/bench.py --no-cpu-baseline --config $CFG > $R/gpurun_out/prof/${TAG}_trace_${CFG}.log 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/p_trace -name "*.db" | head -1) > $R/gpurun_out/prof/${TAG}_kernel_stats_${CFG}.csv
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -- python3 $R/bench.py --no-cpu-baseline --config $CFG --steps 4 --warmup 1 > $R/gpurun_out/prof/${TAG}_fetch_${CFG}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -- python3 $R/bench.py --no-cpu-baseline --config $CFG --steps 4 --warmup 1 > $R/gpurun_out/prof/${TAG}_write_${CFG}.log 2>&1
python3 $R/tools/build_traffic.py $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $CFG $R/gpurun_out/prof/${TAG}_build_traffic_${CFG}.json
cat $R/gpurun_out/prof/${TAG}_build_traffic_${CFG}.json | head -30
