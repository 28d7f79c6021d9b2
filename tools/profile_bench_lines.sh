set -e
R=$PWD
mkdir -p gpurun_out/prof
python bench.py --steps 20 --warmup 5 > gpurun_out/prof/r04_bench_C3.json 2> gpurun_out/prof/r04_bench_C3.err
python bench.py --steps 20 --warmup 5 --config C4 > gpurun_out/prof/r04_bench_C4.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --config C2 > gpurun_out/prof/r04_bench_C2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --config C1 --no-cpu-baseline > gpurun_out/prof/r04_bench_C1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --config C5 --no-cpu-baseline > gpurun_out/prof/r04_bench_C5.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --deterministic 1 --no-cpu-baseline > gpurun_out/prof/r04_bench_C3_deterministic.json 2>/dev/null
MI_BA_UNI=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof/r04_bench_C3_general_kernel.json 2>/dev/null
MI_BA_SUPER=0 python bench.py --steps 20 --warmup 5 --config C4 --no-cpu-baseline > gpurun_out/prof/r04_bench_C4_pair_lists.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_tl
rocprofv3 --kernel-trace -d /tmp/p_tl -- python3 $R/bench.py --no-cpu-baseline --config C3 --steps 20 --warmup 5 > /tmp/p_tl.log 2>&1
python3 $R/tools/trial_timeline.py $(find /tmp/p_tl -name "*.db" | head -1) 8 > $R/gpurun_out/prof/r04_trial_timeline_C3.txt
python3 $R/tools/trial_timeline.py $(find /tmp/p_tl -name "*.db" | head -1) 12 >> $R/gpurun_out/prof/r04_trial_timeline_C3.txt
