set -e
MI_BA_TWO_LEVEL=1 timeout -k 10 500 python -m pytest tests/test_engine_gpu.py -x -q -k "solve_and_backsub or full_solve or tight_pcg or fixed_cameras or long or two_level" 2>&1 | tail -2
for tl in 0 1; do for c in C3 C4; do
echo "== $c two_level $tl"
MI_BA_TWO_LEVEL=$tl timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --config $c | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernel_ms_per_step',{})
print(d['value'], d['ms_per_step'], k['build'], k['pcg'], d.get('pcg_iters_per_step'))"
done; done
R=$PWD; cd /tmp && export TMPDIR=/tmp && MI_BA_TWO_LEVEL=1 rocprofv3 --kernel-trace -d /tmp/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 --config C4 > $R/gpurun_out/tl_prof.log 2>&1 && DB=$(find /tmp/prof -name "*.db" | head -1) && python3 $R/tools/rocpd_summary.py $DB > $R/gpurun_out/tl_kernels_C4.csv
