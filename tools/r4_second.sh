#!/bin/bash
mkdir -p gpurun_out
R=$PWD
python -m pytest tests/test_gpu_eval.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5
python tools/eval_cli_profile.py gpurun_out/r4_eval_cli_profile.txt; head -5 gpurun_out/r4_eval_cli_profile.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4prof -o k -- python $R/bench.py --sub --steps 40 --cpu-frames 0 > $R/gpurun_out/r4_bench_under_rocprof.json 2> $R/gpurun_out/r4_rocprof.err
cd $R
python tools/rocpd_stats.py $(ls gpurun_out/r4prof/*.db gpurun_out/r4prof/*/*.db 2>/dev/null | head -1) --md > gpurun_out/r4_kernel_stats_h3.md
rm -rf gpurun_out/r4prof
head -40 gpurun_out/r4_kernel_stats_h3.md
for c in etnet hyper; do python bench.py --sub --config $c --steps 10 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$c', d['value'], {k: v for k, v in d['roofline'].items() if k in ('kernel','frac','achieved','avg_launch_us','share_of_bracketed_time')}, d.get('roofline_dynamic_filter'))"; done
