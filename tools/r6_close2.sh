#!/bin/bash
# Round 6, after the closing re-take (tools/r6_close.sh): what changed SINCE -- the LPIPS plan cache (drop-in rates), the row-walking
# FireNet kernel (config 3) -- re-measured alone: the default bench line and config 3's kernel table.  The headline's kernels and their
# PMC tables are those of the closing re-take.
R=$PWD; O=$R/gpurun_out/r06b; mkdir -p $O
timeout 1400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cp gpurun_out/bench_full.json $O/bench_default_full.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_fire -o k -- python $R/bench.py --sub --config firenet --cpu-frames 0 > $O/bench_under_rocprof_firenet.json 2> $O/rocprof_fire.err
cd $R
python tools/rocpd_stats.py $(ls $O/prof_fire/*.db $O/prof_fire/*/*.db 2>/dev/null | head -1) --md > $O/kernel_stats_firenet.md; rm -rf $O/prof_fire
tail -c 600 $O/bench_default.json; echo; head -8 $O/kernel_stats_firenet.md | cut -c1-140
