#!/bin/bash
# Round 6: the exact-fp32 step's tables again, after its decoders and encoders moved onto the Winograd kernel (the closing re-take's
# fp32 tables described the step before that): kernel-trace stats (two streams), the SQ pass (one stream), FETCH_SIZE / WRITE_SIZE.
R=$PWD; O=$R/gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EVR_FP32=1 rocprofv3 --kernel-trace --stats -d $O/prof_fp32 -o k -- python $R/bench.py --sub --cpu-frames 0 > $O/bench_under_rocprof_fp32.json 2> $O/rocprof_fp32.err
cd $R
python tools/rocpd_stats.py $(ls $O/prof_fp32/*.db $O/prof_fp32/*/*.db 2>/dev/null | head -1) --md > $O/r06_kernel_stats_fp32.md; rm -rf $O/prof_fp32
bash tools/r6_pmc.sh r06e/fp32_pmc > /dev/null 2>&1; cp $O/fp32_pmc/pmc_sq.md $O/r06_pmc_sq_fp32_single_stream.md
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/fp32_$c
  EVR_FP32=1 rocprofv3 --kernel-trace --pmc $c -d $O/fp32_$c -o p -- python $R/bench.py --sub --no-overlap --cpu-frames 0 --parity-frames 1 --steps 4 --warmup 2 > /dev/null 2> $O/fp32_$c.err
  python $R/tools/rocpd_pmc.py $(ls $O/fp32_$c/*.db $O/fp32_$c/*/*.db 2>/dev/null | head -1) | head -14 > $O/r06_pmc_${c}_fp32.md
  rm -rf $O/fp32_$c
done
head -14 $O/r06_kernel_stats_fp32.md | cut -c1-150; head -10 $O/r06_pmc_sq_fp32_single_stream.md | cut -c1-200
