#!/bin/bash
# GPU tests + the default bench (compact line on stdout, full object in gpurun_out/)
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r4_gputest.log 2>&1
tail -3 gpurun_out/r4_gputest.log
( time python bench.py ) > gpurun_out/r4_bench_stdout.log 2> gpurun_out/r4_bench_stderr.log
tail -c 5000 gpurun_out/r4_bench_stdout.log | tail -1 > gpurun_out/r4_bench_line.json
wc -c gpurun_out/r4_bench_line.json
cat gpurun_out/r4_bench_line.json
