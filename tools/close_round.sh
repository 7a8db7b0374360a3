# The round's closing GPU call: default bench line, kernel stats of the headline run, the full GPU suite  ->  gpurun_out/<tag>/
# (after a change to conv.hip / conv.h / model.cpp / packed.h run tools/profile_round.sh instead: bench.py's roofline.traffic is gated
#  on the sha of those sources, profiles/pmc_traffic.json)
TAG=${1:-r03f}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --sub > $O/bench_under_rocprof.json 2> $O/rocprof.err
cd $R
python tools/rocpd_stats.py $(db $O/prof) --md > $O/kernel_stats.md
rm -rf $O/prof
(time timeout 900 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1
tail -4 $O/gputest.log; head -12 $O/kernel_stats.md | cut -c1-120; tail -c 300 $O/bench_default.json
