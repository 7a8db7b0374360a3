# The round's closing GPU call after a change to the convolution sources: PMC traffic passes (bench.py's roofline.traffic is gated on
# the source sha), an A/B of the new defaults against the old ones, the default bench line, kernel stats, the full GPU suite.
TAG=${1:-r03f}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
python tools/rocpd_pmc.py $(db $O/pmc_fetch) > $O/pmc_fetch_size.md
python tools/rocpd_pmc.py $(db $O/pmc_write) > $O/pmc_write_size.md
python tools/make_pmc_traffic.py $(db $O/pmc_fetch) $(db $O/pmc_write) "profiles/r03_pmc_fetch_size.md + r03_pmc_write_size.md" > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
rm -f gpurun_out/env_ab.txt
bash tools/env_ab.sh 2 "X=" "EVR_WIDE=1" "EVR_LPIPS_BAND5=-1" "EVR_WIDE=1 EVR_LPIPS_BAND5=-1" > /dev/null 2>&1
cp gpurun_out/env_ab.txt $O/env_ab_defaults.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --sub > $O/bench_under_rocprof.json 2> $O/rocprof.err
cd $R
python tools/rocpd_stats.py $(db $O/prof) --md > $O/kernel_stats.md
rm -rf $O/prof
(time timeout 900 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1
cat $O/env_ab_defaults.txt; tail -4 $O/gputest.log; cat $O/pmc_traffic.log | tail -3; tail -c 400 $O/bench_default.json
