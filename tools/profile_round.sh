# Everything profiles/ quotes, in one call:  bash tools/profile_round.sh <tag>
# (bench.py --sub = headline + roofline only: no CPU leg, no sub-runs, no small-batch passes to pollute the per-kernel averages)
TAG=${1:-r02}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --sub > $O/bench_under_rocprof.json 2> $O/rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --sub --steps 4 --warmup 2 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --sub --steps 4 --warmup 2 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o s -- python $R/bench.py --sub --steps 4 --warmup 2 > $O/pmc_sq.json 2> $O/pmc_sq.err
cd $R
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python tools/rocpd_stats.py $(db $O/prof) --md > $O/kernel_stats.md
python tools/rocpd_pmc.py $(db $O/pmc_fetch) > $O/pmc_fetch_size.md
python tools/rocpd_pmc.py $(db $O/pmc_write) > $O/pmc_write_size.md
python tools/rocpd_pmc.py $(db $O/pmc_sq) > $O/pmc_sq.md
python tools/make_pmc_traffic.py $(db $O/pmc_fetch) $(db $O/pmc_write) "profiles/${TAG}_pmc_fetch_size.md + ${TAG}_pmc_write_size.md" > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sq        # the rocpd databases are large; the tables above are what is kept
tail -c 1500 $O/bench_default.json; head -14 $O/kernel_stats.md; grep band $O/pmc_fetch_size.md $O/pmc_write_size.md | head
