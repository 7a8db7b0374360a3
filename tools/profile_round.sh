# Everything profiles/ quotes, in one call:  bash tools/profile_round.sh <tag>
TAG=${1:-r01}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --cpu-frames 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --cpu-frames 0 --steps 4 --warmup 2 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --cpu-frames 0 --steps 4 --warmup 2 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o s -- python $R/bench.py --cpu-frames 0 --steps 4 --warmup 2 > $O/pmc_sq.json 2> $O/pmc_sq.err
cd $R
python tools/rocpd_stats.py $(ls $O/prof/*.db | head -1) --md > $O/kernel_stats.md
python tools/rocpd_pmc.py $(ls $O/pmc_fetch/*.db | head -1) > $O/pmc_fetch_size.md
python tools/rocpd_pmc.py $(ls $O/pmc_write/*.db | head -1) > $O/pmc_write_size.md
python tools/rocpd_pmc.py $(ls $O/pmc_sq/*.db | head -1) > $O/pmc_sq.md
tail -c 1500 $O/bench_default.json; head -12 $O/kernel_stats.md; grep band $O/pmc_fetch_size.md $O/pmc_write_size.md
