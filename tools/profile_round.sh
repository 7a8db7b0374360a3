# Everything profiles/ quotes, in one call:  bash tools/profile_round.sh <tag>
# (bench.py --sub = headline + roofline + a 4-frame oracle comparison: no sub-runs, no small-batch passes to pollute the per-kernel averages)
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cp gpurun_out/bench_full.json $O/bench_default_full.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --sub --steps 40 > $O/bench_under_rocprof.json 2> $O/rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o s -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_sq.json 2> $O/pmc_sq.err
# the other arithmetic modes and configurations: kernel-trace stats only
EVR_ARITH=mx rocprofv3 --kernel-trace --stats -d $O/prof_mx8 -o k -- python $R/bench.py --sub --cpu-frames 0 > $O/bench_under_rocprof_mx8.json 2> $O/rocprof_mx8.err
EVR_ARITH=mx6 rocprofv3 --kernel-trace --stats -d $O/prof_mx6 -o k -- python $R/bench.py --sub --cpu-frames 0 > $O/bench_under_rocprof_mx6.json 2> $O/rocprof_mx6.err
EVR_FP32=1 rocprofv3 --kernel-trace --stats -d $O/prof_fp32 -o k -- python $R/bench.py --sub --cpu-frames 0 > $O/bench_under_rocprof_fp32.json 2> $O/rocprof_fp32.err
rocprofv3 --kernel-trace --stats -d $O/prof_640 -o k -- python $R/bench.py --sub --sensor 640x480 --cpu-frames 0 > $O/bench_under_rocprof_640x480.json 2> $O/rocprof_640.err
rocprofv3 --kernel-trace --stats -d $O/prof_fire -o k -- python $R/bench.py --sub --config firenet --cpu-frames 0 > $O/bench_under_rocprof_firenet.json 2> $O/rocprof_fire.err
rocprofv3 --kernel-trace --stats -d $O/prof_hyper -o k -- python $R/bench.py --sub --config hyper --cpu-frames 0 > $O/bench_under_rocprof_hyper.json 2> $O/rocprof_hyper.err
rocprofv3 --kernel-trace --stats -d $O/prof_color -o k -- python $R/bench.py --sub --config color --steps 8 --parity-frames 1 > $O/bench_under_rocprof_color.json 2> $O/rocprof_color.err
cd $R
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python tools/rocpd_stats.py $(db $O/prof) --md > $O/kernel_stats.md
for m in mx8 mx6 fp32 640 fire hyper color; do python tools/rocpd_stats.py $(db $O/prof_$m) --md > $O/kernel_stats_$m.md; done
python tools/rocpd_pmc.py $(db $O/pmc_fetch) > $O/pmc_fetch_size.md
python tools/rocpd_pmc.py $(db $O/pmc_write) > $O/pmc_write_size.md
python tools/rocpd_pmc.py $(db $O/pmc_sq) > $O/pmc_sq.md
python tools/make_pmc_traffic.py $(db $O/pmc_fetch) $(db $O/pmc_write) "profiles/${TAG}_pmc_fetch_size.md + ${TAG}_pmc_write_size.md" > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/prof $O/prof_* $O/pmc_fetch $O/pmc_write $O/pmc_sq        # the rocpd databases are large; the tables above are what is kept
tail -c 1500 $O/bench_default.json; head -14 $O/kernel_stats.md; grep -i "wide\|band" $O/pmc_fetch_size.md $O/pmc_write_size.md | head
