#!/bin/bash
# Round 6, last GPU call: conv.hip gained the row-walking FireNet kernel after the closing re-take, so the source digest that ties
# profiles/pmc_traffic.json to the build changed and bench.py reported roofline.traffic = null.  The headline kernels are untouched;
# this re-takes exactly the two PMC passes the traffic figure comes from (FETCH_SIZE, WRITE_SIZE), then the default bench line.
R=$PWD; TAG=r06; O=$R/gpurun_out/r06c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --sub --steps 4 --warmup 2 --cpu-frames 0 > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python tools/rocpd_pmc.py $(db $O/pmc_fetch) > $O/pmc_fetch_size.md
python tools/rocpd_pmc.py $(db $O/pmc_write) > $O/pmc_write_size.md
python tools/make_pmc_traffic.py $(db $O/pmc_fetch) $(db $O/pmc_write) "profiles/${TAG}_pmc_fetch_size.md + ${TAG}_pmc_write_size.md" > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cp gpurun_out/bench_full.json $O/bench_default_full.json
tail -c 400 $O/bench_default.json; echo; cat $O/pmc_traffic.json | head -c 600
