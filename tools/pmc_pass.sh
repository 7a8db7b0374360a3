# one rocprofv3 PMC pass over a short bench run: bash tools/pmc_pass.sh <tag> <counters...>
TAG=$1; shift
R=$PWD; mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_$TAG -o p -- python $R/bench.py --cpu-frames 0 --steps 4 --warmup 2 > $R/gpurun_out/pmc_$TAG/bench.json 2> $R/gpurun_out/pmc_$TAG/err.txt
cd $R
python tools/rocpd_pmc.py $(ls gpurun_out/pmc_$TAG/*.db | head -1) > gpurun_out/pmc_$TAG/table.md
grep -E "band|igemm" gpurun_out/pmc_$TAG/table.md | head -40
