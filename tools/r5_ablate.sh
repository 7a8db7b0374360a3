#!/bin/bash
# Timing ablation of the split-K band kernel at small batches (GPU box; needs a library built with
# EVR_EXTRA_HIPCC_FLAGS=-DEVR_BAND_ABLATE -- results are garbage, only the kernel times mean anything):
#   bash tools/r5_ablate.sh [n_seq] [tag]
# EVR_ABLATE bits: 1 = no counted waits / barriers, 2 = no DMA requests, 4 = no epilogue, 32 = no fragment reads (ConvLSTM layers only)
# EVR_LIB=<path> selects the ablation build;  ABL="0 32 35" the masks to run
R=$PWD; NS=${1:-1}; TAG=${2:-r05abl}; O=$R/gpurun_out/$TAG; mkdir -p $O
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
cd /tmp && export TMPDIR=/tmp
for A in ${ABL:-0 1 2 3}; do
  rm -rf $O/prof_$A
  EVR_ABLATE=$A rocprofv3 --kernel-trace --stats -d $O/prof_$A -o k -- python $R/bench.py --sub --cpu-frames 0 --parity-frames 1 --n-seq $NS --steps 200 --warmup 10 --no-overlap > $O/bench_$A.json 2> $O/err_$A.txt
  python $R/tools/rocpd_stats.py $(db $O/prof_$A) --md > $O/stats_$A.md
  rm -rf $O/prof_$A
  echo "== EVR_ABLATE=$A"; grep "band_kernel<4, 2, true" $O/stats_$A.md | cut -d'|' -f2-7
done
