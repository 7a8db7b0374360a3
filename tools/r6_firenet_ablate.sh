#!/bin/bash
# timing ablations of conv3x3_c16_rows_kernel (results garbage): which part of a step costs what
mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06
cd /tmp; export TMPDIR=/tmp
for ab in 0 1 2 4 8 33 6 39 47; do
  EVR_ABLATE=$ab EVR_ABLATE_ALL=1 rocprofv3 --kernel-trace --stats -d $O/ab_$ab -o k -- python $R/bench.py --config firenet --sub --steps 60 --no-overlap --cpu-frames 0 --parity-frames 1 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(ls $O/ab_$ab/*.db $O/ab_$ab/*/*.db 2>/dev/null | head -1) --md > $O/ab_$ab.md; rm -rf $O/ab_$ab
  echo "ablate=$ab: $(grep c16_rows $O/ab_$ab.md | awk -F'|' '{print $2, $5}' | tr '\n' ';')"
done
