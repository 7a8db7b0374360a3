#!/bin/bash
# FireNet (config 3) on the row-walking 16-channel kernel vs the tile kernel: parity, frames/s, per-kernel durations (single stream).
#   bash tools/r6_firenet_rows.sh [quick]
mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "row_kernel" 2>&1 | tail -3
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], (d.get('steady_state') or {}).get('value'))" $1; }
cd /tmp; export TMPDIR=/tmp
for cfg in "rows0:EVR_C16_ROWS=0" "rows1:EVR_C16_ROWS=1" ${1:+} ; do
  name=${cfg%%:*}; kv=${cfg#*:}
  env $kv timeout 300 python $R/bench.py --config firenet --sub --steps 200 > $O/fn_$name.json 2>$O/fn_$name.err; val $O/fn_$name.json
done
if [ "$1" != "quick" ]; then
for cfg in "r1_2:EVR_C16_ROWS_1=2,1" "r1_2b2:EVR_C16_ROWS_1=2,2" "r1b2:EVR_C16_ROWS_1=1,2"; do
  name=${cfg%%:*}; kv=${cfg#*:}
  env $kv timeout 300 python $R/bench.py --config firenet --sub --steps 200 > $O/fn_$name.json 2>$O/fn_$name.err; val $O/fn_$name.json
done
fi
timeout 300 rocprofv3 --kernel-trace --stats -d $O/fn_rows_prof -o k -- python $R/bench.py --config firenet --sub --steps 100 --no-overlap --cpu-frames 0 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls $O/fn_rows_prof/*.db $O/fn_rows_prof/*/*.db 2>/dev/null | head -1) --md > $O/fn_rows_serial_kernels.md; rm -rf $O/fn_rows_prof
head -8 $O/fn_rows_serial_kernels.md | cut -c1-150
