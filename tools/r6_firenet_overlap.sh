#!/bin/bash
# FireNet (config 3): evaluation on a second stream vs on the reconstruction stream, and the kernels' durations in each form.
mkdir -p gpurun_out/r06; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], (d.get('steady_state') or {}).get('value'))" $1; }
python $R/bench.py --config firenet --sub --steps 200 > $O/fn_overlap.json 2>/dev/null; val $O/fn_overlap.json
python $R/bench.py --config firenet --sub --steps 200 --no-overlap > $O/fn_serial.json 2>/dev/null; val $O/fn_serial.json
rocprofv3 --kernel-trace --stats -d $O/fn_serial_prof -o k -- python $R/bench.py --config firenet --sub --steps 200 --no-overlap --cpu-frames 0 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls $O/fn_serial_prof/*.db $O/fn_serial_prof/*/*.db 2>/dev/null | head -1) --md > $O/fn_serial_kernels.md; rm -rf $O/fn_serial_prof
head -22 $O/fn_serial_kernels.md | cut -c1-150
