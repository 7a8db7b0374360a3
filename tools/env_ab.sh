# Interleaved A/B of run-time switches on one box:  bash tools/env_ab.sh <repeats> "<VAR=val ...>" "<VAR=val ...>" ...
#   each quoted argument is one setting (use "X=" for the default); -> gpurun_out/env_ab.txt
#   columns: frames/s over the 40 timed steps, over the 2.2-s steady state, worst image error of the 4-frame oracle replay
R=$1; shift; mkdir -p gpurun_out
for i in $(seq $R); do for s in "$@"; do env $s python bench.py --sub --cpu-frames 0 --steps 40 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$s |', d['value'], (d.get('steady_state') or {}).get('value'), (d.get('score_parity') or {}).get('image_max_abs_err'))" | tee -a gpurun_out/env_ab.txt; done; done
