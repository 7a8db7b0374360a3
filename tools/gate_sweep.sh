# Where the evaluation stream of frame t may start inside frame t+1 (EVR_EVAL_GATE, evreal_amd/pipeline.py): interleaved A/B on one box.
#   bash tools/gate_sweep.sh [repeats] [gates...]  ->  gpurun_out/gate_sweep.txt   (value = the 20 timed steps, second figure = 2.2-s steady state)
R=${1:-3}; shift; G=${@:-res0.conv2 res1.conv2 dec0 dec1}
mkdir -p gpurun_out
for i in $(seq $R); do for g in $G; do EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('gate $g', d['value'], (d.get('steady_state') or {}).get('value'), (d.get('score_parity') or {}).get('image_max_abs_err'))" | tee -a gpurun_out/gate_sweep.txt; done; done
