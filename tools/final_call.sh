bash tools/final_run.sh > gpurun_out/r03f/final_run.out 2>&1
for g in res0.conv2 res1.conv2 enc2.rec dec0 res0.conv2 res1.conv2; do EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('gate $g', d['value'], (d.get('steady_state') or {}).get('value'))" | tee -a gpurun_out/r03f/gate_sweep.txt; done
(time timeout 1000 python -m pytest tests -m gpu -x -q) > gpurun_out/r03f/gputest.log 2>&1; tail -4 gpurun_out/r03f/gputest.log
tail -20 gpurun_out/r03f/final_run.out | cut -c1-300
