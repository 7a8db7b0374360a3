#!/bin/bash
# Socket power and shader clock while the default step runs (GPU box):  bash tools/power_probe.sh [tag]
# rocm-smi is sampled every 0.2 s beside `bench.py --sub --steps 600` (about 7 s of steps); idle samples before and after.
R=$PWD; TAG=${1:-r05}; O=$R/gpurun_out/$TAG; mkdir -p $O
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -v "^$" > $O/power_caps.txt
sample() { rocm-smi --showpower --showclocks --showuse --json 2>/dev/null; echo; }
( for i in $(seq 1 10); do sample; sleep 0.2; done ) > $O/power_idle.jsonl
python $R/bench.py --sub --cpu-frames 0 --parity-frames 1 --steps ${STEPS:-600} --warmup 20 "${@:2}" > $O/power_bench.json 2>/dev/null &
BP=$!
sleep ${LEAD:-14}      # (import + plan + warm-up)
( while kill -0 $BP 2>/dev/null; do sample; sleep 0.2; done ) > $O/power_busy.jsonl
wait $BP
python - $O <<'PY'
import json, sys, re
O = sys.argv[1]
def rows(f):
    out = []
    for l in open(f'{O}/{f}'):
        l = l.strip()
        if not l.startswith('{'): continue
        try: d = json.loads(l)
        except Exception: continue
        c = d.get('card0') or next(iter(d.values()))
        pw = next((float(v) for k, v in c.items() if 'ower' in k and 'W' in k and re.match(r'^[\d.]+$', str(v))), None)
        sclk = next((v for k, v in c.items() if k.startswith('sclk')), None)
        m = re.search(r'(\d+)\s*Mhz', str(sclk) or '', re.I)
        use = next((v for k, v in c.items() if 'GPU use' in k), None)
        out.append((pw, int(m.group(1)) if m else None, use))
    return out
idle, busy = rows('power_idle.jsonl'), rows('power_busy.jsonl')
b = json.loads([l for l in open(f'{O}/power_bench.json') if l.startswith('{')][-1])
print(open(f'{O}/power_caps.txt').read())
def stat(name, r):
    p = [x[0] for x in r if x[0] is not None]; c = [x[1] for x in r if x[1] is not None]
    if p: print(f'{name}: {len(r)} samples, power avg {sum(p)/len(p):.0f} W (min {min(p):.0f}, max {max(p):.0f})' + (f', sclk avg {sum(c)/len(c):.0f} MHz (min {min(c)}, max {max(c)})' if c else ''))
stat('idle', idle)
act = [x for x in busy if x[0] is not None and x[0] > 0.6 * max(y[0] for y in busy if y[0] is not None)]
stat('stepping (samples above 60 % of the peak sample)', act)
print('bench under the sampler:', b['value'], 'frames/s,', b['ms_per_step'], 'ms/step')
PY
