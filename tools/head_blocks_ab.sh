# A/B of the head kernel's persistent grid (EVR_HEAD_BLOCKS: work-groups walking the 8 x 32 tiles; csrc/conv_misc.hip
# launch_head_conv), back to back on one box:  bash tools/head_blocks_ab.sh [repeats]  ->  gpurun_out/head_blocks_ab.txt
# per setting: the head layer alone (single stream, HIP events) and the two-stream headline of a --sub run.
# profiles/r03_head_blocks_ab.txt is the run that moved the default from 768 to 512 (it also carries two builds that are gone
# again: EVR_HEAD_RPU=2 = both row passes of a wave unrolled together, EVR_HEAD_OCC=3 = a 168-register build, which spills).
O=gpurun_out/head_blocks_ab.txt; mkdir -p gpurun_out
pick() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = (d.get('roofline') or {}).get('layers') or {}
print('$1', 'fps', d['value'], 'steady', (d.get('steady_state') or {}).get('value'), 'err', (d.get('score_parity') or {}).get('image_max_abs_err'), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items() if 'head' in k or 'enc0.conv' in k))"; }
one() { env $1 python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | pick "$1 single" | tee -a $O
        env $1 python bench.py --sub --cpu-frames 0 2>/dev/null | pick "$1 step" | tee -a $O; }
for i in $(seq ${1:-2}); do
for hb in ${HB:-768 512 1024}; do one EVR_HEAD_BLOCKS=$hb; done
done
