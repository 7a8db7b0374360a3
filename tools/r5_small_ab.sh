#!/bin/bash
# A/B of the small-batch switches (GPU box):  bash tools/r5_small_ab.sh [n_seq...]
#   EVR_KSPLIT_SMALL=0  at most 4 runs per tile for every launch (default: 8 for launches of <= 64 tiles)   EVR_WIDE_MIN=600  one tile-form threshold
#   EVR_KSPLIT_EPI4=0  one block per tile in the split-K epilogue (round 4's form)   EVR_KSPLIT=8  up to 8 runs per tile   EVR_KSPLIT=0  no split
mkdir -p gpurun_out
run() {   # run <label> <n_seq> <env...>
  local label=$1 ns=$2; shift 2
  env "$@" python bench.py --sub --n-seq $ns --steps 400 --warmup 20 --cpu-frames 0 --parity-frames 4 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); sp = d.get('score_parity') or {}
print('n_seq $ns', '$label', 'frames/s', d['value'], 'steady', (d.get('steady_state') or {}).get('value'), 'err', sp.get('image_max_abs_err'))" | tee -a gpurun_out/r5_small_ab.txt
}
for ns in ${@:-1 4 8}; do
  run default $ns X=1
  run ks_small_off $ns EVR_KSPLIT_SMALL=0
  run twin_min_600 $ns EVR_WIDE_MIN=600
  run epi4_off $ns EVR_KSPLIT_EPI4=0
  run ks8 $ns EVR_KSPLIT=8
  run nosplit $ns EVR_KSPLIT=0
done
