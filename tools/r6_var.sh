# Round 6 (GPU box): timing experiments on the Winograd kernel (EVR_WINO_VAR; results of VAR & 2 are garbage) -- per-layer times, single stream
tag=${1:-r06_var}; out=gpurun_out/$tag; mkdir -p $out; shift
layers() { python bench.py --sub --no-overlap --profile-filter '' --steps 5 --warmup 2 --cpu-frames 0 --parity-frames 1 2>$out/err_$1.txt | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items() if 'rec' in k or 'res' in k))"; }
for v in "$@"; do
  ( export EVR_FP32=1 $v; layers "$v" ) | tee -a $out/var_times.txt
done
