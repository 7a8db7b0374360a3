# timing ablation of the conv kernels (results are garbage, only the layer times mean something)
for A in ${ABLATES:-0 2 4 6}; do
  echo "== ABLATE $A"
  EVR_ABLATE=$A python bench.py --no-overlap --cpu-frames 0 --profile-filter "" --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(' '.join('%s=%d'%(k,v['us']) for k,v in d['roofline']['layers'].items()))
"
done
