mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], (d.get('steady_state') or {}).get('value'))" $1; }
for n in 8 16 24 32 48 64 96; do timeout 300 python $R/bench.py --config firenet --sub --steps 200 --n-seq $n --cpu-frames 0 > $O/fn_n$n.json 2>$O/fn_n$n.err; val $O/fn_n$n.json; done
