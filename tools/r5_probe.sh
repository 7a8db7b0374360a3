#!/bin/bash
# Round-5 measurements (GPU box):  bash tools/r5_probe.sh <section> [tag]
#   small      rocprofv3 kernel-trace tables of the two-stream step at 1 and 4 sequences
#   place      evaluation placement at 64 sequences: step time + per-kernel in-step times for {default gate, gate after dec2,
#              ungated, same stream}  
#   clock      ConvLSTM launches under GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES: default tile rule vs EVR_WIDE=3 (item 7)
R=$PWD; SEC=${1:-small}; TAG=${2:-r05}; O=$R/gpurun_out/$TAG; mkdir -p $O
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <name> <env...> -- <bench args...>
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf $O/prof_$name
  env "${envs[@]}" rocprofv3 --kernel-trace --stats -d $O/prof_$name -o k -- python $R/bench.py --sub --cpu-frames 0 --parity-frames 1 "$@" > $O/bench_$name.json 2> $O/rocprof_$name.err
  python $R/tools/rocpd_stats.py $(db $O/prof_$name) --md > $O/kernel_stats_$name.md
  rm -rf $O/prof_$name
  python - "$O/bench_$name.json" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('value'))
PY
}
case $SEC in
small)
  trace nseq1 X=1 -- --n-seq 1 --steps 300 --warmup 20
  trace nseq1_single X=1 -- --n-seq 1 --steps 300 --warmup 20 --no-overlap
  trace nseq4 X=1 -- --n-seq 4 --steps 200 --warmup 10
  ;;
place)
  trace gate_default X=1 -- --steps 40
  trace gate_dec2 EVR_EVAL_GATE=dec2 -- --steps 40
  trace gate_dec1 EVR_EVAL_GATE=dec1 -- --steps 40
  trace gate_none EVR_EVAL_GATE=none -- --steps 40
  trace same_stream X=1 -- --steps 40 --no-overlap
  ;;
clock)
  for v in default wide3; do
    E=X=1; [ $v = wide3 ] && E=EVR_WIDE=3
    rm -rf $O/clk_$v
    env $E rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/clk_$v -o p -- python $R/bench.py --sub --cpu-frames 0 --steps 6 --warmup 2 --parity-frames 1 > /dev/null 2>&1
    python $R/tools/rocpd_pmc.py $(db $O/clk_$v) > $O/clk_$v.md
    rm -rf $O/clk_$v
    # un-profiled time of the same form (HIP events inside bench.py)
    env $E python $R/bench.py --sub --cpu-frames 0 --steps 40 --parity-frames 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$v', 'frames/s', d['value'], 'ConvLSTM avg us in step', r['avg_launch_us'], 'alone', (r.get('single_stream') or {}).get('avg_launch_us'))" | tee -a $O/traffic_vs_clock.txt
  done
  cd $R
  python - $O <<'PY' | tee -a $O/traffic_vs_clock.txt
import re, sys
O = sys.argv[1]
for v in ('default', 'wide3'):
    rows = {}
    for l in open(f'{O}/clk_{v}.md'):
        m = re.match(r'\| `(.*?)` \| (\w+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', l)
        if m and 'wide_kernel<true' in m.group(1):
            rows.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(4)), float(m.group(5)), int(m.group(3)))
    for k, d in rows.items():
        g, us, n = d['GRBM_GUI_ACTIVE']; b = d['SQ_VALU_MFMA_BUSY_CYCLES'][0]
        print(f"{v:8s} {k}: {n} launches, {us:7.1f} us, clock {g / 8 / us / 1e3:.2f} GHz, MFMA busy {b / 1024 / (g / 8):.2f} of the SIMD cycles")
PY
  ;;
esac
cd $R; ls $O | head -40
