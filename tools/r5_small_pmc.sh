#!/bin/bash
# Clock and pipe shares of the small-batch launches (GPU box):  bash tools/r5_small_pmc.sh [n_seq] [tag]
# three PMC passes over `bench.py --sub --n-seq N --no-overlap`; per kernel: time, clock (GRBM_GUI_ACTIVE / 8 XCDs / time), matrix-busy share
R=$PWD; NS=${1:-1}; TAG=${2:-r05pmc1}; O=$R/gpurun_out/$TAG; mkdir -p $O
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <name> <counters...>
  local name=$1; shift
  rm -rf $O/p_$name
  rocprofv3 --kernel-trace --pmc "$@" -d $O/p_$name -o p -- python $R/bench.py --sub --cpu-frames 0 --parity-frames 1 --n-seq $NS --steps ${STEPS:-8} --warmup 2 ${EXTRA---no-overlap} > /dev/null 2> $O/err_$name.txt
  python $R/tools/rocpd_pmc.py $(db $O/p_$name) > $O/pmc_$name.md
  rm -rf $O/p_$name
}
# (three counters per pass: larger sets crashed rocprofv3 on these launch-dense runs)
pass sq GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass valu SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
pass wait SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
cd $R
python - $O <<'PY'
import re, sys
O = sys.argv[1]
rows = {}
for f in ('sq', 'lds', 'valu', 'wait'):
    try: lines = list(open(f'{O}/pmc_{f}.md'))
    except OSError: continue
    for l in lines:
        m = re.match(r'\| `(.*?)` \| (\w+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', l)
        if m: rows.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(4)), float(m.group(5)), int(m.group(3)))
for k, d in rows.items():
    if 'GRBM_GUI_ACTIVE' not in d or not ('conv' in k or 'head' in k): continue
    g, us, n = d['GRBM_GUI_ACTIVE']; cyc = g / 8
    out = f"{k[:72]:72s} n={n:5d} {us:7.1f} us  clock {cyc / us / 1e3:.2f} GHz"
    for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES', 'SQ_INSTS_LDS', 'SQ_INSTS_VALU'):
        if c in d: out += f"  {c[3:]}={d[c][0]:.3g}"
    print(out)
PY
