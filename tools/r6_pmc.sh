#!/bin/bash
# Round 6 (GPU box): where the wave cycles of the exact-fp32 step go, every kernel alone on the chip (single stream), one PMC pass.
#   bash tools/r6_pmc.sh <tag> [VAR=val ...]   -> gpurun_out/<tag>/pmc_sq.md
R=$PWD; TAG=${1:-r06_pmc}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc
env EVR_FP32=1 "$@" rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS \
  -d $O/pmc -o p -- python $R/bench.py --sub --no-overlap --cpu-frames 0 --parity-frames 1 --steps 4 --warmup 2 > $O/bench.json 2> $O/err.txt
cd $R
python tools/rocpd_pmc.py $(ls $O/pmc/*.db $O/pmc/*/*.db 2>/dev/null | head -1) > $O/pmc_raw.md
rm -rf $O/pmc
python - $O <<'PY' | tee $O/pmc_sq.md
import re, sys
O = sys.argv[1]
rows = {}
for l in open(f'{O}/pmc_raw.md'):
    m = re.match(r'\| `(.*?)` \| (\w+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', l)
    if m:
        rows.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(4)), float(m.group(5)), int(m.group(3)))
print('| kernel | launches | avg us | clock GHz | matrix busy | parked (waitcnt/barrier) | issue-stalled | issuing | VALU instructions per wave quad-cycle (SQ_WAVE_CYCLES counts quad-cycles) | LDS wait |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', (0, 0, 0))[1] * kv[1].get('GRBM_GUI_ACTIVE', (0, 0, 0))[2]):
    if 'GRBM_GUI_ACTIVE' not in d or 'SQ_WAVE_CYCLES' not in d: continue
    g, us, n = d['GRBM_GUI_ACTIVE']
    if us < 20: continue
    wc = d['SQ_WAVE_CYCLES'][0]; b = d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0,))[0]
    wa = d.get('SQ_WAIT_ANY', (0,))[0]; wi = d.get('SQ_WAIT_INST_ANY', (0,))[0]; ai = d.get('SQ_ACTIVE_INST_ANY', (0,))[0]
    va = d.get('SQ_INSTS_VALU', (0,))[0]; wl = d.get('SQ_WAIT_INST_LDS', (0,))[0]
    print(f"| `{k}` | {n} | {us:.1f} | {g / 8 / us / 1e3:.2f} | {b / 1024 / (g / 8):.2f} | {wi / wc:.2f} | {(wa - wi) / wc:.2f} | {ai / wc:.2f} | {va / wc:.3f} | {wl / wc:.3f} |")
PY
