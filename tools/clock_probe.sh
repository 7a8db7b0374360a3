#!/bin/bash
# Shader clock of the ConvLSTM kernel on real data vs on zero tensors (the no-store timing variant of tools/ablate_wide.sh leaves every
# activation zero): GRBM_GUI_ACTIVE / 8 XCDs / kernel time.   bash tools/ablate_wide.sh build 64 (here) ; bash tools/clock_probe.sh (GPU box)
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in real zero; do
  L=$R/evreal_amd/libevreal_hip.so; [ $v = zero ] && L=$R/tools/_bin/libevreal_ab64.so
  rm -rf $R/gpurun_out/clk_$v
  EVR_LIB=$L rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/clk_$v -o p -- python $R/bench.py --sub --cpu-frames 0 --steps 6 --warmup 2 --parity-frames 1 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/clk_$v/*.db $R/gpurun_out/clk_$v/*/*.db 2>/dev/null | head -1) > $R/gpurun_out/clk_$v.md
  rm -rf $R/gpurun_out/clk_$v
done
cd $R
python - <<'PY' | tee gpurun_out/r4_clock_probe.txt
import re
for v in ('real', 'zero'):
    rows = {}
    for l in open(f'gpurun_out/clk_{v}.md'):
        m = re.match(r'\| `(.*?)` \| (\w+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', l)
        if m and 'wide_kernel<true' in m.group(1):
            rows.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(4)), float(m.group(5)))
    for k, d in rows.items():
        g, us = d['GRBM_GUI_ACTIVE']; b = d['SQ_VALU_MFMA_BUSY_CYCLES'][0]
        print(f"{v:5s} {k}: {us:7.1f} us, clock {g / 8 / us / 1e3:.2f} GHz, MFMA busy {b / 1024 / (g / 8):.2f} of the SIMD cycles")
PY
