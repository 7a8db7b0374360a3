#!/bin/bash
# Timing-ablation variants of conv3x3_wide_kernel in the default (h3) arithmetic:  bash tools/ablate_wide.sh build <mask>...   (here, no GPU)
#                                                                                  bash tools/ablate_wide.sh run <mask>...     (on the GPU box)
# mask bits: 1 no waits / barriers in the main loop, 2 no LDS-DMA requests in the loop, 4 no epilogue, 32 no MFMAs (results are garbage)
mode=$1; shift
R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/tools/_bin $R/gpurun_out
if [ "$mode" = build ]; then
  for m in "$@"; do
    ( /opt/rocm/bin/hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/evreal_amd/csrc -Wall -Wno-unused-function -fno-fast-math \
        -DEVR_ARITH=3 -DEVR_WIDE_ABLATE=$m -x hip $R/evreal_amd/csrc/conv.hip -o $R/tools/_bin/conv.h3.ab$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $R/evreal_amd/csrc/_obj/*.o | grep -v 'conv.hip.h3.o') $R/tools/_bin/conv.h3.ab$m.o -o $R/tools/_bin/libevreal_ab$m.so && echo built $m ) &
  done; wait
else
  for m in "$@"; do
    L=$R/evreal_amd/libevreal_hip.so; [ "$m" != 0 ] && L=$R/tools/_bin/libevreal_ab$m.so
    EVR_LIB=$L python $R/bench.py --sub --no-overlap --profile-filter '' --steps 6 --warmup 2 --cpu-frames 0 --parity-frames 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ablate=$m', 'fps', d['value'], ' '.join(f\"{k}={v['us']:.0f}\" for k, v in d['roofline']['layers'].items()))" | tee -a $R/gpurun_out/r4_ablate_wide.txt
  done
fi
