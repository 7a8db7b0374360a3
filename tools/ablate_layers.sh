# Per-layer timing ablation of the convolution kernels (build with EVR_EXTRA_HIPCC_FLAGS=-DEVR_BAND_ABLATE first):
#   bash tools/ablate_layers.sh [masks...]   -> one line per EVR_ABLATE value: layer -> us (single stream, 64 sequences)
# EVR_ABLATE bits: 1 no barriers/waits, 2 no LDS-DMA requests, 4 no epilogue, 8 no border masks, 16 no epilogue-operand loads, 32 no MFMAs (wide kernel)
mkdir -p gpurun_out
for ab in ${@:-0 16 4 2 1 32}; do
  EVR_ABLATE=$ab EVR_ABLATE_ALL=1 python bench.py --sub --no-overlap --profile-filter '' --steps 6 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ablate=$ab', 'fps', d['value'], ' '.join(f\"{k}={v['us']:.0f}\" for k, v in d['roofline']['layers'].items()))" | tee -a gpurun_out/r4_ablate.txt
done
