"""CPU emulation of the matrix-core operand formats, run through the oracle network (test infrastructure only).

Compares, against the oracle's exact fp32 convolutions, the image error after N recurrent frames of
  bf16x3    x = hi + lo (bf16 each), acc = hi*hi + hi*lo + lo*hi                      (3 bf16 MFMAs per 16 k)
  f16mx8    x = f16(x) + fp8((x - f16(x)) * 2^12) * 2^-12, corrections computed in fp8:
            acc = xh*wh + 2^-16 * (fp8(xl*2^12) * fp8(w*2^4) + fp8(x*2^-2) * fp8(wl*2^18))
            (2 f16 MFMAs + 1 MX-scaled fp8 MFMA of twice the K per 32 k = 2/3 of the matrix cycles)
Only convolutions with >= 32 input channels are emulated (the head runs on its own kernel).
    python tools/split_scheme_sim.py [frames] [H] [W]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import oracle.model as om
from evreal_amd import weights as wts

_conv2d, _convT = F.conv2d, F.conv_transpose2d
MODE = ['exact']

def bf16(x): return x.to(torch.bfloat16).to(torch.float32)
def f16(x): return x.clamp(-65504, 65504).to(torch.float16).to(torch.float32)
def f8(x): return x.clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32)

def parts(x, w):
    if MODE[0] == 'bf16x3':
        xh, wh = bf16(x), bf16(w); xl, wl = bf16(x - xh), bf16(w - wh)
        return [(xh, wh, 1.0), (xh, wl, 1.0), (xl, wh, 1.0)]
    if MODE[0] == 'bf16x1':
        return [(bf16(x), bf16(w), 1.0)]
    xh, wh = f16(x), f16(w)
    xl8, x8 = f8((x - xh) * 2.0**12), f8(x * 2.0**-2)
    w8, wl8 = f8(w * 2.0**4), f8((w - wh) * 2.0**18)
    return [(xh, wh, 1.0), (xl8, w8, 2.0**-16), (x8, wl8, 2.0**-16)]

def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    if MODE[0] == 'exact' or w.shape[1] < 32:
        return _conv2d(x, w, b, stride, padding, *a, **k)
    out = None
    for xa, wa, s in parts(x, w):
        y = _conv2d(xa.double(), wa.double(), None, stride, padding, *a, **k) * s
        out = y if out is None else out + y
    if b is not None: out = out + b.double().view(1, -1, 1, 1)
    return out.float()

def convT(x, w, b=None, stride=1, padding=0, output_padding=0, *a, **k):
    if MODE[0] == 'exact' or w.shape[0] < 32:
        return _convT(x, w, b, stride, padding, output_padding, *a, **k)
    out = None
    for xa, wa, s in parts(x, w):
        y = _convT(xa.double(), wa.double(), None, stride, padding, output_padding, *a, **k) * s
        out = y if out is None else out + y
    if b is not None: out = out + b.double().view(1, -1, 1, 1)
    return out.float()

om.F.conv2d, om.F.conv_transpose2d = conv2d, convT

def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    torch.set_num_threads(8)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in wts.synth_state_dict(wts.unet_recurrent_schema(norm='BN'), seed=3).items()}
    rng = np.random.default_rng(5)
    xs = []
    for f in range(frames):
        v = np.zeros((1, 5, H, W), np.float32)
        m = rng.random(v.shape) < 0.2
        v[m] = rng.normal(0, 1.5, m.sum()).astype(np.float32)
        xs.append(torch.from_numpy(v))
    res = {}
    for mode in ['exact', 'bf16x3', 'f16mx8', 'bf16x1']:
        MODE[0] = mode
        net = om.UNetRecurrentOracle(sd, norm='BN', final_activation='sigmoid') if 'sigmoid' in str(sys.argv) else om.UNetRecurrentOracle(sd, norm='BN')
        res[mode] = [net(x).numpy().copy() for x in xs]
    for mode in ['bf16x3', 'f16mx8', 'bf16x1']:
        errs = [np.abs(a - b).max() for a, b in zip(res[mode], res['exact'])]
        print(f'{mode:8s} max|err| first {errs[0]:.2e} last {errs[-1]:.2e} worst {max(errs):.2e}   image range [{res["exact"][-1].min():.3f}, {res["exact"][-1].max():.3f}]')

if __name__ == '__main__':
    main()
