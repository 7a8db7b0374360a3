"""The robust percentile normalization alone on 64 sigmoid-range frames of 346x260 (HIP events): python tools/pct_bench.py"""
import sys, torch
sys.path.insert(0, '.')
from evreal_amd.prepost import post_process_normalization
g = torch.Generator(device='cuda'); g.manual_seed(0)
src = torch.sigmoid(torch.randn((64, 260, 346), device='cuda', generator=g) * 1.5)
a = src.clone()
for _ in range(5): a.copy_(src); post_process_normalization(a, 'robust')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(50):
    a.copy_(src); e0.record(); post_process_normalization(a, 'robust'); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
print('robust norm (select + apply) us', round(tot / 50 * 1e3, 1), 'checksum', float(a.double().sum()))
