import torch, time, sys
sys.path.insert(0, '.')
from evreal_amd.prepost import Metrics
m = Metrics()
a = torch.rand((64, 260, 346), device='cuda'); b = torch.rand((64, 260, 346), device='cuda')
for _ in range(5): out = m(a, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): out = m(a, b)
e1.record(); torch.cuda.synchronize()
print('metrics us', round(e0.elapsed_time(e1) / 50 * 1e3, 1), out[0].cpu().numpy().tolist())
