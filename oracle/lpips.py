"""Oracle: LPIPS (AlexNet backbone, v0.1) on torch-CPU.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference obtains the metric from pyiqa (`pyiqa.create_metric('lpips')`,
utils/eval_metrics.py:115) on inputs built by cv2torch(img, num_ch=3) (utils/eval_utils.py:46-54); pyiqa is not in
the reference tree, not installed, and downloads its weights at run time.  This restates the published algorithm
(Zhang et al., CVPR 2018; richzhang/PerceptualSimilarity v0.1, which pyiqa wraps), with pyiqa's state-dict names:
inputs in [0,1] -> 2x-1 -> (x - shift)/scale -> AlexNet relu1..relu5 -> channel-unit-normalised features (eps 1e-10)
-> squared difference -> non-negative 1x1 'lin' weights -> spatial mean -> sum over layers.
"""
import torch
import torch.nn.functional as F

SHIFT = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
SCALE = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)
CONVS = [("net.slice1.0", 4, 2), ("net.slice2.3", 1, 2), ("net.slice3.6", 1, 1), ("net.slice4.8", 1, 1),
         ("net.slice5.10", 1, 1)]


def features(sd, x):
    """x: [n,3,H,W] already scaled.  Returns the five relu outputs of torchvision's alexnet.features."""
    outs = []
    for i, (name, stride, pad) in enumerate(CONVS):
        if i in (1, 2):
            x = F.max_pool2d(x, kernel_size=3, stride=2)
        x = torch.relu(F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], stride=stride, padding=pad))
        outs.append(x)
    return outs


def lpips(sd, img, ref):
    """img, ref: float32 [n,H,W] in [0,1] (gray).  Returns [n] float64 scores."""
    sd = {k: torch.as_tensor(v).float() for k, v in sd.items()}
    prep = lambda a: ((2 * torch.as_tensor(a).float().unsqueeze(1).repeat(1, 3, 1, 1) - 1) - SHIFT) / SCALE
    f0, f1 = features(sd, prep(img)), features(sd, prep(ref))
    total = torch.zeros(len(img), dtype=torch.float64)
    for l, (a, b) in enumerate(zip(f0, f1)):
        an = a / (torch.sqrt(torch.sum(a ** 2, dim=1, keepdim=True)) + 1e-10)
        bn = b / (torch.sqrt(torch.sum(b ** 2, dim=1, keepdim=True)) + 1e-10)
        d = (an - bn) ** 2
        w = sd[f'lin{l}.model.1.weight'].view(1, -1, 1, 1)
        total += (d * w).sum(dim=1).double().mean(dim=(1, 2))
    return total.numpy()
