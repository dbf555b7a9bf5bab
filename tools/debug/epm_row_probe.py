import os, sys, torch
sys.path.insert(0, "/root/repo")
from spatiotemporalentropymodel_amd import functional as F
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, C, K in (("EPM.0", 1152, 768), ("EPM.2", 768, 576), ("EPM.4", 576, 384)):
    B, H, W = 16, 16, 16
    x = torch.randn(B, C, H, W, device=dev); dy = torch.randn(B, K, H, W, device=dev)
    xp, dyp = F.F16Planes.split(x), F.F16Planes.split(dy)
    ref = None
    for sel in (0, 4):
        for split in (0, 4, 8, 16):
            with F.tuning(wg3_row=sel, wg3_split=split):
                s, el = F.wgrad_f16x3_plan((B, C, H, W), K, 1, 1, 0)
                dwp = torch.empty(el, device=dev)
                F.conv2d_wgrad_f16x3(xp, dyp, K, 1, 1, 0, dwp, s)
                d = dwp.view(s, 1, K, C).sum(0)
                if ref is None: ref = d
                err = float((d - ref).abs().max() / ref.abs().max())
                t = timeit(lambda: F.conv2d_wgrad_f16x3(xp, dyp, K, 1, 1, 0, dwp, s))
            print(f"{name} form {sel} split {split}->{s}: {t:6.1f} us  diff {err:.1e}")
