import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lyricalignment_amd import finetune as ft
F = torch.nn.functional
torch.manual_seed(0)
B, T, V = 2, 1500, 21128
for scale in (0.05, 1.0):
    logits = (torch.randn(B, T, V + 1) * scale).cuda()
    for labels in (torch.tensor([[5, 9, 9, 300, -100], [7, 3, 402, -100, -100]]), torch.tensor([[2769, 4638, 4638, 21127, 8000, 1], [15000, 671, 671, -100, -100, -100]])):
        l3, dlog = ft.multitask_loss(logits, None, labels, vocab_size=V, scale=0.5)
        for dt in (torch.float64, torch.float32):
            x = logits.to(dt).clone().requires_grad_(True)
            lsm = F.log_softmax(x[:, :, :V], dim=2).transpose(0, 1)
            loss = F.ctc_loss(lsm, labels.cuda(), torch.full((B,), T, dtype=torch.long, device="cuda"), (labels != -100).sum(1).cuda())
            (loss * 0.5).backward()
            g = x.grad
            d = (dlog.double() - g.double())
            print(f"scale {scale} labels max {int(labels.max())} ref {dt}: loss hip {float(l3[2]):.6f} torch {float(loss):.6f}; grad max|diff| {float(d.abs().max()):.3e} vs max|g| {float(g.abs().max()):.3e}; rel L2 {float(d.norm() / g.double().norm()):.3e}")
