"""GPU parity of the fine-tune pieces: CE / BCE / CTC losses (forward + gradient w.r.t. the logits) against the
reference's own functions (golden) and torch; fused clip + AdamW against torch.optim.AdamW + clip_grad_norm_."""
import numpy as np
import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu


def test_losses_match_reference_golden():
    from lyricalignment_amd import finetune as ft
    z = load_npz("losses.npz")
    V = int(z["vocab_size"])
    logits = torch.from_numpy(z["logits"]).cuda()
    losses, g = ft.multitask_loss(logits, torch.from_numpy(z["frame_labels"]), None, vocab_size=V)
    l = losses.cpu().numpy()
    np.testing.assert_allclose(l[0] + l[1], z["ce"], rtol=2e-6)                  # compute_ce_loss = word CE + silence BCE
    np.testing.assert_allclose(g.cpu().numpy(), z["g_ce"], rtol=0, atol=2e-7)
    losses, g = ft.multitask_loss(logits, None, torch.from_numpy(z["labels"]), vocab_size=V)
    np.testing.assert_allclose(losses.cpu().numpy()[2], z["ctc"], rtol=2e-6)
    np.testing.assert_allclose(g.cpu().numpy(), z["g_ctc"], rtol=0, atol=1e-5)   # fp32 log-space alpha/beta in both; order of log-adds differs
    both, g = ft.multitask_loss(logits, torch.from_numpy(z["frame_labels"]), torch.from_numpy(z["labels"]), vocab_size=V, scale=0.125)
    np.testing.assert_allclose(g.cpu().numpy(), 0.125 * (z["g_ce"] + z["g_ctc"]), rtol=0, atol=2e-6)   # loss / accum_grad_steps


@pytest.mark.parametrize("B,T,V,Ls", [(2, 300, 500, [26, 9]), (3, 120, 2000, [40, 1, 17]), (2, 1500, 21128, [26, 11])])
def test_losses_match_torch(B, T, V, Ls):
    from lyricalignment_amd import finetune as ft
    rs = np.random.RandomState(B * T)
    logits = torch.from_numpy((rs.randn(B, T, V + 1) * 2).astype(np.float32))
    Lmax = max(Ls)
    labels = torch.full((B, Lmax), -100, dtype=torch.long)
    for b, L in enumerate(Ls):
        lab = rs.randint(1, V, size=L)
        if L > 3:
            lab[2] = lab[1]
        labels[b, :L] = torch.from_numpy(lab)
    fl = torch.full((B, T - 7), -100, dtype=torch.long)
    for b in range(B):
        for k in range(10):
            s = rs.randint(0, T - 30)
            fl[b, s:s + 12] = int(rs.randint(1, V))
    ref = logits.double().requires_grad_(True)   # float64 torch reference: fp32 CTC recursions lose 2-3 digits at T >= 300
    flp = ft.pad_frame_labels(fl, T).clone()
    tgt = flp.clone(); tgt[tgt != -100] -= 1
    ce = torch.nn.functional.cross_entropy(ref[:, :, 1:V].transpose(1, 2), tgt)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(ref[:, :, V], (flp == -100).double())
    lsm = torch.nn.functional.log_softmax(ref[:, :, :V], dim=2).transpose(0, 1)
    ctc = torch.nn.functional.ctc_loss(lsm, labels, torch.full((B,), T, dtype=torch.long), (labels != -100).sum(1))
    (ce + bce + ctc).backward()
    losses, g = ft.multitask_loss(logits.cuda(), fl, labels, vocab_size=V)
    l = losses.cpu().numpy()
    np.testing.assert_allclose(l, [ce.item(), bce.item(), ctc.item()], rtol=2e-5)
    np.testing.assert_allclose(g.cpu().numpy(), ref.grad.numpy(), rtol=0, atol=2e-6)


def test_fused_clip_adamw_matches_torch():
    from lyricalignment_amd import finetune as ft
    g = torch.Generator().manual_seed(0)
    p1, p2 = torch.randn(100003, generator=g), torch.randn(5000, generator=g)
    ref1, ref2 = torch.nn.Parameter(p1.clone()), torch.nn.Parameter(p2.clone())
    opt = torch.optim.AdamW([{"params": [ref1], "lr": 5e-3}, {"params": [ref2], "lr": 5e-6}], weight_decay=1e-5)
    mine = ft.FlatAdamW([{"params": p1.clone().cuda(), "lr": 5e-3}, {"params": p2.clone().cuda(), "lr": 5e-6}], weight_decay=1e-5)
    for step in range(4):
        g1, g2 = torch.randn(100003, generator=g) * (3.0 if step % 2 else 0.001), torch.randn(5000, generator=g)
        ref1.grad, ref2.grad = g1.clone(), g2.clone()
        total = torch.nn.utils.clip_grad_norm_([ref1, ref2], 1.0)
        opt.step()
        ss = mine.step([g1.cuda(), g2.cuda()], max_norm=1.0)
        np.testing.assert_allclose(float(ss.item()) ** 0.5, float(total), rtol=1e-5)
        np.testing.assert_allclose(mine.groups[0]["params"].cpu().numpy(), ref1.detach().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(mine.groups[1]["params"].cpu().numpy(), ref2.detach().numpy(), rtol=0, atol=2e-6)
