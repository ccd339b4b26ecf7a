import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gtn_applications_amd import engine as E
g = torch.Generator().manual_seed(0)
B, T, C, L = 128, 1000, 100, 44
x = torch.randn(B, T, C, generator=g).cuda()
tgt = torch.randint(C - 2, (B, L), generator=g).tolist()
tg = E.CtcTargets(tgt, x.device)
ws, nll = E.ctc_forward(x, tg, C - 1, 0)
torch.cuda.synchronize()
rej = E.ctc_rejected(ws, B, T, tg.max_len).cpu().numpy()
P, nb = tg.max_len + 1, (T + 15) // 16
o = B * 2 * nb * P * 2; o = (o + 1) & ~1; o += 2 * B * 2 * nb + 2 * B + B
pbad = ws[o:o + 2 * B].view(torch.int32).cpu().numpy()
print("rejected", rej.sum(), "of", B, " pbad sum", pbad.sum())
ws2, nll2 = E.ctc_forward(x, tg, C - 1, 1)
print("max |nll fast - nll log|", (nll - nll2).abs().max().item(), "nll[0]", nll[0].item(), nll2[0].item())
