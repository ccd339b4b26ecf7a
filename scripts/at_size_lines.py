"""One line each for the two criteria the BASELINE configurations do not name, at a size (GPU box):
  STC              T=1000, 100 selected classes (200 augmented columns), B=128, L=44 -- STCLoss fwd + bwd, and the STC module
                   (the torch-side alphabet augmentation of stc.py:199-220 included) on [T, B, C] log-probabilities
  ConvTransduce1D  configs/iamdb/convtrans.json: 200 word pieces, kernel 7, stride 4, scale sqrt, batch 8, 400 frames
-> JSON lines (ms per step = fwd + bwd, HIP events around the loop; bytes = what the criterion must read and write once)."""
import json, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd.criterions import stc, transducer as tr

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def timed(fn, steps=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def main():
    rs = np.random.RandomState(0)
    B, T, Cp, L = 128, 1000, 100, 44
    x = torch.log_softmax(torch.tensor(rs.randn(B, T, 2 * Cp).astype(np.float32)), 2).cuda().requires_grad_(True)
    targets = [rs.randint(1, Cp, size=L).tolist() for _ in range(B)]

    def stc_fn():
        x.grad = None
        stc.STCLoss(x, targets, 0.3, "mean").backward()

    ms = timed(stc_fn)
    alg = 8 * B * T * 2 * Cp
    print(json.dumps({"workload": f"STCLoss fwd+bwd T={T} columns={2 * Cp} B={B} L={L} (stc.py:67-129)", "ms_per_step": ms,
                      "value": B / ms * 1e3, "unit": "utt/s", "algorithmic_bytes": alg, "achieved_GBps": alg / ms / 1e6,
                      "frac_of_8TBps": alg / ms / 1e6 / 8000}))
    lp = torch.log_softmax(torch.tensor(rs.randn(T, B, Cp).astype(np.float32)), 2).cuda().requires_grad_(True)
    mod = stc.STC(0, 0.5, 0.1, 1000, "mean")

    def stc_mod():
        lp.grad = None
        mod(lp, targets).backward()

    ms = timed(stc_mod)
    print(json.dumps({"workload": f"STC module fwd+bwd (alphabet augmentation + STCLoss) T={T} C={Cp} B={B} L={L} (stc.py:174-221)",
                      "ms_per_step": ms, "value": B / ms * 1e3, "unit": "utt/s"}))

    with open(os.path.join(ROOT, "benchmarks", "word_pieces_tokens_1000.txt")) as f:
        tokens = sorted(l.strip() for l in f)
    graphemes = sorted(set(c for t in tokens for c in t))
    g2i = {c: i for i, c in enumerate(graphemes)}
    short = sorted((t for t in tokens if len(t) <= 3), key=lambda t: (len(t), t))[:200]
    lexicon = [tuple(g2i[c] for c in t) for t in short]
    blank, ks, stride, Bc, Tc = len(graphemes), 7, 4, 8, 400
    layer = tr.ConvTransduce1D(lexicon, ks, stride, blank, scale="sqrt", normalize="none").cuda()
    xc = torch.log_softmax(torch.tensor(rs.randn(Bc, Tc, blank + 1).astype(np.float32)), 2).cuda().requires_grad_(True)
    Tout = (Tc + 2 * (ks // 2) - ks) // stride + 1
    w = torch.tensor(rs.randn(Bc, Tout, len(lexicon)).astype(np.float32)).cuda()

    def conv_fn():
        xc.grad = None
        (layer(xc) * w).sum().backward()

    ms = timed(conv_fn)
    alg = 4 * (2 * Bc * Tc * (blank + 1) + 2 * Bc * Tout * len(lexicon))
    print(json.dumps({"workload": f"ConvTransduce1D fwd+bwd, {len(lexicon)} word pieces, kernel {ks}, stride {stride}, scale sqrt, "
                                  f"B={Bc} T={Tc} C={blank + 1} (configs/iamdb/convtrans.json; transducer.py:370-556)",
                      "ms_per_step": ms, "value": Bc / ms * 1e3, "unit": "utt/s", "windows_x_entries": Bc * Tout * len(lexicon),
                      "algorithmic_bytes": alg, "achieved_GBps": alg / ms / 1e6}))


if __name__ == "__main__":
    main()
