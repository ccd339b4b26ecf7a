import sys, time, random, torch
sys.path.insert(0, "/root/repo")
from gtn_applications_amd.criterions import transducer as TR
rnd = random.Random(0)
ntok, K, ks, stride, B, T = 200, 1000, 7, 4, 8, 256
lexicon = [tuple(rnd.randrange(ntok) for _ in range(rnd.choice([1, 2, 3, 4, 5]))) for _ in range(K)]
conv = TR.ConvTransduce1D(lexicon, ks, stride, ntok, scale="sqrt")
x = torch.randn(B, T, ntok + 1, device="cuda", requires_grad=True)
def step():
    x.grad = None
    out = conv(x)
    out.backward(torch.ones_like(out))
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
n = B * ((T + 2 * (ks // 2) - ks) // stride + 1) * K
print(f"conv fwd+bwd: {dt*1e3:.3f} ms, {n} window x entry DPs, {n/dt/1e9:.2f} G DP/s")
