import os, sys, time, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["WFL_PACK_TRACE"] = "1"
import torch, bench
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import transducer as TR
tokens, g2i = bench.word_pieces()
rnd = random.Random(0)
crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
crit.tokens.arc_sort(True)
for it in range(6):
    batch = [torch.tensor([g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)]) for _ in range(64)]
    flat, off, lens = E.flatten_any(batch)
    t0 = time.perf_counter()
    p = E.PackedLattice.transducer_batch(crit.tokens, crit.lexicon, None, flat, off, len(tokens) + 1, torch.device("cuda"))
    print("total %.0f us" % ((time.perf_counter() - t0) * 1e6))
    time.sleep(0.0005)
