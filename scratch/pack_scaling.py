"""Transducer batch packer (wfl_transducer_pack_batch): time per batch of 64 against the number of pool threads, and its phases."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from gtn_applications_amd import _native as N, engine as E
from gtn_applications_amd.criterions import transducer as TR
tokens, g2i = bench.word_pieces()
crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
crit.tokens.arc_sort(True)
rnd = random.Random(0)
NB = 40
tb = [[torch.tensor([g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)]) for _ in range(64)] for _ in range(NB)]
flats = [E.flatten_any(t) for t in tb]
Cc = len(tokens) + 1
def run(i, nt):
    flat, off, _ = flats[i]
    h = N.lib.wfl_transducer_pack_batch(crit.tokens._h, crit.lexicon._h, None, flat.ctypes.data, off.ctypes.data, 64, Cc, nt)
    N.lib.wfl_lattice_host_free(h)
for nt in (1, 2, 4, 8, 16, 31, 0):
    for i in range(5): run(i, nt)
    t0 = time.perf_counter()
    for i in range(5, NB): run(i, nt)
    print(f"threads {nt:2d}: {(time.perf_counter() - t0) / (NB - 5) * 1e6:8.1f} us per batch of 64")
    # with a pause between calls (a training step's worth): are the workers asleep by then?
    t = 0.0
    for i in range(5, 25):
        time.sleep(0.0004)
        t0 = time.perf_counter(); run(i, nt); t += time.perf_counter() - t0
    print(f"            {t / 20 * 1e6:8.1f} us with 400 us between calls")
