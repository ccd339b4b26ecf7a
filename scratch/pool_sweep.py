"""wfl_transducer_pack_batch wall time vs number of host threads (cfg4 batch, fresh targets)."""
import os, sys, subprocess, time
if len(sys.argv) > 1:
    os.environ["WFL_HOST_THREADS"] = sys.argv[1]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import random, numpy as np
    from gtn_applications_amd import _native as N, engine as E
    from gtn_applications_amd.criterions import transducer as TR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tokens = sorted(l.strip() for l in open(os.path.join(root, "benchmarks", "word_pieces_tokens_1000.txt")))
    g2i = {t: i for i, t in enumerate(sorted(set(c for t in tokens for c in t)))}
    crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
    crit.tokens.arc_sort(True)
    rnd = random.Random(0)
    batches = [E.flatten_targets([[g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)] for _ in range(64)]) for _ in range(60)]
    def run(i):
        flat, off, _ = batches[i]
        h = N.lib.wfl_transducer_pack_batch(crit.tokens._h, crit.lexicon._h, None, flat.ctypes.data, off.ctypes.data, 64, 1001, 0)
        N.lib.wfl_lattice_host_free(h)
    for i in range(10): run(i)
    t0 = time.perf_counter()
    for i in range(10, 60): run(i)
    print(f"threads {sys.argv[1]:>4s}: {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us per batch of 64")
else:
    for n in ("1", "4", "8", "16", "24", "32", "48", "64", "128"):
        subprocess.run([sys.executable, __file__, n])
