"""The general (epsilon-aware, log-domain) lattice sweep on the bigram Transducer's numerator as the reference builds
it -- alignments o make_transitions_graph(2, N), two epsilon arcs per acceptor (N = 81, T = 250, L = 44): time of
wfl_lattice_forward (events around 20 calls)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import transducer as TR

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, T, L = 81, 250, 44
torch.manual_seed(0)
crit = TR.Transducer([(i,) for i in range(N)], {i: i for i in range(N)}, ngram=2, reduction="mean", blank="optional",
                     allow_repeats=False).cuda()
C = N + 1
x = torch.randn(B, T, C).cuda()
targets = [t.squeeze() for t in torch.randint(N, size=(B, L)).split(1)]
params = crit.transition_params.detach()
crit.tokens.arc_sort(True)
nb, entry = TR._pack_entry(targets, crit.tokens, crit.lexicon, crit.transitions, C, x.device, "mean")
pack = entry[0]
for _ in range(3):
    st = E.lattice_forward(x, pack, weights=params, need_beta=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    st = E.lattice_forward(x, pack, weights=params, need_beta=True)
e1.record()
torch.cuda.synchronize()
import ctypes
from gtn_applications_amd import _native as NL
off = ctypes.c_int64()
NL.check(NL.lib.wfl_lattice_formats_offset(ctypes.byref(pack.desc), T, ctypes.byref(off)))
fm = st.alpha[off.value:off.value + B].view(torch.int32).cpu().tolist()
print("formats (0 log domain, 1 probability domain):", sorted(set(fm)), "probability-domain utterances:", fm.count(1), "of", B)
print(f"B={B}: states {pack.desc.max_states} arcs {pack.desc.max_arcs} eps {pack.desc.max_eps}: lattice_forward {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
