"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/minigtn.py header).

Pins oracle/minigtn.py against the reference's own tests.  Runs ONLY where /root/reference exists
(the build container); nothing on the GPU box calls it.

What it does
  1. registers oracle/minigtn.py as `sys.modules["gtn"]` and puts /root/reference on sys.path, so
     the reference's criterion sources (criterions/{ctc,asg,stc,transducer}.py) import unmodified
     and run their own orchestration on the oracle's WFST primitives;
  2. provides the stale flat-layout spellings the reference's tests import (`utils.CTCLoss`,
     `utils.ASGLossFunction`, `utils.pack_replabels`, top-level `transducer`;
     tests/transducer_test.py:17-19, tests/utils_test.py:19) as aliases of the reference's own
     objects;
  3. loads /root/reference/tests/{gtn_ctc,gtn_asg,gtn_stc,transducer,utils}_test.py from where they
     lie and runs them with unittest -- every literal known-answer vector the reference holds for
     the path;
  4. with --write-golden: evaluates the reference criteria (again on the oracle primitives) on
     seeded random inputs and writes inputs + expected outputs as JSON fixtures under tests/golden/.
     Fixtures are data only (numbers); no reference source text is written anywhere.

Usage:  python oracle/pin_against_reference.py [--write-golden]
"""
import argparse
import importlib.util
import json
import os
import random
import sys
import types
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"


def install_reference_on_oracle():
    if not os.path.isdir(REF):
        raise SystemExit("pin_against_reference: /root/reference is not available here")
    sys.path.insert(0, HERE)
    import minigtn

    sys.modules["gtn"] = minigtn
    sys.path.insert(0, REF)
    from criterions import asg, ctc, stc, transducer  # the reference's own sources

    utils_shim = types.ModuleType("utils")
    utils_shim.CTCLoss = ctc.CTCLoss
    utils_shim.CTCLossFunction = ctc.CTCLossFunction
    utils_shim.ASGLoss = asg.ASGLoss
    utils_shim.ASGLossFunction = asg.ASGLossFunction
    utils_shim.pack_replabels = asg.pack_replabels
    utils_shim.unpack_replabels = asg.unpack_replabels
    sys.modules["utils"] = utils_shim
    sys.modules["transducer"] = transducer
    return minigtn, ctc, asg, stc, transducer


def run_reference_tests():
    names = ["gtn_ctc_test", "gtn_asg_test", "gtn_stc_test", "transducer_test", "utils_test"]
    suite = unittest.TestSuite()
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "tests"))  # transducer_test.py:535 loads trans_backoff_test.txt by relative path
    try:
        for name in names:
            spec = importlib.util.spec_from_file_location(
                "ref_" + name, os.path.join(REF, "tests", name + ".py")
            )
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            suite.addTests(unittest.defaultTestLoader.loadTestsFromModule(mod))
        result = unittest.TextTestRunner(verbosity=2).run(suite)
    finally:
        os.chdir(cwd)
    return result


# --------------------------------------------------------------------------------------------------
# golden vectors
# --------------------------------------------------------------------------------------------------
def _tolist(t):
    return t.detach().cpu().double().numpy().tolist()


def write_golden(ctc, asg, stc, transducer):
    import torch

    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_default_dtype(torch.float32)
    cases = {}

    # ---- CTC (criterions/ctc.py) ----------------------------------------------------------------
    for name, (B, T, C, Ls, reduction, normalise) in {
        "ctc_rand_none": (3, 12, 7, [4, 0, 6], "none", True),
        "ctc_rand_mean": (4, 20, 15, [11, 2, 3, 5], "mean", True),
        "ctc_raw_scores": (2, 16, 9, [5, 7], "none", False),  # benchmark style: raw randn as "log_probs"
    }.items():
        g = torch.Generator().manual_seed(sum(map(ord, name)))
        x = torch.randn(B, T, C, generator=g)
        blank = C - 1
        targets = [torch.randint(0, C - 1, (L,), generator=g).tolist() for L in Ls]
        if name == "ctc_rand_mean":
            targets[1] = [1, 1]  # forced blank between repeats
            targets[3] = [0, 0, 0, 0, 0]
        x.requires_grad_(True)
        lp = torch.nn.functional.log_softmax(x, 2) if normalise else x
        loss = ctc.CTCLoss(lp, targets, blank, reduction)
        loss.backward()
        cases[name] = dict(
            kind="ctc", inputs=_tolist(x), log_softmax=normalise, targets=targets, blank=blank,
            reduction=reduction, loss=loss.item(), grad=_tolist(x.grad),
        )

    # ---- ASG (criterions/asg.py) ----------------------------------------------------------------
    for name, (B, T, C, Ls, reduction) in {
        "asg_rand_none": (3, 10, 6, [4, 2, 7], "none"),
        "asg_rand_mean": (2, 14, 8, [5, 9], "mean"),
    }.items():
        g = torch.Generator().manual_seed(len(name) * 31 + 5)
        x = torch.randn(B, T, C, generator=g, requires_grad=True)
        W = (0.5 * torch.randn(C + 1, C, generator=g)).requires_grad_(True)
        targets = [torch.randint(0, C, (L,), generator=g).tolist() for L in Ls]
        loss = asg.ASGLoss(x, W, targets, reduction)
        loss.backward()
        cases[name] = dict(
            kind="asg", inputs=_tolist(x), transitions=_tolist(W), targets=targets,
            reduction=reduction, loss=loss.item(), grad=_tolist(x.grad), trans_grad=_tolist(W.grad),
        )
    # ASG module: replabels + garbage + viterbi (asg.py:191-237)
    g = torch.Generator().manual_seed(99)
    crit = asg.ASG(5, num_replabels=2, use_garbage=True)
    with torch.no_grad():
        crit.transitions.copy_(0.3 * torch.randn(crit.N + 1, crit.N, generator=g))
    x = torch.randn(2, 12, crit.N, generator=g, requires_grad=True)
    tg = [torch.tensor([1, 1, 1, 3]), torch.tensor([0, 2, 2])]
    loss = crit(x, tg)
    loss.backward()
    vit = [p.tolist() for p in crit.viterbi(x.detach())]
    cases["asg_module"] = dict(
        kind="asg_module", num_classes=5, num_replabels=2, use_garbage=True,
        transitions=_tolist(crit.transitions), inputs=_tolist(x), targets=[t.tolist() for t in tg],
        loss=loss.item(), grad=_tolist(x.grad), trans_grad=_tolist(crit.transitions.grad), viterbi=vit,
    )

    # ---- STC (criterions/stc.py) ----------------------------------------------------------------
    for name, (B, T, C, Ls, p0, reduction) in {
        "stc_rand_none": (2, 8, 6, [3, 2], 0.7, "none"),
        "stc_rand_mean": (3, 10, 5, [2, 4, 1], 0.4, "mean"),
    }.items():
        g = torch.Generator().manual_seed(len(name) * 7 + 3)
        x = torch.randn(T, B, C, generator=g, requires_grad=True)
        targets = [torch.randint(1, C, (L,), generator=g).tolist() for L in Ls]
        crit = stc.STC(0, p0, p0, 1, reduction)
        crit.eval()
        lp = torch.nn.functional.log_softmax(x, 2)
        loss = crit(lp, targets)
        loss.backward()
        cases[name] = dict(
            kind="stc", inputs=_tolist(x), targets=targets, p0=p0, plast=p0, thalf=1,
            reduction=reduction, loss=loss.item(), grad=_tolist(x.grad),
        )

    # ---- Transducer (criterions/transducer.py) ----------------------------------------------------
    def run_transducer(name, tokens, g2i, targets, T, seed, scale=1.0, trans_scale=0.0, **kw):
        g = torch.Generator().manual_seed(seed)
        crit = transducer.Transducer(tokens=tokens, graphemes_to_idx=g2i, **kw)
        C = len(tokens) + int(kw.get("blank", "none") != "none")
        x = (scale * torch.randn(len(targets), T, C, generator=g)).requires_grad_(True)
        if crit.transition_params is not None and trans_scale:
            with torch.no_grad():
                crit.transition_params.copy_(
                    trans_scale * torch.randn(crit.transition_params.numel(), generator=g)
                )
        loss = crit(x, targets)
        loss.backward()
        vit = [p.tolist() for p in crit.viterbi(x.detach())]
        rec = dict(
            kind="transducer", tokens=[list(t) if not isinstance(t, str) else t for t in tokens],
            graphemes_to_idx={str(k): v for k, v in g2i.items()}, grapheme_keys_are_int=not isinstance(tokens[0], str),
            targets=[list(map(int, t)) for t in targets], kwargs=kw, inputs=_tolist(x),
            loss=loss.item(), grad=_tolist(x.grad), viterbi=vit,
        )
        if crit.transition_params is not None:
            rec["transition_params"] = _tolist(crit.transition_params)
            rec["transition_grad"] = _tolist(crit.transition_params.grad)
        cases[name] = rec

    wp = ["a", "b", "ab", "ba", "aba"]
    g2i = {"a": 0, "b": 1}
    run_transducer("tr_decomp_none", wp, g2i, [[0, 1, 0], [1, 0]], 6, 11)
    run_transducer("tr_decomp_blank_opt", wp, g2i, [[0, 1, 0], [0, 0, 1]], 7, 12,
                   blank="optional", reduction="mean")
    run_transducer("tr_decomp_norepeat", wp, g2i, [[0, 1, 0, 1], [1, 1]], 8, 13,
                   blank="optional", allow_repeats=False, reduction="mean")
    run_transducer("tr_decomp_forced", wp, g2i, [[0, 1, 0], [1]], 8, 14, blank="forced")
    toks = [(i,) for i in range(4)]
    gi = {i: i for i in range(4)}
    run_transducer("tr_ngram1", toks, gi, [[0, 1, 2], [3, 3]], 6, 21, trans_scale=0.5, ngram=1)
    run_transducer("tr_ngram2_blank", toks, gi, [[0, 1, 2], [3, 3]], 7, 22, trans_scale=0.5,
                   ngram=2, blank="optional", allow_repeats=False, reduction="mean")
    run_transducer("tr_ngram2_noblank", toks, gi, [[2, 1], [0, 3, 3]], 6, 23, trans_scale=0.5, ngram=2)

    # ---- ConvTransduce1D (criterions/transducer.py:370-556), the reference module itself ------------
    def run_conv(name, lexicon, ks, stride, blank_idx, B, T, C, seed, param_scale=0.0, **kw):
        g = torch.Generator().manual_seed(seed)
        layer = transducer.ConvTransduce1D(lexicon, ks, stride, blank_idx, **kw)
        if layer.kernel_params is not None and param_scale:
            with torch.no_grad():
                layer.kernel_params.copy_(param_scale * torch.randn(layer.kernel_params.numel(), generator=g))
        x = torch.randn(B, T, C, generator=g).requires_grad_(True)
        out = layer(x)
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        rec = dict(kind="conv", lexicon=[list(l) for l in lexicon], kernel_size=ks, stride=stride,
                   blank_idx=blank_idx, kwargs=kw, inputs=_tolist(x), out_weights=_tolist(w),
                   outputs=_tolist(out), grad=_tolist(x.grad))
        if layer.kernel_params is not None:
            rec["kernel_params"] = _tolist(layer.kernel_params)
            rec["kernel_grad"] = _tolist(layer.kernel_params.grad)
        cases[name] = rec

    lex4 = [(0, 0), (0, 1), (1, 0), (1, 1)]
    run_conv("conv_basic", lex4, 5, 3, 2, 2, 8, 3, 31)
    run_conv("conv_no_optional_blank", lex4, 5, 2, 2, 2, 7, 3, 32, blank_optional=False)
    run_conv("conv_spike_sqrt", [(0,), (1, 2), (2, 1, 0)], 5, 1, 3, 1, 6, 4, 33, spike=True, scale="sqrt")
    run_conv("conv_learned_post", [(0, 1, 2), (2,), (1, 1)], 7, 4, 3, 2, 9, 4, 34, param_scale=0.5,
             learn_params=True, normalize="post", scale="linear")
    run_conv("conv_viterbi_pre", lex4, 5, 3, 2, 2, 8, 3, 35, viterbi=True, normalize="pre")
    run_conv("conv_viterbi_learned", [(0, 1), (1,), (0, 0)], 5, 2, 2, 1, 7, 3, 36, param_scale=0.7,
             viterbi=True, learn_params=True)

    # ---- structural goldens of the graph builders ----------------------------------------------
    def dump(gr):
        return dict(
            num_nodes=gr.num_nodes(), start=gr.start_nodes(), accept=gr.accept_nodes(),
            arcs=[[gr.src[a], gr.dst[a], gr.ilab[a], gr.olab[a], gr.w[a]] for a in range(gr.num_arcs())],
        )

    import torch as _t

    builders = {
        "ctc_graph_0_1_1": dump(ctc.CTCLossFunction.create_ctc_graph([0, 1, 1], 2)),
        "ctc_graph_empty": dump(ctc.CTCLossFunction.create_ctc_graph([], 2)),
        "asg_fal_2_2_1": dump(asg.ASGLossFunction.create_force_align_graph([2, 2, 1])),
        "asg_transitions_c3": dump(
            asg.ASGLossFunction.create_transitions_graph(_t.arange(12, dtype=_t.float32).view(4, 3))
        ),
        "stc_graph_1_2": dump(stc.STCLossFunction.create_stc_graph([1, 2], 4, 0.5)),
        "token_none_rep": dump(transducer.make_token_graph(["a", "b", "c"], "none", True)),
        "token_opt_rep": dump(transducer.make_token_graph(["a", "b", "c"], "optional", True)),
        "token_opt_norep": dump(transducer.make_token_graph(["a", "b", "c"], "optional", False)),
        "token_forced_rep": dump(transducer.make_token_graph(["a", "b", "c"], "forced", True)),
        "lexicon_wp": dump(transducer.make_lexicon_graph(wp, g2i)),
        "chain_3_1_2": dump(transducer.make_chain_graph([3, 1, 2])),
        "ngram1_3": dump(transducer.make_transitions_graph(1, 3)),
        "ngram2_3": dump(transducer.make_transitions_graph(2, 3)),
        "ngram3_2": dump(transducer.make_transitions_graph(3, 2)),
        "kernel_0_0_opt": dump(transducer.make_kernel_graph([0, 0], 2, True)),
        "kernel_0_1_opt": dump(transducer.make_kernel_graph([0, 1], 2, True)),
        "kernel_0_1_noopt_spike": dump(transducer.make_kernel_graph([0, 1], 2, False, spike=True)),
    }

    with open(os.path.join(out_dir, "criterion_cases.json"), "w") as fid:
        json.dump(cases, fid)
    with open(os.path.join(out_dir, "builder_graphs.json"), "w") as fid:
        json.dump(builders, fid)
    print(f"wrote {len(cases)} criterion cases and {len(builders)} builder graphs to {out_dir}")


def write_round2_golden(transducer):
    """Round-2 fixtures, kept apart from criterion_cases.json so that the round-1 vectors stay byte-identical:
      * transition_builder.json -- the reference's scripts/build_transitions.py functions (count / prune /
        blank grams / self loops / build_graph) run on a small seeded corpus, on the oracle's Graph: corpus,
        options and the resulting node / arc lists (arc order included);
      * transducer_wordpieces_1000.npz -- the reference's Transducer module with the 1000 word pieces of
        benchmarks/word_pieces_tokens_1000.txt (BASELINE configs[3]: blank optional, no repeats, mean) on a short
        seeded input: loss and dense gradient (inputs are regenerated from the seed by the test)."""
    import numpy as np
    import torch

    out_dir = os.path.join(REPO, "tests", "golden")
    spec = importlib.util.spec_from_file_location("ref_build_transitions", os.path.join(REF, "scripts", "build_transitions.py"))
    bt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bt)

    def dump(gr):
        return dict(num_nodes=gr.num_nodes(), start=gr.start_nodes(), accept=gr.accept_nodes(),
                    arcs=[[gr.src[a], gr.dst[a], gr.ilab[a], gr.olab[a]] for a in range(gr.num_arcs())])

    rnd = random.Random(5)
    tokens = list("abcde")
    # a corpus with a skewed distribution, so that pruning thresholds bite differently per order
    lines = ["".join(rnd.choice("aaabbcde"[: rnd.randint(3, 8)]) for _ in range(rnd.randint(1, 9))) for _ in range(60)]
    t2i = {t: e for e, t in enumerate(tokens)}
    cases = {}
    for name, (prune, blank, loops, nobackoff) in {
        "unigram": ([0], "none", False, False),
        "bigram": ([0, 1], "none", False, False),
        "trigram_pruned": ([0, 2, 4], "none", False, False),
        "bigram_blank_optional": ([0, 1], "optional", False, False),
        "bigram_blank_forced": ([0, 1], "forced", False, False),
        "trigram_self_loops": ([0, 1, 3], "none", True, False),
        "bigram_no_backoff": ([0, 0], "none", False, True),
        "trigram_blank_optional_loops": ([0, 1, 2], "optional", True, False),
    }.items():
        counts = bt.count_ngrams(lines, len(prune), t2i)
        kept = bt.prune_ngrams(counts, prune)
        if blank != "none":
            kept = bt.add_blank_grams(kept, len(t2i), blank)
        if loops:
            kept = bt.add_self_loops(kept)
        gr = bt.build_graph(kept, nobackoff)
        cases[name] = dict(prune=prune, blank=blank, add_self_loops=loops, disable_backoff=nobackoff, graph=dump(gr),
                           kept=[[list(g) for g in grams] for grams in kept])
    with open(os.path.join(out_dir, "transition_builder.json"), "w") as fid:
        json.dump(dict(tokens=tokens, lines=lines, cases=cases), fid)
    print(f"wrote {len(cases)} transition-builder graphs")

    with open(os.path.join(out_dir, "word_pieces_tokens_1000.txt"), "r") as fid:
        wp = sorted(l.strip() for l in fid)
    graphemes = sorted(set(c for t in wp for c in t))
    g2i = {t: i for i, t in enumerate(graphemes)}
    rnd = random.Random(11)
    B, T, L, seed = 2, 24, 3, 2024
    targets = [[g2i[c] for _ in range(L) for c in rnd.choice(wp)] for _ in range(B)]
    crit = transducer.Transducer(wp, g2i, blank="optional", allow_repeats=False, reduction="mean")
    x = torch.randn(B, T, len(wp) + 1, generator=torch.Generator().manual_seed(seed)).requires_grad_(True)
    loss = crit(x, [torch.tensor(t) for t in targets])
    loss.backward()
    np.savez_compressed(os.path.join(out_dir, "transducer_wordpieces_1000.npz"), seed=seed, B=B, T=T,
                        targets=np.array([len(t) for t in targets] + [v for t in targets for v in t]),
                        loss=loss.item(), grad=x.grad.numpy().astype(np.float32),
                        x_checksum=float(x.detach().double().sum()))
    print("wrote transducer_wordpieces_1000.npz: loss", loss.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-golden", action="store_true")
    ap.add_argument("--write-round2-golden", action="store_true")
    args = ap.parse_args()
    random.seed(0)
    _, ctc, asg, stc, transducer = install_reference_on_oracle()
    result = run_reference_tests()
    ok = result.wasSuccessful()
    print("REFERENCE TESTS ON ORACLE:", "PASS" if ok else "FAIL",
          f"(run={result.testsRun}, skipped={len(result.skipped)})")
    if args.write_golden:
        if not ok:
            raise SystemExit("refusing to write golden vectors from an unpinned oracle")
        write_golden(ctc, asg, stc, transducer)
    if args.write_round2_golden:
        if not ok:
            raise SystemExit("refusing to write golden vectors from an unpinned oracle")
        write_round2_golden(transducer)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
