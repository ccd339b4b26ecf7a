"""CPU tests (-m "not gpu") of the native host side: libwfl.so loads and exports every symbol that
include/wfl.h declares, the graph builders reproduce the reference's builders arc by arc, and the
C++ graph algebra (compose / remove / project / viterbi_path / pack) agrees with the oracle.
No device compute is called here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from gtn_applications_amd import _native as N
from gtn_applications_amd import engine as E
from gtn_applications_amd import graph as G
from gtn_applications_amd.criterions import asg, ctc, stc, transducer as TR
from oracle import criteria as OC
from oracle import minigtn as MG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "wfl.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = set(re.findall(r"\b(wfl_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    assert declared == set(N.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(N.lib, name), name
    assert N.lib.wfl_version() >= 1


def test_desc_struct_matches_header_layout():
    # one known field pattern: pack a tiny batch and read sizes back through the ctypes mirror
    flat = np.array([1, 2, 2], np.int32)
    off = np.array([0, 2, 3], np.int64)
    p = E.PackedLattice.ctc(flat, off, 0, 4, None)
    d = p.desc
    assert (d.B, d.max_states, d.total_states, d.shared) == (2, 5, 8, 0)
    assert d.total_arcs == (5 + 4 + 1) + (3 + 2 + 0)
    assert d.float_words >= d.total_arcs + 2 * d.total_states
    assert list(p.field("state_off", 3)) == [0, 5, 8]
    assert list(p.field("labels", d.total_labels)) == [0, 1, 2, 0, 2]


def as_oracle(g):
    a = g.arrays()
    o = MG.Graph(False)
    for s, c in zip(a["start"], a["accept"]):
        o.add_node(bool(s), bool(c))
    for k in range(len(a["src"])):
        o.add_arc(int(a["src"][k]), int(a["dst"][k]), int(a["ilabel"][k]), int(a["olabel"][k]), float(a["weight"][k]))
    return o


def dump(g):
    a = g.arrays()
    return dict(
        num_nodes=len(a["start"]), start=np.nonzero(a["start"])[0].tolist(), accept=np.nonzero(a["accept"])[0].tolist(),
        arcs=[[int(a["src"][k]), int(a["dst"][k]), int(a["ilabel"][k]), int(a["olabel"][k]), float(a["weight"][k])]
              for k in range(len(a["src"]))],
    )


def test_builders_match_reference(golden_dir):
    import torch

    with open(os.path.join(golden_dir, "builder_graphs.json")) as f:
        want = json.load(f)
    wp, g2i = ["a", "b", "ab", "ba", "aba"], {"a": 0, "b": 1}
    mine = {
        "ctc_graph_0_1_1": ctc.CTCLossFunction.create_ctc_graph([0, 1, 1], 2),
        "ctc_graph_empty": ctc.CTCLossFunction.create_ctc_graph([], 2),
        "asg_fal_2_2_1": asg.ASGLossFunction.create_force_align_graph([2, 2, 1]),
        "asg_transitions_c3": asg.ASGLossFunction.create_transitions_graph(torch.arange(12.0).view(4, 3)),
        "stc_graph_1_2": stc.STCLossFunction.create_stc_graph([1, 2], 4, 0.5),
        "token_none_rep": TR.make_token_graph(["a", "b", "c"], "none", True),
        "token_opt_rep": TR.make_token_graph(["a", "b", "c"], "optional", True),
        "token_opt_norep": TR.make_token_graph(["a", "b", "c"], "optional", False),
        "token_forced_rep": TR.make_token_graph(["a", "b", "c"], "forced", True),
        "lexicon_wp": TR.make_lexicon_graph(wp, g2i),
        "chain_3_1_2": TR.make_chain_graph([3, 1, 2]),
        "ngram1_3": TR.make_transitions_graph(1, 3),
        "ngram2_3": TR.make_transitions_graph(2, 3),
        "ngram3_2": TR.make_transitions_graph(3, 2),
        "kernel_0_0_opt": TR.make_kernel_graph([0, 0], 2, True),
        "kernel_0_1_opt": TR.make_kernel_graph([0, 1], 2, True),
        "kernel_0_1_noopt_spike": TR.make_kernel_graph([0, 1], 2, False, spike=True),
    }
    assert set(mine) == set(want)
    for name, g in mine.items():
        got = dump(g)
        assert got["num_nodes"] == want[name]["num_nodes"], name
        assert got["start"] == want[name]["start"] and got["accept"] == want[name]["accept"], name
        assert [a[:4] for a in got["arcs"]] == [a[:4] for a in want[name]["arcs"]], name
        np.testing.assert_allclose([a[4] for a in got["arcs"]], [a[4] for a in want[name]["arcs"]], atol=1e-6)


def test_token_graph_errors():
    with pytest.raises(ValueError):
        TR.make_token_graph(["a"], "none", False)
    with pytest.raises(ValueError):
        TR.Transducer(["a"], {"a": 0}, blank="sometimes")
    with pytest.raises(ValueError):
        TR.Transducer(["a"], {"a": 0}, ngram=1, transitions=G.Graph())


def score(g, x):
    """oracle forward score of emissions o g for a host graph of either library"""
    og = g if isinstance(g, MG.Graph) else as_oracle(g)
    return MG.forward_score(MG.intersect(OC.emissions_graph(x, False), MG.project_input(og))).item()


@pytest.mark.parametrize("blank,repeats", [("none", True), ("optional", True), ("optional", False), ("forced", True)])
def test_alignment_graphs_equivalent_to_oracle(blank, repeats):
    wp, g2i = ["a", "b", "ab", "ba", "aba"], {"a": 0, "b": 1}
    tokens = TR.make_token_graph(wp, blank, repeats)
    lexicon = TR.make_lexicon_graph(wp, g2i)
    oracle = OC.TransducerOracle(wp, g2i, blank=blank, allow_repeats=repeats)
    rs = np.random.RandomState(1)
    C = len(wp) + int(blank != "none")
    for target in ([0, 1, 0], [1, 1, 0, 0], [0], [1, 0, 1, 0, 1]):
        mine, _ = TR._alignment_graph(target, tokens, lexicon, None)
        ref = oracle.alignment_graph(target)
        assert mine.num_arcs() == ref.num_arcs() and mine.num_nodes() == ref.num_nodes()
        assert MG.isomorphic(as_oracle(mine), MG.project_input(ref))
        x = rs.randn(7, C)
        assert score(mine, x) == pytest.approx(score(ref, x), rel=1e-6, abs=1e-6)


def test_compose_with_transitions_and_provenance():
    toks = [(i,) for i in range(4)]
    g2i = {i: i for i in range(4)}
    tokens = TR.make_token_graph(toks, "optional", False)
    lexicon = TR.make_lexicon_graph(toks, g2i)
    trans = TR.make_transitions_graph(2, 5)
    trans.arc_sort()
    ali, wid = TR._alignment_graph([0, 1, 2], tokens, lexicon, trans)
    a = ali.arrays()
    t = trans.arrays()
    assert wid.shape[0] == ali.num_arcs() and wid.min() >= 0
    # each composed arc carries the input label of the transition arc it came from
    np.testing.assert_array_equal(a["ilabel"], t["ilabel"][wid])
    oracle = OC.TransducerOracle(toks, g2i, ngram=2, blank="optional", allow_repeats=False)
    ref = MG.intersect(oracle.transitions, oracle.alignment_graph([0, 1, 2]))
    assert ali.num_arcs() == ref.num_arcs() and ali.num_nodes() == ref.num_nodes()


def test_remove_project_viterbi_equal_isomorphic(tmp_path):
    g = G.Graph()
    for k in range(4):
        g.add_node(k == 0, k == 3)
    g.add_arc(0, 1, 1, G.epsilon, 0.5)
    g.add_arc(1, 2, G.epsilon, G.epsilon, 0.0)
    g.add_arc(2, 3, 2, 7, 1.5)
    g.add_arc(0, 3, 3, 3, 1.0)
    r = G.remove(G.project_output(g))
    assert dump(r)["arcs"] == dump_oracle(MG.remove(MG.project_output(as_oracle(g))))["arcs"]
    best = G.viterbi_path(g)
    assert best.labels_to_list() == [1, G.epsilon, 2] and best.labels_to_list(False) == [G.epsilon, G.epsilon, 7]
    assert G.equal(g, g) and G.isomorphic(g, g)
    h = G.Graph()
    for k in (3, 2, 1, 0):  # same graph, nodes renumbered n -> 3-n
        h.add_node(k == 0, k == 3)
    h.add_arc(3, 2, 1, G.epsilon, 0.5), h.add_arc(2, 1, G.epsilon, G.epsilon, 0.0)
    h.add_arc(1, 0, 2, 7, 1.5), h.add_arc(3, 0, 3, 3, 1.0)
    assert G.isomorphic(g, h) and not G.equal(g, h)
    path = tmp_path / "g.txt"
    G.savetxt(path, g)
    assert G.equal(G.loadtxt(path), g)


def dump_oracle(g):
    return dict(arcs=[[g.src[a], g.dst[a], g.ilab[a], g.olab[a], g.w[a]] for a in range(g.num_arcs())])


def test_viterbi_path_prefers_shortest_decoding():
    # tests/transducer_test.py:318-353: [1,1,3,3,0] with repeats allowed decodes to [1,3,0]
    tokens = TR.make_token_graph(["a", "b", "c", "d"], "none", True)
    path = G.compose(TR.make_chain_graph([1, 1, 3, 3, 0]), tokens)
    best = G.remove(G.project_output(G.viterbi_path(path)))
    assert best.labels_to_list() == [1, 3, 0]


def test_backoff_text_fixture_loads(golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "reference_literals.json")) as f:
        c = json.load(f)["backoff_transitions"]
    lines = [" ".join(map(str, c["start"])), " ".join(map(str, c["accept"]))]
    lines += [" ".join(map(str, a)) for a in c["arcs"]]
    p = tmp_path / "backoff.txt"
    p.write_text("\n".join(lines) + "\n")
    g = G.loadtxt(p)
    assert (g.num_nodes(), g.num_arcs()) == (8, 37)
    a = g.arrays()
    assert a["ilabel"][0] == G.epsilon and a["start"].tolist() == [0] * 7 + [1]


def test_pack_epsilon_levels_and_sorting():
    trans = TR.make_transitions_graph(2, 3)  # 5 nodes: start, 3 histories, </s>; epsilon arcs into </s>
    wid = np.arange(trans.num_arcs(), dtype=np.int32)
    p = E.PackedLattice.from_graphs([trans], 3, None, wids=[wid], B=4, shared=True)
    d = p.desc
    assert (d.B, d.shared, d.max_levels, d.total_eps, d.total_arcs) == (4, 1, 2, 4, 12)
    lv = p.field("lvl_ptr", 3)
    assert list(lv) == [0, 4, 5]  # the </s> node is the only level-1 state
    dst = p.field("arc_dst", 12)
    assert (np.diff(dst) >= 0).all()
    np.testing.assert_array_equal(np.sort(p.field("arc_wid", 12)), np.arange(12))
    np.testing.assert_array_equal(np.sort(p.field("eps_wid", 4)), np.arange(12, 16))
    g = G.Graph()
    g.add_node(True), g.add_node(False, True)
    g.add_arc(0, 1, G.epsilon), g.add_arc(1, 0, G.epsilon)
    with pytest.raises(N.WflError):
        E.PackedLattice.from_graphs([g], 3, None)
    g2 = G.Graph()
    g2.add_node(True, True)
    g2.add_arc(0, 0, 5)
    with pytest.raises(N.WflError):  # label outside [0, C)
        E.PackedLattice.from_graphs([g2], 3, None)


def test_bulk_packers_match_generic_packer():
    targets = [[1, 2, 2, 0], [], [3]]
    flat, off, _ = E.flatten_targets(targets)
    fast = E.PackedLattice.ctc(flat, off, 4, 5, None)
    slow = E.PackedLattice.from_graphs([ctc.CTCLossFunction.create_ctc_graph(t, 4) for t in targets], 5, None)
    for name in ("state_off", "arc_off", "in_ptr", "out_ptr", "out_arc", "arc_src", "arc_dst", "arc_slot", "arc_lab",
                 "labels", "arc_orig"):
        n = {"state_off": 4, "arc_off": 4, "in_ptr": fast.desc.total_states + 3, "out_ptr": fast.desc.total_states + 3,
             "labels": fast.desc.total_labels}.get(name, fast.desc.total_arcs)
        np.testing.assert_array_equal(fast.field(name, n), slow.field(name, n), err_msg=name)
    np.testing.assert_array_equal(fast.host_floats, slow.host_floats)
    fast = E.PackedLattice.stc(flat, off, 5, float(np.log(0.5)), 10, None)
    slow = E.PackedLattice.from_graphs([stc.STCLossFunction.create_stc_graph(t, 5, 0.5) for t in targets], 10, None)
    np.testing.assert_array_equal(fast.host_ints, slow.host_ints)
    np.testing.assert_allclose(fast.host_floats, slow.host_floats, rtol=1e-6)
    # ASG force alignment: the closed-form packer against the generic one on the same arcs and weight indices
    rs = np.random.RandomState(3)
    C = 7
    targets = [[1, 2, 2, 0], [], [3], [6] * 9] + [rs.randint(0, C, size=rs.randint(1, 30)).tolist() for _ in range(20)]
    flat, off, _ = E.flatten_targets(targets)
    graphs, wids = [], []
    for t in targets:
        g = G.Graph(False)
        g.add_node(True, False)
        wid = []
        for l in range(1, len(t) + 1):
            g.add_node(False, l == len(t))
            c = t[l - 1]
            g.add_arc(l - 1, l, c)
            g.add_arc(l, l, c)
            wid += [c if l == 1 else (1 + c) * C + t[l - 2], (1 + c) * C + c]
        graphs.append(g)
        wids.append(np.asarray(wid, dtype=np.int32))
    fast = E.PackedLattice.asg_force_align(flat, off, C, None)
    slow = E.PackedLattice.from_graphs(graphs, C, None, wids=wids)
    np.testing.assert_array_equal(fast.host_ints, slow.host_ints)
    np.testing.assert_array_equal(fast.host_floats, slow.host_floats)
    for f in ("max_states", "max_arcs", "max_labels", "max_levels", "total_states", "total_arcs", "total_labels"):
        assert getattr(fast.desc, f) == getattr(slow.desc, f), f


def test_replabels_match_reference_vectors(golden_dir):
    with open(os.path.join(golden_dir, "reference_literals.json")) as f:
        c = json.load(f)["replabels"]
    for n, want in c["pack"].items():
        assert asg.pack_replabels(c["pack_in"], int(n)) == want
    for n, want in c["unpack"].items():
        assert asg.unpack_replabels(c["unpack_in"], int(n)) == want


def test_compat_aliases():
    import sys

    from gtn_applications_amd import compat

    saved = {k: sys.modules.get(k) for k in ("utils", "transducer", "criterions", "criterions.ctc")}
    try:
        sys.modules.pop("utils", None)
        compat.install()
        from utils import ASGLossFunction, CTCLoss, pack_replabels  # noqa: F401
        import transducer  # noqa: F401
        from criterions import ctc as c2

        assert c2.CTCLoss is CTCLoss and transducer.Transducer is TR.Transducer
        assert pack_replabels([0, 0, 1], 1) == [1, 0, 2]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_conv_kernel_table_matches_kernel_graphs():
    """the device table of ConvTransduce1D numbers arcs exactly like make_kernel_graph inserts them
    (transducer.py:351-364), for every blank_optional / spike combination"""
    from gtn_applications_amd.criterions import transducer as TR

    lexicon = [(0, 0), (0, 1), (1,), (2, 1, 1, 0), ()]
    for bo in (True, False):
        for spike in (True, False):
            tab = TR._KernelTable(lexicon, bo, spike)
            n = 0
            for k, tok in enumerate(lexicon):
                g = TR.make_kernel_graph(tok, 3, bo, spike)
                a = g.arrays()
                src, dst, lab = a["src"].tolist(), a["dst"].tolist(), a["ilabel"].tolist()
                assert tab.table[k, 0] == len(tok) and tab.table[k, 34] == n
                assert (src[0], dst[0], lab[0]) == (0, 0, 3)
                ns = 0 if spike else 1
                for i, c in enumerate(tok):
                    base = tab.table[k, 18 + i] - n
                    assert (src[base], dst[base], lab[base]) == (2 * i, 2 * i + 1, c)
                    if ns:
                        assert (src[base + 1], dst[base + 1], lab[base + 1]) == (2 * i + 1, 2 * i + 1, c)
                    assert (src[base + 1 + ns], dst[base + 1 + ns], lab[base + 1 + ns]) == (2 * i + 1, 2 * i + 2, 3)
                    assert (src[base + 2 + ns], dst[base + 2 + ns], lab[base + 2 + ns]) == (2 * i + 2, 2 * i + 2, 3)
                    if (tab.table[k, 1] >> i) & 1:
                        assert (src[base + 3 + ns], dst[base + 3 + ns], lab[base + 3 + ns]) == (2 * i - 1, 2 * i + 1, c)
                n += g.num_arcs()
            assert tab.num_arcs == n


def test_load_criterion_factory(golden_dir):
    """utils.load_criterion (utils.py:245-273): criterion types, output sizes, unknown type"""
    import types

    import gtn_applications_amd as pkg

    pre = types.SimpleNamespace(num_tokens=5, tokens=["a", "b", "ab", "ba", "aba"], graphemes_to_index={"a": 0, "b": 1})
    crit, n = pkg.load_criterion("ctc", pre, {})
    assert n == 6 and crit.blank == 5
    crit, n = pkg.load_criterion("asg", pre, {"num_replabels": 1, "use_garbage": True})
    assert n == 7 and tuple(crit.transitions.shape) == (8, 7)
    crit, n = pkg.load_criterion("transducer", pre, {"blank": "optional", "allow_repeats": False, "ngram": 2})
    assert n == 6 and crit.transition_params is not None and crit.reduction == "mean"
    crit, n = pkg.load_criterion("transducer", pre, {})
    assert n == 5 and crit.transition_params is None
    with pytest.raises(ValueError):
        pkg.load_criterion("seq2seq", pre, {})
    from gtn_applications_amd import compat

    assert compat.install().load_criterion is pkg.load_criterion


def test_ctc_workspace_fields_lie_inside_the_workspace():
    """wfl_ctc_workspace_field (diagnostics): every field is inside the workspace wfl_ctc_workspace sizes, and
    the fields do not overlap"""
    import ctypes

    from gtn_applications_amd import _native as N

    for (B, T, L) in [(1, 1, 0), (5, 83, 11), (128, 1000, 44), (3, 2000, 255)]:
        total = ctypes.c_int64()
        N.check(N.lib.wfl_ctc_workspace(B, T, 10, L, ctypes.byref(total)))
        spans = []
        for field in (N.CTC_WS_REJECTED, N.CTC_WS_STATUS, N.CTC_WS_LOG2Z, N.CTC_WS_ZRANGE):
            off, n = ctypes.c_int64(), ctypes.c_int64()
            N.check(N.lib.wfl_ctc_workspace_field(B, T, L, field, ctypes.byref(off), ctypes.byref(n)))
            assert 0 <= off.value and off.value + n.value <= total.value
            spans.append((off.value, off.value + n.value))
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    off, n = ctypes.c_int64(), ctypes.c_int64()
    assert N.lib.wfl_ctc_workspace_field(4, 100, 10, 99, ctypes.byref(off), ctypes.byref(n)) != 0  # unknown field


def test_dense_workspace_flags_lie_inside_the_workspace():
    """wfl_dense_workspace_field: the per-utterance flags engine.dense_flagged reads are inside what
    wfl_dense_workspace sizes, 8 bytes per utterance, 4-byte aligned; an unknown field is an error."""
    import ctypes

    from gtn_applications_amd import _native as N

    for (B, T, C) in [(1, 1, 3), (5, 83, 28), (128, 1000, 100), (16, 250, 192)]:
        part, total = ctypes.c_int64(), ctypes.c_int64()
        N.check(N.lib.wfl_dense_workspace(B, T, C, ctypes.byref(part), ctypes.byref(total)))
        off, n = ctypes.c_int64(), ctypes.c_int64()
        N.check(N.lib.wfl_dense_workspace_field(B, T, N.DENSE_WS_FLAGS, ctypes.byref(off), ctypes.byref(n)))
        assert n.value == 8 * B and off.value % 4 == 0 and 0 < off.value and off.value + n.value <= total.value
    assert N.lib.wfl_dense_workspace_field(4, 100, 99, ctypes.byref(off), ctypes.byref(n)) != 0


def test_targets_on_device_accepts_tensors_and_lists_alike():
    """the criterion modules pass lists of 1-D tensors (one torch.cat), the functions lists of lists: same flat
    labels, offsets and lengths either way; empty targets included"""
    import torch

    from gtn_applications_amd import engine as E

    rows = [[3, 1, 2], [], [7], [5, 5, 0, 1]]
    cpu = torch.device("cpu")
    a = E.targets_on_device([torch.tensor(r, dtype=torch.long) for r in rows], cpu)
    b = E.CtcTargets(rows, cpu)
    assert a.flat.tolist() == b.flat.tolist() and a.offsets.tolist() == b.offsets.tolist()
    assert list(a.lens) == list(b.lens) and a.max_len == b.max_len == 4 and a.B == b.B == 4
    assert E.targets_on_device(a, cpu) is a  # prebuilt targets pass through


# =================================================================================================
# round 2: graph file formats, the offline transition builder, compat names, label validation
# =================================================================================================
def _sample_graph():
    g = G.Graph()
    for k in range(5):
        g.add_node(k in (0, 4), k in (2, 3))
    g.add_arc(0, 1, 1, G.epsilon, 0.5), g.add_arc(1, 2, G.epsilon, G.epsilon, -2.25)
    g.add_arc(4, 3, 2, 7, 1.5), g.add_arc(0, 3, 3, 3, 1e-3), g.add_arc(3, 3, 0, 0, float("-inf"))
    return g


def test_binary_graph_format_round_trip_and_sniffing(tmp_path):
    import struct

    g = _sample_graph()
    pb, pt = str(tmp_path / "g.bin"), str(tmp_path / "g.txt")
    G.save(pb, g)
    G.savetxt(pt, g)
    assert G.equal(G.load(pb), g) and G.equal(G.load(pt), g)  # load sniffs binary vs text
    raw = open(pb, "rb").read()
    assert len(raw) == 16 + 4 * (2 + 2) + 20 * 5
    assert struct.unpack("<4i", raw[:16]) == (5, 2, 2, 5)
    # the other plausible order of the counts (nodes, arcs, start, accept) is recognised by file size + id ranges
    n, ns, na, m = struct.unpack("<4i", raw[:16])
    alt = str(tmp_path / "alt.bin")
    open(alt, "wb").write(struct.pack("<4i", n, m, ns, na) + raw[16:])
    assert G.equal(G.load(alt), g)
    # anything inconsistent is rejected loudly, never mis-parsed
    for bad in (raw[:-3], raw + b"\0\0\0\0", struct.pack("<4i", 5, 2, 2, 6) + raw[16:], b"\x01\x02"):
        p = str(tmp_path / "bad.bin")
        open(p, "wb").write(bad)
        with pytest.raises(N.WflError):
            G.load(p)
    with pytest.raises(N.WflError):
        G.load(str(tmp_path / "missing.bin"))


def test_load_criterion_reads_transitions_file(tmp_path):
    import gtn_applications_amd as pkg

    trans = TR.make_transitions_graph(2, 4, True)  # 3 tokens + blank
    for name, writer in (("t.bin", G.save), ("t.txt", G.savetxt)):
        path = str(tmp_path / name)
        writer(path, trans)

        class Pre:
            num_tokens = 3
            tokens = ["a", "b", "c"]
            graphemes_to_index = {"a": 0, "b": 1, "c": 2}

        crit, out = pkg.load_criterion("transducer", Pre, {"blank": "optional", "transitions": path})
        assert out == 4 and crit.transition_params.numel() == trans.num_arcs()
        assert G.isomorphic(crit.transitions, TR._zero_weight_view(trans))


def test_transition_builder_matches_reference_script_goldens(golden_dir, tmp_path):
    from gtn_applications_amd import transitions_builder as TB

    with open(os.path.join(golden_dir, "transition_builder.json")) as f:
        gold = json.load(f)
    t2i = {t: e for e, t in enumerate(gold["tokens"])}
    for name, c in gold["cases"].items():
        counts = TB.count_ngrams(gold["lines"], len(c["prune"]), t2i)
        kept = TB.prune_ngrams(counts, c["prune"])
        if c["blank"] != "none":
            kept = TB.add_blank_grams(kept, len(t2i), c["blank"])
        if c["add_self_loops"]:
            kept = TB.add_self_loops(kept)
        assert [[list(g) for g in grams] for grams in kept] == c["kept"], name
        got = dump(TB.build_graph(kept, c["disable_backoff"]))
        want = c["graph"]
        assert (got["num_nodes"], got["start"], got["accept"]) == (want["num_nodes"], want["start"], want["accept"]), name
        assert [a[:4] for a in got["arcs"]] == want["arcs"], name
        whole = TB.build_transitions(gold["lines"], gold["tokens"], c["prune"], c["blank"], c["add_self_loops"],
                                     c["disable_backoff"])
        assert [a[:4] for a in dump(whole)["arcs"]] == want["arcs"], name
    # the CLI writes a file that load_criterion's reader takes back (text by default, the binary layout with --binary)
    data, toks = tmp_path / "text", tmp_path / "tokens"
    data.write_text("\n".join(gold["lines"]) + "\n")
    toks.write_text("\n".join(gold["tokens"]) + "\n")
    for extra in ([], ["--binary"], ["--text"]):
        out = str(tmp_path / ("trans" + "".join(extra)))
        g = TB.main(["--data_path", str(data), "--tokens", str(toks), "--prune", "0", "1", "--blank", "optional",
                     "--save_path", out] + extra)
        assert G.equal(G.load(out), g)
        assert [a[:4] for a in dump(g)["arcs"]] == gold["cases"]["bigram_blank_optional"]["graph"]["arcs"]
    with pytest.raises(ValueError):
        TB.build_transitions(gold["lines"], gold["tokens"], [3, 1])
    with pytest.raises(ValueError):  # a kept bigram whose suffix unigram was pruned
        TB.build_graph([[(0,)], [(TB.START_IDX, 0), (0, 1)]])


def test_compat_registers_models_and_utils_names():
    import sys

    from gtn_applications_amd import compat

    saved = {k: sys.modules.get(k) for k in ("utils", "models", "criterions", "transducer")}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        compat.install()
        import models
        import utils

        assert models.load_criterion is utils.load_criterion and callable(models.load_from_checkpoint)
        assert utils.CTCLoss is ctc.CTCLoss and sys.modules["transducer"] is TR
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def test_target_labels_are_range_checked_before_any_kernel_sees_them():
    class FakeTargets:
        pass

    for flat, ok in (([0, 3, 2], True), ([0, 4], False), ([-1, 2], False), ([], True)):
        tg = FakeTargets()
        arr = np.asarray(flat, np.int32)
        tg.label_min = int(arr.min()) if arr.size else 0
        tg.label_max = int(arr.max()) if arr.size else -1
        if ok:
            E.check_labels(tg, 4, "t")
        else:
            with pytest.raises(ValueError):
                E.check_labels(tg, 4, "t")


@pytest.mark.parametrize("with_transitions", [False, True])
def test_native_batch_builder_equals_per_utterance_graph_algebra(with_transitions):
    """wfl_transducer_pack_batch (thread pool over the batch, transducer.py:262-281,296) produces exactly the
    packed batch that composing / removing / projecting utterance by utterance and then packing does --
    serial and threaded, several times over (the pool is persistent).  For this token graph (blank optional, no
    repeats) the packer writes the alignments down without the composition (wfl_graph_token_alignments): the
    per-utterance side does the same, and that graph is checked to be isomorphic to the composed one."""
    rs = np.random.RandomState(4)
    pieces = ["a", "b", "ab", "ba", "aba", "bab", "c", "ca"]
    g2i = {"a": 0, "b": 1, "c": 2}
    tokens = TR.make_token_graph(pieces, "optional", False)
    lexicon = TR.make_lexicon_graph(pieces, g2i)
    C = len(pieces) + 1
    trans = None
    if with_transitions:
        trans = TR._zero_weight_view(TR.make_transitions_graph(2, C, True))
        trans.arc_sort()
    tokens.arc_sort(True)
    for B in (1, 3, 37):
        rows = [[g2i[ch] for _ in range(rs.randint(1, 6)) for ch in pieces[rs.randint(len(pieces))]] for _ in range(B)]
        graphs, wids = zip(*[TR._alignment_graph(r, tokens, lexicon, trans, direct=True) for r in rows])
        for r, g in zip(rows, graphs):
            assert G.isomorphic(g, TR._alignment_graph(r, tokens, lexicon, trans)[0])
        want = E.PackedLattice.from_graphs(list(graphs), C, None, wids=list(wids) if with_transitions else None)
        flat, off, _ = E.flatten_targets(rows)
        for nthreads in (1, 0, 0):
            got = E.PackedLattice.transducer_batch(tokens, lexicon, trans, flat, off, C, None, nthreads)
            assert bytes(got.desc) == bytes(want.desc)
            np.testing.assert_array_equal(got.host_ints, want.host_ints)
            np.testing.assert_array_equal(got.host_floats, want.host_floats)
    # a target that cannot be spelled with the pieces: an empty acceptor, not an error (loss +inf downstream)
    flat, off, _ = E.flatten_targets([[3, 0], [0]])  # grapheme 3 is in no piece
    got = E.PackedLattice.transducer_batch(tokens, lexicon, trans, flat, off, C, None)
    assert got.desc.B == 2 and list(got.field("state_off", 3))[1] == 0
    # tensors are flattened without tolist()
    import torch

    f2, o2, l2 = E.flatten_any([torch.tensor([2, 2, 1]), torch.tensor([0])])
    assert f2.tolist() == [2, 2, 1, 0] and o2.tolist() == [0, 3, 4] and l2 == [3, 1]


def test_token_alignments_written_down_directly_are_the_composed_ones():
    """wfl_graph_token_alignments against project_input(remove(compose(tokens, tokens_target))) (transducer.py:273-276)
    on the benchmark's 1000 word pieces: isomorphic for targets of 0 .. 15 pieces (repeated pieces, single graphemes,
    several decompositions); other token graphs are refused (NULL -> the caller composes)."""
    import random

    import bench

    pieces, g2i = bench.word_pieces()
    tokens = TR.make_token_graph(pieces, "optional", False)
    lexicon = TR.make_lexicon_graph(pieces, g2i)
    tokens.arc_sort(True)
    rnd = random.Random(3)
    for trial in range(24):
        n = (1, 2, 3, 5, 15)[trial % 5]
        words = [rnd.choice(pieces) for _ in range(n)]
        if trial % 4 == 0 and n > 1:
            words[1] = words[0]  # the same piece twice in a row: only through a blank
        target = [g2i[ch] for w in words for ch in w]
        tt = G.remove(G.project_output(G.compose(TR.make_chain_graph(target), lexicon)))
        direct = G.token_alignments(tokens, tt)
        assert direct is not None
        composed = G.project_input(G.remove(G.compose(tokens, tt)))
        assert direct.num_nodes() == composed.num_nodes() and direct.num_arcs() == composed.num_arcs()
        assert G.isomorphic(direct, composed), words
    small = ["a", "b", "ab"]
    tt = G.remove(G.project_output(G.compose(TR.make_chain_graph([0, 1]), TR.make_lexicon_graph(small, {"a": 0, "b": 1}))))
    for blank, repeats in (("optional", True), ("none", True), ("forced", True)):
        assert G.token_alignments(TR.make_token_graph(small, blank, repeats), tt) is None


def test_host_pool_jobs_with_different_participant_counts():
    """The persistent host pool under jobs whose participant counts alternate (B = 3 uses 2 workers, B = 37 all of
    them), issued from two caller threads: a worker that wakes late must decide from the generation word it saw
    whether it takes part (pack.cpp HostPool) -- every result byte-equal to the serial one, many times over."""
    import threading

    rs = np.random.RandomState(11)
    pieces = ["a", "b", "ab", "ba", "aba", "bab", "c", "ca"]
    g2i = {"a": 0, "b": 1, "c": 2}
    tokens = TR.make_token_graph(pieces, "optional", False)
    lexicon = TR.make_lexicon_graph(pieces, g2i)
    tokens.arc_sort(True)
    C = len(pieces) + 1
    cases = []
    for B in (3, 37, 2, 17):
        rows = [[g2i[ch] for _ in range(rs.randint(1, 6)) for ch in pieces[rs.randint(len(pieces))]] for _ in range(B)]
        flat, off, _ = E.flatten_targets(rows)
        want = E.PackedLattice.transducer_batch(tokens, lexicon, None, flat, off, C, None, 1)
        cases.append((flat, off, np.array(want.host_ints), np.array(want.host_floats)))
    errors = []

    def hammer(order):
        try:
            for it in range(60):
                flat, off, ints, floats = cases[order[it % len(order)]]
                got = E.PackedLattice.transducer_batch(tokens, lexicon, None, flat, off, C, None, 0)
                np.testing.assert_array_equal(got.host_ints, ints)
                np.testing.assert_array_equal(got.host_floats, floats)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=hammer, args=(o,)) for o in ([0, 1, 2, 3], [1, 0, 3, 2])]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[0]


def test_batch_builder_writes_into_the_callers_buffer():
    """wfl_transducer_pack_batch_into: [floats | reserved | pad to 16 B | ints] laid out in the caller's (staging)
    buffer, byte-equal to the library-owned blobs; a buffer that is too small falls back to library storage."""
    import ctypes

    rs = np.random.RandomState(9)
    pieces = ["a", "b", "ab", "ba", "aba", "c", "ca"]
    g2i = {"a": 0, "b": 1, "c": 2}
    tokens = TR.make_token_graph(pieces, "optional", False)
    lexicon = TR.make_lexicon_graph(pieces, g2i)
    tokens.arc_sort(True)
    C = len(pieces) + 1
    rows = [[g2i[ch] for _ in range(rs.randint(1, 6)) for ch in pieces[rs.randint(len(pieces))]] for _ in range(11)]
    flat, off, _ = E.flatten_targets(rows)
    want = E.PackedLattice.transducer_batch(tokens, lexicon, None, flat, off, C, None, 1)
    nf, ni = want.host_floats.size, want.host_ints.size
    for reserve in (0, 5):
        off_i = (4 * (nf + reserve) + 15) & ~15
        for nbytes, fits in ((off_i + 4 * ni + 64, True), (off_i + 4 * ni, True), (off_i + 4 * ni - 4, False)):
            buf = np.full(nbytes, 0xAB, dtype=np.uint8)
            h = N.lib.wfl_transducer_pack_batch_into(tokens._h, lexicon._h, None, flat.ctypes.data, off.ctypes.data,
                                                     len(off) - 1, C, 0, buf.ctypes.data, nbytes, reserve)
            N.check_handle(h)
            try:
                ext = N.lib.wfl_lattice_host_external(h)
                assert ext == (off_i if fits else -1)
                d = N.lib.wfl_lattice_host_desc(h).contents
                assert (int(d.float_words), int(d.int_words)) == (nf, ni)
                if fits:
                    np.testing.assert_array_equal(buf[:4 * nf].view(np.float32), want.host_floats)
                    np.testing.assert_array_equal(buf[off_i:off_i + 4 * ni].view(np.int32), want.host_ints)
                    assert not buf[4 * (nf + reserve):off_i].any()  # the pad is zeroed, the reserve is the caller's
                else:
                    got = np.ctypeslib.as_array(ctypes.cast(N.lib.wfl_lattice_host_ints(h),
                                                            ctypes.POINTER(ctypes.c_int32)), (ni,))
                    np.testing.assert_array_equal(got, want.host_ints)
                    assert (buf == 0xAB).all()
            finally:
                N.lib.wfl_lattice_host_free(h)


def test_asg_has_no_on_chip_class_limit():
    """asg.py:198-199 sizes `transitions` to any N: beyond the on-chip limit of the LDS-resident kernels the entry
    points switch to the batched per-frame product (csrc/dense_wide.h); the workspace query says what it needs."""
    on_chip = N.lib.wfl_dense_on_chip_classes()
    assert 128 <= on_chip <= 256 and asg.max_classes() >= 8192
    crit = asg.ASG(1000, 1, True)
    assert crit.N == 1002 and tuple(crit.transitions.shape) == (1003, 1002)
    part, ws = ctypes.c_int64(), ctypes.c_int64()
    assert N.lib.wfl_dense_workspace(4, 50, 1002, ctypes.byref(part), ctypes.byref(ws)) == 0
    assert part.value % (1002 * 1002) == 0 and ws.value > 2 * 4 * 1002 * 1002  # K slabs of dW; P, P^T + bookkeeping
    assert N.lib.wfl_dense_workspace(4, 50, on_chip, ctypes.byref(part), ctypes.byref(ws)) == 0
    assert ws.value < 1 << 20  # the on-chip kernels' workspace does not hold the matrix
    with pytest.raises(NotImplementedError):
        asg.ASG(asg.max_classes(), 1, True)


def test_cpp_autograd_extension_is_built_and_exports_the_ctc_node():
    """csrc/torch_ops.cpp -> _wfl_torch.so (no compute call: there is no GPU here)."""
    import torch  # noqa: F401
    from gtn_applications_amd import _wfl_torch

    assert callable(_wfl_torch.ctc_step)


def test_cpython_helper_factors_and_content_key():
    """_wflpy: per-utterance loss / gradient factors (ctc.py:53-58,87) and the 128-bit content key of the target cache."""
    from gtn_applications_amd import _wflpy

    lens = [4, 0, 1, 7, 3]
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    B = len(lens)
    fac = np.full((6, B), np.nan, dtype=np.float32)
    _wflpy.factors_into(off.ctypes.data, B, fac.ctypes.data)
    mean = np.array([1.0 / n if n else 1.0 for n in lens], dtype=np.float32)
    np.testing.assert_array_equal(fac[0], np.ones(B, np.float32))
    np.testing.assert_array_equal(fac[1], mean)
    np.testing.assert_allclose(fac[2], np.float32(1.0 / B), rtol=1e-7)
    np.testing.assert_allclose(fac[3], mean / B, rtol=1e-6)
    np.testing.assert_allclose(fac[4], -fac[2], rtol=0)
    np.testing.assert_allclose(fac[5], -fac[3], rtol=0)
    rs = np.random.RandomState(0)
    buf = rs.randint(0, 256, size=1000).astype(np.uint8)
    keys = set()
    for n in (0, 1, 8, 15, 16, 17, 31, 32, 999, 1000):
        k = _wflpy.content_key(buf.ctypes.data, n)
        assert k == _wflpy.content_key(buf.copy().ctypes.data, n)  # content, not address
        keys.add(k)
    assert len(keys) == 10  # (prefixes of different length hash differently)
    other = buf.copy()
    other[500] ^= 1
    assert _wflpy.content_key(other.ctypes.data, 1000) != _wflpy.content_key(buf.ctypes.data, 1000)
    assert _wflpy.same_bytes(buf.ctypes.data, buf.tobytes()) and not _wflpy.same_bytes(other.ctypes.data, buf.tobytes())


def test_numpy_stand_in_of_the_staging_helper_equals_it():
    """engine._staging_helper falls back to _wflpy_np when csrc/wflpy.c was not built: same flattening, same factors,
    same byte comparison (the content key may differ: it only keys a per-process cache)."""
    from gtn_applications_amd import _wflpy, _wflpy_np

    rows = [[3, 1, 4], [], [1, 5, 9, 2, 6], (5, 3)]
    B = len(rows)
    out = {}
    for name, mod in (("c", _wflpy), ("np", _wflpy_np)):
        off = np.zeros(B + 1, np.int64)
        flat = np.full(16, -7, np.int32)
        fac = np.zeros(6 * B, np.float32)
        res = mod.flatten_into(rows, flat.ctypes.data, flat.size, off.ctypes.data)
        mod.factors_into(off.ctypes.data, B, fac.ctypes.data)
        assert mod.flatten_into(rows, flat.ctypes.data, 3, off.ctypes.data) is None and off[B] == 10  # too small: size reported
        key = mod.content_key(flat.ctypes.data, 40)
        assert key == mod.content_key(flat.ctypes.data, 40) and len(key) == 2
        assert mod.same_bytes(flat.ctypes.data, flat[:10].tobytes()) and not mod.same_bytes(flat.ctypes.data, b"\\x01" * 8)
        with pytest.raises(TypeError):
            mod.flatten_into([np.arange(3)], flat.ctypes.data, flat.size, off.ctypes.data)
        out[name] = (res, off.copy(), flat.copy(), fac.copy())
    assert out["c"][0] == out["np"][0] == (10, 5, 1, 9)
    for a, b in zip(out["c"][1:], out["np"][1:]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("blank,repeats", [("none", True), ("optional", True), ("forced", True), ("optional", False)])
def test_batched_decode_written_down_directly_equals_the_graph_algebra(blank, repeats):
    """wfl_transducer_decode_batch (the batch-parallel decode stage of Transducer.viterbi, transducer.py:221-232):
    for every graph make_token_graph can build the direct decode equals, utterance by utterance,
    remove(project_output(viterbi_path(compose(chain(labels), tokens)))) -- on random label sequences, on sequences
    the graph rejects, on labels outside its alphabet and on the empty sequence."""
    rs = np.random.RandomState(7)
    ntok = 5
    tokens = TR.make_token_graph([(i,) for i in range(ntok)], blank=blank, allow_repeats=repeats)
    hi = ntok + (1 if blank != "none" else 0)
    seqs = [rs.randint(0, hi, size=rs.randint(1, 14)).tolist() for _ in range(120)]
    seqs += [rs.randint(0, 2, size=12).tolist() for _ in range(20)]  # (long runs of repeats)
    if blank != "none":
        b = ntok
        seqs += [[b, 1, 1, b, 1, b], [b, 1, b, 2, 2, b, b], [1, b], [b, 1], [b], [b, b], [b, 1, 2, b], [3, 3, b, 3]]
    seqs += [[], [0], [ntok + 1, 0], [0, ntok + 3]]  # (outside the alphabet: the graph algebra decides)
    flat = np.array([v for s in seqs for v in s], np.int32)
    offs = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=offs[1:])
    tokens.arc_sort()
    out, out_off = G.transducer_decode_batch(tokens, flat, offs)
    want = []
    for s in seqs:
        path = G.viterbi_path(G.compose(TR.make_chain_graph(s), tokens))
        want.append(G.remove(G.project_output(path)).labels_to_list())
    got = [out[out_off[i]:out_off[i + 1]].tolist() for i in range(len(seqs))]
    assert got == want
    if blank == "forced":
        assert any(len(w) == 0 and len(s) > 2 for w, s in zip(want, seqs))  # (rejected sequences were among them)
    # the same through the graph algebra on the pool (what any other token graph gets)
    other = G.Graph(False)
    a = tokens.arrays()
    other.add_nodes(a["start"], a["accept"])
    other.add_arcs(a["src"], a["dst"], a["ilabel"], a["olabel"], a["weight"])
    other.add_arc(0, 0, 1000, 1000, 0.0)  # (one arc more: no longer a make_token_graph, same language on these labels)
    out2, off2 = G.transducer_decode_batch(other, flat, offs)
    assert [out2[off2[i]:off2[i + 1]].tolist() for i in range(len(seqs))] == want


def test_batched_viterbi_post_processing_equals_the_row_by_row_spelling():
    """ASG.viterbi / CTC.viterbi collapse, drop and unpack their paths for the whole batch with array operations
    (criterions/asg.py::collapse_and_unpack, engine.collapse_rows); the reference does it row by row
    (asg.py:228-234, ctc.py:130-134).  Same lists."""
    import itertools

    import numpy as np
    import torch

    from gtn_applications_amd import engine as E
    from gtn_applications_amd.criterions import asg

    rs = np.random.RandomState(0)
    for trial in range(60):
        B, T = rs.randint(1, 9), rs.randint(1, 60)
        R = rs.randint(1, 4)
        C = R + rs.randint(1, 6) + 1
        garbage = None if trial % 3 == 0 else C - 1
        # (few classes and long runs: repeats, replabels behind labels / behind replabels / at a row's start, empty results)
        paths = rs.randint(0, C, size=(B, T)).astype(np.int32)
        if trial % 2:
            paths = np.repeat(paths[:, ::3], 3, axis=1)[:, :T]
        want = []
        for row in paths.tolist():
            col = [p for p, _ in itertools.groupby(row)]
            if garbage is not None:
                col = [p for p in col if p != garbage]
            want.append(asg.unpack_replabels(col, R))
        got = asg.collapse_and_unpack(paths, garbage, R)
        assert [g.tolist() for g in got] == want
        assert all(g.dtype == torch.int32 for g in got)
        blank = C - 1
        flat, lens = E.collapse_rows(paths.astype(np.int64), drop=blank)
        rows = [r.tolist() for r in E.split_rows(flat, lens, torch.int64)]
        assert rows == [[p for p in (q for q, _ in itertools.groupby(row)) if p != blank] for row in paths.tolist()]


def test_batched_asg_target_preparation_equals_the_row_by_row_spelling():
    """ASG.forward prepares a batch of tensor targets with array operations (criterions/asg.py::pack_targets_batch); the
    reference packs replabels and interleaves the garbage label target by target (asg.py:201-208).  Same lists."""
    import numpy as np
    import torch

    from gtn_applications_amd.criterions import asg

    rs = np.random.RandomState(1)
    for trial in range(80):
        B = rs.randint(1, 9)
        R = rs.randint(1, 4)
        ntok = rs.randint(1, 4)  # (few tokens: long runs, runs longer than the replabels cover)
        garbage = None if trial % 3 == 0 else ntok + R
        targets = [torch.from_numpy(rs.randint(0, ntok, size=rs.randint(0, 14)).astype(np.int64)) for _ in range(B)]
        want = [asg.pack_replabels(t.tolist(), R) for t in targets]
        if garbage is not None:
            for i, tgt in enumerate(want):
                inter = [garbage] * (2 * len(tgt) + 1)
                inter[1::2] = tgt
                want[i] = inter
        got = asg.pack_targets_batch(targets, R, garbage)
        assert [g.tolist() for g in got] == want
        # and the round trip the criterion relies on (asg.py:35-49)
        if garbage is None:
            assert [asg.unpack_replabels(g.tolist(), R) for g in got] == [t.tolist() for t in targets]


def test_series_log_of_the_general_sweep_is_a_double_precision_log():
    """csrc/lattice_kernels.hip::lse_log (the log of every log-add of the general lattice sweep) restated: mantissa in
    [sqrt(1/2), sqrt(2)), 2 atanh((m - 1) / (m + 1)) by nine terms of its series, + e ln 2.  Against math.log over the
    sums it sees (1 .. a few thousand terms, each <= 1)."""
    import math

    import numpy as np

    rs = np.random.RandomState(0)
    s = np.concatenate([1 + rs.rand(100000) * 10, np.exp(rs.rand(100000) * 14), [1.0, 2.0, 1 + 1e-12, 1e6, 3.999999, 4.0]])
    m, e = np.frexp(s)
    low = m < 0.70710678118654752
    m = np.where(low, m * 2, m)
    e = np.where(low, e - 1, e)
    z = (m - 1) / (m + 1)
    z2 = z * z
    p = np.full_like(z, 1.0 / 17)
    for k in (15, 13, 11, 9, 7, 5, 3, 1):
        p = p * z2 + 1.0 / k
    got = e * 0.69314718055994531 + 2 * z * p
    assert np.abs(got - np.log(s)).max() < 4e-15
    assert got[200000] == 0.0  # log 1
