"""
ORACLE -- TEST INFRASTRUCTURE ONLY (runs in the build container, where /root/reference exists).

Pins gtn_applications_amd/wordpieces.py against the reference's scripts/make_wordpieces.py: loads the REFERENCE script
from where it lies (its `utils` import satisfied by a stub that only provides module_from_file -- the real utils.py
imports gtn and the model zoo, neither of which the script uses), runs its own `train_spm_model` / `save_pieces` and its
`iamdb_pieces` route on a seeded toy corpus, and writes inputs + the produced token / lexicon lists as a JSON fixture
(tests/golden/wordpieces_toy.json: data only).  The toy iamdb directory is laid out as datasets/iamdb.py:221-246 reads
it (lines.txt + the split files); the reference's iamdb loader itself imports torchvision (absent here), so its
`load_metadata` is provided to the script by this file's restatement -- the part under test is the script, not the loader.

Usage:  python oracle/pin_wordpieces.py --write-golden
"""
import argparse
import importlib.util
import json
import os
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
WORDSEP = "▁"


def toy_words(rnd):
    syll = ["mo", "ve", "ta", "ri", "on", "el", "st", "an", "qu", "ip", "ka", "lo", "mi", "ne", "ur", "sh", "th", "er"]
    words = ["".join(rnd.choice(syll) for _ in range(rnd.randint(1, 4))) for _ in range(160)]
    return words + ["MOVE", "a/b", "the", "and", "of"]


def toy_corpus(seed=0, n_lines=400):
    rnd = random.Random(seed)
    words = toy_words(rnd)
    weights = [1.0 / (1 + i % 37) for i in range(len(words))]
    return [rnd.choices(words, weights, k=rnd.randint(3, 9)) for _ in range(n_lines)]


def write_toy_iamdb(root, lines):
    """lines.txt in the IAM layout (9+ blank-separated fields, words joined by '|') + split files naming the last
    quarter of the lines (so the script trains on the first three quarters)."""
    keys = []
    with open(os.path.join(root, "lines.txt"), "w") as fid:
        fid.write("#--- lines.txt (toy) ---#\n")
        for i, ws in enumerate(lines):
            key = f"a{i // 40:02d}-{i // 8 % 5:03d}-{i % 8:02d}"
            keys.append(key)
            fid.write(f"{key} ok 154 19 408 746 1661 89 {'|'.join(ws)}\n")
    cut = 3 * len(keys) // 4
    held = keys[cut:]
    for name, part in (("trainset", held[0::4]), ("validationset1", held[1::4]), ("validationset2", held[2::4]),
                       ("testset", held[3::4])):
        with open(os.path.join(root, name + ".txt"), "w") as fid:
            fid.write("\n".join(part) + "\n")


def load_reference_script():
    if not os.path.isdir(REF):
        raise SystemExit("pin_wordpieces: /root/reference is not available here")
    sys.path.insert(0, REPO)
    from gtn_applications_amd import wordpieces as W

    def module_from_file(name, path):  # (what utils.py:38-43 does, minus importing torchvision through iamdb.py)
        assert os.path.basename(path) == "iamdb.py", path
        m = types.ModuleType(name)
        m.SPLITS = dict(W.IAMDB_SPLITS)
        m.load_metadata = lambda data_dir, wordsep: W.iamdb_line_texts(data_dir, wordsep)
        return m

    stub = types.ModuleType("utils")
    stub.module_from_file = module_from_file
    saved = sys.modules.get("utils")
    sys.modules["utils"] = stub
    try:
        spec = importlib.util.spec_from_file_location("ref_make_wordpieces", os.path.join(REF, "scripts", "make_wordpieces.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            sys.modules.pop("utils", None)
        else:
            sys.modules["utils"] = saved
    return mod


def read_outputs(prefix, n):
    with open(prefix + f"_tokens_{n}.txt") as f:
        tokens = f.read().split("\n")
    with open(prefix + f"_lex_{n}.txt") as f:
        lex = [l.rstrip("\n") for l in f]
    return tokens, lex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-golden", action="store_true")
    args = ap.parse_args()
    ref = load_reference_script()
    lines = toy_corpus()
    num = 60
    with tempfile.TemporaryDirectory() as tmp:
        write_toy_iamdb(tmp, lines)
        ns = argparse.Namespace(dataset="iamdb", data_dir=tmp, text_file=None, output_prefix=os.path.join(tmp, "ref"), num_pieces=num)
        ref.iamdb_pieces(ns)
        tokens, lex = read_outputs(ns.output_prefix, num)
        # the script's two helpers on plain sentences (the json-set routes' core, make_wordpieces.py:61-81)
        text = [WORDSEP.join(ws).lower() for ws in lines[:300]]
        sp = ref.train_spm_model(iter(text), 41)
        vocab = sorted(set(w for t in text for w in t.split(WORDSEP) if w))
        ref.save_pieces(sp, 40, os.path.join(tmp, "plain"), vocab)
        tokens2, lex2 = read_outputs(os.path.join(tmp, "plain"), 40)
    out = {"generator": "oracle/pin_wordpieces.py (reference scripts/make_wordpieces.py run on a seeded toy corpus)",
           "corpus_seed": 0, "iamdb": {"lines": lines, "num_pieces": num, "tokens": tokens, "lexicon": lex},
           "plain": {"sentences": text, "num_pieces": 40, "tokens": tokens2, "lexicon": lex2}}
    print(f"iamdb route: {len(tokens)} tokens, {len(lex)} lexicon lines; plain: {len(tokens2)} tokens, {len(lex2)} lines")
    if args.write_golden:
        path = os.path.join(REPO, "tests", "golden", "wordpieces_toy.json")
        with open(path, "w") as f:
            json.dump(out, f, ensure_ascii=False, indent=0)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
