"""gtn_applications_amd/wordpieces.py (counterpart of the reference's scripts/make_wordpieces.py) against
tests/golden/wordpieces_toy.json -- token and lexicon files the REFERENCE script produced from a seeded toy corpus in
the build container (oracle/pin_wordpieces.py).  CPU only (sentencepiece); the trained unigram model is deterministic
for a given corpus and flag set, so the lists must be identical."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

spm = pytest.importorskip("sentencepiece")


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "wordpieces_toy.json")) as f:
        return json.load(f)


def _write_iamdb(root, lines):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pin_wordpieces as PW  # (only its toy-directory writer: test infrastructure calling test infrastructure)

    PW.write_toy_iamdb(str(root), lines)


def _read(prefix, n):
    with open(f"{prefix}_tokens_{n}.txt") as f:
        tokens = f.read().split("\n")
    with open(f"{prefix}_lex_{n}.txt") as f:
        return tokens, [l.rstrip("\n") for l in f]


def test_iamdb_route_through_the_command_line_equals_the_reference_output(golden, tmp_path):
    case = golden["iamdb"]
    _write_iamdb(tmp_path, case["lines"])
    prefix = str(tmp_path / "out")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_wordpieces.py"), "--dataset", "iamdb",
                          "--data_dir", str(tmp_path), "--output_prefix", prefix, "--num_pieces", str(case["num_pieces"])],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "Building word pieces for iamdb" in run.stdout
    tokens, lex = _read(prefix, case["num_pieces"])
    assert tokens == case["tokens"]
    assert lex == case["lexicon"]
    assert "/" in tokens  # the user-defined symbol of make_wordpieces.py:34
    # the token file loads as a Transducer token set and the lexicon spells every word with those tokens
    known = set(tokens)
    for line in lex:
        word, *pieces = line.split(" ")
        assert "".join(pieces).lstrip("▁") == word and all(p in known for p in pieces), line


def test_train_and_save_helpers_equal_the_reference_output(golden, tmp_path):
    from gtn_applications_amd import wordpieces as W

    case = golden["plain"]
    n = case["num_pieces"]
    sp = W.train_spm_model(iter(case["sentences"]), n + 1)
    vocab = W.words_of(case["sentences"])
    W.save_pieces(sp, n, str(tmp_path / "plain"), vocab)
    tokens, lex = _read(str(tmp_path / "plain"), n)
    assert tokens == case["tokens"]
    assert lex == case["lexicon"]


def test_json_set_reader_and_splits(tmp_path):
    """`<split>.json` lines -> transcripts with blanks as word separators (datasets/audioset.py:168-178)."""
    from gtn_applications_amd import wordpieces as W

    with open(tmp_path / "train-clean-100.json", "w") as f:
        f.write(json.dumps({"text": " the quick  fox ", "audio": "x.flac", "duration": 1.0}) + "\n")
        f.write(json.dumps({"text": "jumps"}) + "\n")
    assert W.json_split_texts(str(tmp_path), "train-clean-100") == ["the▁quick▁▁fox", "jumps"]
    assert W.JSON_TRAIN_SPLITS == {"librispeech": ["train-clean-100"], "wsj": ["train_si284"]}
    assert W.words_of(["the▁quick▁▁fox", "the"]) == ["fox", "quick", "the"]
