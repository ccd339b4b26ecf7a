"""Counterpart of the reference's scripts/make_wordpieces.py: same command line, same two output files
(`<prefix>_tokens_<N>.txt`, `<prefix>_lex_<N>.txt`).  The code lives in gtn_applications_amd/wordpieces.py.

  python scripts/make_wordpieces.py --dataset iamdb --data_dir <iamdb> --output_prefix word_pieces --num_pieces 1000
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.wordpieces import main  # noqa: E402

if __name__ == "__main__":
    main()
