// Shared helpers for libwfl.so (host side): error reporting and the graph structure.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wfl.h"

namespace wfl {

void set_error(const char* fmt, ...);

// Host WFST.  Arcs are stored SoA in insertion order (arc id == index), which is what
// Graph.set_weights / transition_params rely on (asg.py:66, transducer.py:174-179).
struct Adjacency {
  // CSR over nodes; `idx` holds arc ids ordered by (node, key) where key is the matching label
  std::vector<int64_t> ptr;
  std::vector<int32_t> idx;
  // Label table of the nodes with many out-arcs over a dense label range (the token graph of the Transducer has a
  // thousand arcs per node, one per label): tab[tab_base[n] + (label - lab_lo[n])] = position in the node's range of
  // the first arc with that label, -1 if none.  lab_w[n] == 0: no table, binary search.  One probe instead of ten
  // through an index array into a 4 MB label array (each one a cache miss: 88 ns per composed arc -> 30).
  std::vector<int32_t> lab_lo, lab_w, tab;
  std::vector<int64_t> tab_base;
  // first arc (pointer into idx) of node n whose key equals `lab`, or nullptr; `end` receives the end of n's range
  const int32_t* find(int n, int32_t lab, const std::vector<int32_t>& key, const int32_t*& end) const {
    const int32_t* lo = idx.data() + ptr[n];
    end = idx.data() + ptr[n + 1];
    if (!lab_w.empty() && lab_w[n] > 0) {
      const int64_t off = (int64_t)lab - lab_lo[n];
      if (off < 0 || off >= lab_w[n]) return nullptr;
      const int32_t pos = tab[tab_base[n] + off];
      return pos < 0 ? nullptr : lo + pos;
    }
    // (binary search by hand: <algorithm> is not included here)
    const int32_t* hi = end;
    while (lo < hi) {
      const int32_t* mid = lo + (hi - lo) / 2;
      if (key[*mid] < lab)
        lo = mid + 1;
      else
        hi = mid;
    }
    return (lo != end && key[*lo] == lab) ? lo : nullptr;
  }
};

// Prefix trie over the input-label spellings of a lexicon-shaped transducer (see lexicon_decompose in graph.cpp)
struct LexTrie {
  struct Term {
    int32_t olabel;
    float w;
  };
  // node n: children [child_ptr[n], child_ptr[n+1]) sorted by label; terminals [term_ptr[n], term_ptr[n+1])
  std::vector<int32_t> child_ptr, child_label, child_node, term_ptr;
  std::vector<Term> terms;
};

}  // namespace wfl

struct wfl_graph {
  std::vector<uint8_t> start, accept;
  std::vector<int32_t> src, dst, il, ol;
  std::vector<float> w;
  // user-visible iteration order of arcs per node: -1 insertion, 0 by ilabel, 1 by olabel
  int sort_mode = -1;
  // lazily built, label-sorted adjacency used by compose (independent of sort_mode)
  mutable std::mutex mu;
  mutable bool out_by_il_ok = false, out_by_ol_ok = false;
  mutable wfl::Adjacency out_by_il, out_by_ol;
  // lazily built: 0 not looked at yet, 1 `lex_trie` is valid, -1 the graph is not lexicon-shaped
  mutable int lex_state = 0;
  mutable std::shared_ptr<wfl::LexTrie> lex_trie;
  // lazily found: 0 not looked at yet, N > 0 the graph is make_token_graph(N tokens, blank optional, no repeats), -1 not
  mutable int tok_state = 0;
  // lazily found: 0 not looked at yet, 1 + mode (wfl::kTok*) with `tok_n` tokens, -1 no make_token_graph at all
  mutable int tok_mode = 0, tok_n = 0;

  int num_nodes() const { return (int)start.size(); }
  int64_t num_arcs() const { return (int64_t)src.size(); }
  void invalidate() {
    out_by_il_ok = out_by_ol_ok = false;
    lex_state = 0;
    lex_trie.reset();
    tok_state = 0;
    tok_mode = 0, tok_n = 0;
  }
  const wfl::Adjacency& out_sorted(bool by_olabel) const;
};

namespace wfl {
// All decompositions of `target` into the entries of a lexicon-shaped transducer, as the acceptor
// remove(project_output(compose(chain(target), lexicon))) would give it (transducer.py:269), computed by walking a
// prefix trie of the entries instead of composing with the reference's unshared-prefix lexicon graph.  Returns
// nullptr (no error set) if `lexicon` does not have that shape: the caller composes generically.
wfl_graph* lexicon_decompose(const wfl_graph* lexicon, const int32_t* target, int len);
// All frame-level alignments of the token sequences of `tokens_target` (an acceptor over token labels, one start
// node), as the acceptor project_input(remove(compose(tokens, tokens_target))) would give it (transducer.py:273-276),
// written down directly for the token graph of make_token_graph(N, blank="optional", allow_repeats=False)
// (transducer.py:78-123; the Transducer benchmark's): composing with its N^2 token -> token arcs is half of the
// packer's time per utterance.  Returns nullptr (no error set) if `tokens` does not have that shape or
// `tokens_target` more than one start node: the caller composes generically.  The result is isomorphic to the generic
// one, not identical (node and arc order differ).
wfl_graph* token_alignments(const wfl_graph* tokens, const wfl_graph* tokens_target);
// The four graphs make_token_graph can build (transducer.py:78-123): (blank, allow_repeats)
enum { kTokNoneRepeats = 0, kTokOptionalRepeats = 1, kTokForcedRepeats = 2, kTokOptionalNoRepeats = 3 };
// which of them `tokens` is, arc for arc (cached on the graph), with its number of tokens; -1: none
int token_graph_kind(const wfl_graph* tokens, int* n_tokens);
// the token sequence a frame-label sequence transduces to through such a graph (Transducer.viterbi's decode stage,
// transducer.py:221-229); false (nothing set) if `tokens` is not one of them or a label is outside its alphabet
bool token_decode(const wfl_graph* tokens, const int32_t* labels, int64_t n, std::vector<int32_t>& out);
}  // namespace wfl

struct wfl_lattice_host {
  wfl_lattice_desc desc;
  std::vector<int32_t> ints;
  std::vector<float> floats;
  // >= 0: the blobs were written to the caller's buffer ([floats | reserved | pad to 16 B | ints]) and this is the
  // byte offset of the int blob in it; `ints` / `floats` are then empty (wfl_transducer_pack_batch_into)
  int64_t external_ints_offset = -1;
};
