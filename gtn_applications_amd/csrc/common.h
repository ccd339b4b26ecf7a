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

  int num_nodes() const { return (int)start.size(); }
  int64_t num_arcs() const { return (int64_t)src.size(); }
  void invalidate() {
    out_by_il_ok = out_by_ol_ok = false;
    lex_state = 0;
    lex_trie.reset();
  }
  const wfl::Adjacency& out_sorted(bool by_olabel) const;
};

namespace wfl {
// All decompositions of `target` into the entries of a lexicon-shaped transducer, as the acceptor
// remove(project_output(compose(chain(target), lexicon))) would give it (transducer.py:269), computed by walking a
// prefix trie of the entries instead of composing with the reference's unshared-prefix lexicon graph.  Returns
// nullptr (no error set) if `lexicon` does not have that shape: the caller composes generically.
wfl_graph* lexicon_decompose(const wfl_graph* lexicon, const int32_t* target, int len);
}  // namespace wfl

struct wfl_lattice_host {
  wfl_lattice_desc desc;
  std::vector<int32_t> ints;
  std::vector<float> floats;
};
