// Shared helpers for libwfl.so (host side): error reporting and the graph structure.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
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

  int num_nodes() const { return (int)start.size(); }
  int64_t num_arcs() const { return (int64_t)src.size(); }
  void invalidate() { out_by_il_ok = out_by_ol_ok = false; }
  const wfl::Adjacency& out_sorted(bool by_olabel) const;
};

struct wfl_lattice_host {
  wfl_lattice_desc desc;
  std::vector<int32_t> ints;
  std::vector<float> floats;
};
