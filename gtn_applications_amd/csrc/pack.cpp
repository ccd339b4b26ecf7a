// Lattice packer: turns per-utterance acceptors into the flat device format described in
// include/wfl.h (wfl_lattice_desc).  Host only.  The three bulk builders restate the reference's
// label-graph constructors without per-arc host calls:
//   CTC  criterions/ctc.py:15-29, ASG force-align asg.py:72-81 composed with the dense transitions
//   graph asg.py:54-69, STC stc.py:23-64.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include "common.h"

using wfl::set_error;

namespace {

const float NEG = -std::numeric_limits<float>::infinity();

struct Arc {
  int32_t src, dst, lab, wid, orig;
  float w;
};

struct Builder {
  int C = 0;
  // per-field accumulators
  std::vector<int32_t> state_off{0}, arc_off{0}, eps_off{0}, lab_off{0}, lvl_off{0};
  std::vector<int32_t> in_ptr, out_ptr, out_arc, ein_ptr, eout_ptr, eout_arc;
  std::vector<int32_t> arc_src, arc_dst, arc_slot, arc_lab, arc_wid, arc_orig;
  std::vector<int32_t> eps_src, eps_dst, eps_wid, eps_orig;
  std::vector<int32_t> labels, lvl_ptr, slot_ptr, slot_arc;
  std::vector<float> arc_w, eps_w, start_w, accept_w;
  int max_states = 0, max_arcs = 0, max_eps = 0, max_labels = 0, max_levels = 0;
  // scratch reused across utterances
  std::vector<int32_t> level, perm, inv, order, tmp, slot_of;
  std::vector<Arc> lab_arcs, eps_arcs;

  // Adds one utterance.  `arcs` may be reordered.  Returns false (error set) on invalid input.
  bool add(int Q, const uint8_t* start, const uint8_t* accept, std::vector<Arc>& arcs) {
    lab_arcs.clear(), eps_arcs.clear();
    for (const Arc& a : arcs) {
      if (a.src < 0 || a.src >= Q || a.dst < 0 || a.dst >= Q) {
        set_error("lattice_pack: arc endpoint out of range");
        return false;
      }
      if (a.lab == WFL_EPSILON)
        eps_arcs.push_back(a);
      else if (a.lab >= 0 && a.lab < C)
        lab_arcs.push_back(a);
      else {
        set_error("lattice_pack: arc label %d outside [0,%d)", a.lab, C);
        return false;
      }
    }
    // epsilon levels (longest epsilon-path depth); the epsilon subgraph must be acyclic
    level.assign(Q, 0);
    int n_levels = 1;
    if (!eps_arcs.empty()) {
      std::vector<int32_t> indeg(Q, 0);
      std::vector<std::vector<int32_t>> eout(Q);
      for (const Arc& a : eps_arcs) indeg[a.dst]++, eout[a.src].push_back(a.dst);
      std::vector<int32_t> stack;
      for (int q = 0; q < Q; ++q)
        if (!indeg[q]) stack.push_back(q);
      int seen = 0;
      while (!stack.empty()) {
        const int q = stack.back();
        stack.pop_back();
        ++seen;
        for (int d : eout[q]) {
          level[d] = std::max(level[d], level[q] + 1);
          if (--indeg[d] == 0) stack.push_back(d);
        }
      }
      if (seen != Q) {
        set_error("lattice_pack: epsilon arcs form a cycle");
        return false;
      }
      n_levels = 1 + *std::max_element(level.begin(), level.end());
    }
    // renumber states by (level, id)
    perm.resize(Q);  // new -> old
    std::iota(perm.begin(), perm.end(), 0);
    if (n_levels > 1) std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return level[a] < level[b]; });
    inv.resize(Q);  // old -> new
    for (int i = 0; i < Q; ++i) inv[perm[i]] = i;
    lvl_ptr.push_back(0);
    for (int l = 0, i = 0; l < n_levels; ++l) {
      while (i < Q && level[perm[i]] == l) ++i;
      lvl_ptr.push_back(i);
    }
    lvl_off.push_back((int32_t)lvl_ptr.size());
    for (int i = 0; i < Q; ++i) {
      start_w.push_back(start[perm[i]] ? 0.f : NEG);
      accept_w.push_back(accept[perm[i]] ? 0.f : NEG);
    }
    // distinct labels -> slots
    tmp.clear();
    for (const Arc& a : lab_arcs) tmp.push_back(a.lab);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    const int K = (int)tmp.size();
    if ((int)slot_of.size() < C) slot_of.assign(C, -1);
    for (int k = 0; k < K; ++k) slot_of[tmp[k]] = k, labels.push_back(tmp[k]);
    lab_off.push_back((int32_t)labels.size());

    auto emit_csr = [&](std::vector<Arc>& v, bool labelled) {
      for (Arc& a : v) a.src = inv[a.src], a.dst = inv[a.dst];
      std::stable_sort(v.begin(), v.end(), [](const Arc& a, const Arc& b) { return a.dst < b.dst; });
      const int n = (int)v.size();
      std::vector<int32_t>& ip = labelled ? in_ptr : ein_ptr;
      std::vector<int32_t>& op = labelled ? out_ptr : eout_ptr;
      std::vector<int32_t>& oa = labelled ? out_arc : eout_arc;
      // in_ptr
      const size_t ib = ip.size();
      ip.resize(ib + Q + 1, 0);
      for (const Arc& a : v) ip[ib + a.dst + 1]++;
      for (int q = 0; q < Q; ++q) ip[ib + q + 1] += ip[ib + q];
      // out order
      order.resize(n);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return v[a].src < v[b].src; });
      const size_t ob = op.size();
      op.resize(ob + Q + 1, 0);
      for (const Arc& a : v) op[ob + a.src + 1]++;
      for (int q = 0; q < Q; ++q) op[ob + q + 1] += op[ob + q];
      oa.insert(oa.end(), order.begin(), order.end());
      if (labelled) {  // by-slot order: the gradient kernel sums the arcs of one emission column without atomics
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return slot_of[v[a].lab] < slot_of[v[b].lab]; });
        const size_t sb = slot_ptr.size();
        slot_ptr.resize(sb + K + 1, 0);
        for (const Arc& a : v) slot_ptr[sb + slot_of[a.lab] + 1]++;
        for (int k = 0; k < K; ++k) slot_ptr[sb + k + 1] += slot_ptr[sb + k];
        slot_arc.insert(slot_arc.end(), order.begin(), order.end());
      }
      for (const Arc& a : v) {
        const float w = (a.w != a.w) ? NEG : a.w;  // NaN weight == impossible arc
        if (labelled) {
          arc_src.push_back(a.src), arc_dst.push_back(a.dst), arc_slot.push_back(slot_of[a.lab]);
          arc_lab.push_back(a.lab), arc_wid.push_back(a.wid), arc_orig.push_back(a.orig), arc_w.push_back(w);
        } else {
          eps_src.push_back(a.src), eps_dst.push_back(a.dst), eps_wid.push_back(a.wid);
          eps_orig.push_back(a.orig), eps_w.push_back(w);
        }
      }
    };
    emit_csr(lab_arcs, true);
    emit_csr(eps_arcs, false);
    for (int k = 0; k < K; ++k) slot_of[tmp[k]] = -1;
    state_off.push_back(state_off.back() + Q);
    arc_off.push_back(arc_off.back() + (int32_t)lab_arcs.size());
    eps_off.push_back(eps_off.back() + (int32_t)eps_arcs.size());
    max_states = std::max(max_states, Q);
    max_arcs = std::max(max_arcs, (int)lab_arcs.size());
    max_eps = std::max(max_eps, (int)eps_arcs.size());
    max_labels = std::max(max_labels, K);
    max_levels = std::max(max_levels, n_levels);
    return true;
  }

  wfl_lattice_host* finish(int B, int shared) {
    auto* h = new wfl_lattice_host();
    wfl_lattice_desc& d = h->desc;
    memset(&d, 0, sizeof(d));
    d.B = B, d.shared = shared;
    d.max_states = max_states, d.max_arcs = max_arcs, d.max_eps = max_eps;
    d.max_labels = std::max(1, max_labels), d.max_levels = max_levels;
    d.total_states = state_off.back(), d.total_arcs = arc_off.back(), d.total_eps = eps_off.back();
    d.total_labels = (int64_t)labels.size();
    auto put = [&](int64_t& off, const std::vector<int32_t>& v) {
      off = (int64_t)h->ints.size();
      h->ints.insert(h->ints.end(), v.begin(), v.end());
      while (h->ints.size() % 4) h->ints.push_back(0);  // keep every array 16-byte aligned
    };
    put(d.state_off, state_off), put(d.arc_off, arc_off), put(d.eps_off, eps_off), put(d.lab_off, lab_off);
    put(d.lvl_off, lvl_off), put(d.in_ptr, in_ptr), put(d.out_ptr, out_ptr), put(d.out_arc, out_arc);
    put(d.ein_ptr, ein_ptr), put(d.eout_ptr, eout_ptr), put(d.eout_arc, eout_arc);
    put(d.arc_src, arc_src), put(d.arc_dst, arc_dst), put(d.arc_slot, arc_slot), put(d.arc_lab, arc_lab);
    put(d.arc_wid, arc_wid), put(d.eps_src, eps_src), put(d.eps_dst, eps_dst), put(d.eps_wid, eps_wid);
    put(d.labels, labels), put(d.lvl_ptr, lvl_ptr), put(d.arc_orig, arc_orig), put(d.eps_orig, eps_orig);
    put(d.slot_ptr, slot_ptr), put(d.slot_arc, slot_arc);
    d.int_words = (int64_t)h->ints.size();
    auto putf = [&](int64_t& off, const std::vector<float>& v) {
      off = (int64_t)h->floats.size();
      h->floats.insert(h->floats.end(), v.begin(), v.end());
      while (h->floats.size() % 4) h->floats.push_back(0.f);
    };
    putf(d.arc_w, arc_w), putf(d.eps_w, eps_w), putf(d.start_w, start_w), putf(d.accept_w, accept_w);
    d.float_words = (int64_t)h->floats.size();
    return h;
  }
};

}  // namespace

extern "C" {

wfl_lattice_host* wfl_lattice_pack(const wfl_graph* const* graphs, const int32_t* const* wid, int n_graphs, int B,
                                   int shared, int C) {
  if (n_graphs < 1 || (shared && n_graphs != 1) || (!shared && n_graphs != B)) {
    set_error("lattice_pack: need one graph per utterance, or exactly one shared graph");
    return nullptr;
  }
  Builder bld;
  bld.C = C;
  std::vector<Arc> arcs;
  for (int b = 0; b < n_graphs; ++b) {
    const wfl_graph* g = graphs[b];
    if (!g) {
      set_error("lattice_pack: null graph %d", b);
      return nullptr;
    }
    arcs.clear();
    const int64_t m = g->num_arcs();
    for (int64_t a = 0; a < m; ++a) {
      arcs.push_back({g->src[a], g->dst[a], g->il[a], wid && wid[b] ? wid[b][a] : -1, (int32_t)a, g->w[a]});
    }
    if (!bld.add(g->num_nodes(), g->start.data(), g->accept.data(), arcs)) return nullptr;
  }
  return bld.finish(B, shared);
}

wfl_lattice_host* wfl_lattice_pack_ctc(const int32_t* targets, const int64_t* offsets, int B, int blank, int C) {
  if (blank < 0 || blank >= C) {
    set_error("pack_ctc: blank %d outside [0,%d)", blank, C);
    return nullptr;
  }
  Builder bld;
  bld.C = C;
  std::vector<Arc> arcs;
  std::vector<uint8_t> st, ac;
  for (int b = 0; b < B; ++b) {
    const int32_t* y = targets + offsets[b];
    const int L = (int)(offsets[b + 1] - offsets[b]);
    const int S = 2 * L + 1;
    arcs.clear();
    st.assign(S, 0), ac.assign(S, 0);
    st[0] = 1, ac[S - 1] = 1;
    if (S >= 2) ac[S - 2] = 1;
    int32_t id = 0;
    for (int s = 0; s < S; ++s) {
      const int32_t lab = (s & 1) ? y[(s - 1) / 2] : blank;
      arcs.push_back({s, s, lab, -1, id++, 0.f});
      if (s > 0) arcs.push_back({s - 1, s, lab, -1, id++, 0.f});
      if ((s & 1) && s > 1 && lab != y[(s - 1) / 2 - 1]) arcs.push_back({s - 2, s, lab, -1, id++, 0.f});
    }
    if (!bld.add(S, st.data(), ac.data(), arcs)) return nullptr;
  }
  return bld.finish(B, 0);
}

wfl_lattice_host* wfl_lattice_pack_asg_fal(const int32_t* targets, const int64_t* offsets, int B, int C) {
  Builder bld;
  bld.C = C;
  std::vector<Arc> arcs;
  std::vector<uint8_t> st, ac;
  for (int b = 0; b < B; ++b) {
    const int32_t* y = targets + offsets[b];
    const int L = (int)(offsets[b + 1] - offsets[b]);
    arcs.clear();
    st.assign(L + 1, 0), ac.assign(L + 1, 0);
    st[0] = 1, ac[L] = 1;
    if (L == 0) ac[0] = 0;  // asg.py:75-77: no accepting node for an empty target
    int32_t id = 0;
    for (int l = 1; l <= L; ++l) {
      const int32_t c = y[l - 1];
      if (c < 0 || c >= C) {
        set_error("pack_asg_fal: label %d outside [0,%d)", c, C);
        return nullptr;
      }
      const int32_t enter = (l == 1) ? c : (1 + c) * C + y[l - 2];  // W[0,c] or W[1+c, prev]
      arcs.push_back({l - 1, l, c, enter, id++, 0.f});
      arcs.push_back({l, l, c, (1 + c) * C + c, id++, 0.f});
    }
    if (!bld.add(L + 1, st.data(), ac.data(), arcs)) return nullptr;
  }
  return bld.finish(B, 0);
}

wfl_lattice_host* wfl_lattice_pack_stc(const int32_t* targets, const int64_t* offsets, int B, int star_idx,
                                       float log_prob, int C) {
  Builder bld;
  bld.C = C;
  std::vector<Arc> arcs;
  std::vector<uint8_t> st, ac;
  const int32_t BLANK = 0;  // stc.py:13
  for (int b = 0; b < B; ++b) {
    const int32_t* y = targets + offsets[b];
    const int L = (int)(offsets[b + 1] - offsets[b]);
    const int S = 2 * L + 1, Q = S + L + 1;
    arcs.clear();
    st.assign(Q, 0), ac.assign(Q, 0);
    st[0] = 1, ac[S - 1] = 1;
    if (S >= 2) ac[S - 2] = 1;
    int32_t id = 0;
    for (int s = 0; s < S; ++s) {
      const int32_t lab = (s & 1) ? y[(s - 1) / 2] : BLANK;
      if (lab == BLANK) arcs.push_back({s, s, lab, -1, id++, 0.f});
      if (s > 0) arcs.push_back({s - 1, s, lab, -1, id++, 0.f});
      if ((s & 1) && s > 1) arcs.push_back({s - 2, s, lab, -1, id++, 0.f});
    }
    for (int l = 0; l <= L; ++l) {
      const int p1 = 2 * l - 1, p2 = 2 * l, c = S + l;
      if (l == L) ac[c] = 1;
      const int32_t star = (l == L) ? star_idx : star_idx + y[l];
      if (p1 >= 0) arcs.push_back({p1, c, star, -1, id++, log_prob});
      arcs.push_back({p2, c, star, -1, id++, log_prob});
      arcs.push_back({c, c, star, -1, id++, log_prob});
      if (l < L) arcs.push_back({c, 2 * l + 1, y[l], -1, id++, 0.f});
      arcs.push_back({c, p2, BLANK, -1, id++, 0.f});
    }
    if (!bld.add(Q, st.data(), ac.data(), arcs)) return nullptr;
  }
  return bld.finish(B, 0);
}

void wfl_lattice_host_free(wfl_lattice_host* h) { delete h; }
const wfl_lattice_desc* wfl_lattice_host_desc(const wfl_lattice_host* h) { return &h->desc; }
const int32_t* wfl_lattice_host_ints(const wfl_lattice_host* h) { return h->ints.data(); }
const float* wfl_lattice_host_floats(const wfl_lattice_host* h) { return h->floats.data(); }

}  // extern "C"
