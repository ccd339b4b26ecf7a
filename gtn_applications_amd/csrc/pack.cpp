// Lattice packer: turns per-utterance acceptors into the flat device format described in
// include/wfl.h (wfl_lattice_desc).  Host only.  The three bulk builders restate the reference's
// label-graph constructors without per-arc host calls:
//   CTC  criterions/ctc.py:15-29, ASG force-align asg.py:72-81 composed with the dense transitions
//   graph asg.py:54-69, STC stc.py:23-64.
#include <pthread.h>
#include <climits>
#include <unistd.h>
#include <sys/syscall.h>
#include <linux/futex.h>
#include <semaphore.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>

#include "common.h"

using wfl::set_error;

namespace {

const float NEG = -std::numeric_limits<float>::infinity();

struct Arc {
  int32_t src, dst, lab, wid, orig;
  float w;
};

// Row stride of the per-utterance compact emission rows (wfl_lattice_desc::max_labels): the widest label set of the
// batch, rounded up to a multiple of four so that every row starts 16-byte aligned (the sweeps load them as float4).
inline int pad_labels(int k) { return (std::max(1, k) + 3) & ~3; }

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
  std::vector<int32_t> level, perm, inv, order, tmp, slot_of, cnt;
  std::vector<Arc> lab_arcs, eps_arcs, sorted;

  // Empties the accumulators but keeps their capacity (a builder cached across batches allocates nothing in steady state).
  void reset(int classes) {
    C = classes;
    for (auto* v : {&state_off, &arc_off, &eps_off, &lab_off, &lvl_off}) v->assign(1, 0);
    for (auto* v : {&in_ptr, &out_ptr, &out_arc, &ein_ptr, &eout_ptr, &eout_arc, &arc_src, &arc_dst, &arc_slot, &arc_lab, &arc_wid,
                    &arc_orig, &eps_src, &eps_dst, &eps_wid, &eps_orig, &labels, &lvl_ptr, &slot_ptr, &slot_arc})
      v->clear();
    for (auto* v : {&arc_w, &eps_w, &start_w, &accept_w}) v->clear();
    max_states = max_arcs = max_eps = max_labels = max_levels = 0;
  }

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
    // distinct labels -> slots (ascending label order).  The distinct ones are collected through the C-sized marker
    // table first: sorting all the arcs' labels (hundreds, most of them repeats) was a fifth of this function.
    if ((int)slot_of.size() < C) slot_of.assign(C, -1);
    tmp.clear();
    for (const Arc& a : lab_arcs)
      if (slot_of[a.lab] < 0) slot_of[a.lab] = 0, tmp.push_back(a.lab);
    std::sort(tmp.begin(), tmp.end());
    const int K = (int)tmp.size();
    for (int k = 0; k < K; ++k) slot_of[tmp[k]] = k, labels.push_back(tmp[k]);
    lab_off.push_back((int32_t)labels.size());

    // stable counting sort of the indices 0..n-1 by key(i) in [0, nkeys): the keys are state / slot numbers, so
    // this is O(n + nkeys) without the temporary buffer std::stable_sort allocates (the packer runs per utterance
    // and per step on the host path of the criteria)
    auto counting_order = [&](int n, int nkeys, auto key, std::vector<int32_t>& out) {
      cnt.assign(nkeys + 1, 0);
      for (int i = 0; i < n; ++i) cnt[key(i) + 1]++;
      for (int k = 0; k < nkeys; ++k) cnt[k + 1] += cnt[k];
      out.resize(n);
      for (int i = 0; i < n; ++i) out[cnt[key(i)]++] = i;
    };
    auto emit_csr = [&](std::vector<Arc>& v, bool labelled) {
      for (Arc& a : v) a.src = inv[a.src], a.dst = inv[a.dst];
      {
        counting_order((int)v.size(), Q, [&](int i) { return v[i].dst; }, order);
        sorted.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) sorted[i] = v[order[i]];
        v.swap(sorted);
      }
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
      counting_order(n, Q, [&](int i) { return v[i].src; }, order);
      const size_t ob = op.size();
      op.resize(ob + Q + 1, 0);
      for (const Arc& a : v) op[ob + a.src + 1]++;
      for (int q = 0; q < Q; ++q) op[ob + q + 1] += op[ob + q];
      oa.insert(oa.end(), order.begin(), order.end());
      if (labelled) {  // by-slot order: the gradient kernel sums the arcs of one emission column without atomics
        counting_order(n, std::max(K, 1), [&](int i) { return slot_of[v[i].lab]; }, order);
        const size_t sb = slot_ptr.size();
        slot_ptr.resize(sb + K + 1, 0);
        for (const Arc& a : v) slot_ptr[sb + slot_of[a.lab] + 1]++;
        for (int k = 0; k < K; ++k) slot_ptr[sb + k + 1] += slot_ptr[sb + k];
        slot_arc.insert(slot_arc.end(), order.begin(), order.end());
      }
      // (one resize per array and indexed stores: seven push_backs per arc were a quarter of this function)
      if (labelled) {
        const size_t at = arc_src.size();
        for (auto* vec : {&arc_src, &arc_dst, &arc_slot, &arc_lab, &arc_wid, &arc_orig}) vec->resize(at + n);
        arc_w.resize(at + n);
        for (int i = 0; i < n; ++i) {
          const Arc& a = v[i];
          arc_src[at + i] = a.src, arc_dst[at + i] = a.dst, arc_slot[at + i] = slot_of[a.lab], arc_lab[at + i] = a.lab;
          arc_wid[at + i] = a.wid, arc_orig[at + i] = a.orig;
          arc_w[at + i] = (a.w != a.w) ? NEG : a.w;  // NaN weight == impossible arc
        }
      } else {
        for (const Arc& a : v) {
          eps_src.push_back(a.src), eps_dst.push_back(a.dst), eps_wid.push_back(a.wid);
          eps_orig.push_back(a.orig), eps_w.push_back((a.w != a.w) ? NEG : a.w);
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

  // The ASG force-alignment acceptor of target y[0..L) in closed form -- exactly what add() produces for its arcs
  // (l-1 -> l and l -> l labelled y[l-1], in that order, weights W[0,c] / W[1+c,prev] / W[1+c,c]; asg.py:72-81), without
  // the generic sorts: a chain's arcs are already grouped by destination, each state has its self loop and the arc to
  // its successor as out-arcs.  (The generic path costs 2.6 us per utterance, this one a few hundred ns; the ASG
  // criterion packs one such acceptor per utterance and step.)
  bool add_force_align(const int32_t* y, int L) {
    const int Q = L + 1, A = 2 * L;
    for (int l = 0; l < L; ++l)
      if (y[l] < 0 || y[l] >= C) {
        set_error("pack_asg_fal: label %d outside [0,%d)", y[l], C);
        return false;
      }
    lvl_ptr.push_back(0), lvl_ptr.push_back(Q);
    lvl_off.push_back((int32_t)lvl_ptr.size());
    for (int q = 0; q < Q; ++q) {
      start_w.push_back(q == 0 ? 0.f : NEG);
      accept_w.push_back(q == L && L > 0 ? 0.f : NEG);  // (asg.py:75-77: no accepting node for an empty target)
    }
    tmp.assign(y, y + L);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    const int K = (int)tmp.size();
    if ((int)slot_of.size() < C) slot_of.assign(C, -1);
    for (int k = 0; k < K; ++k) slot_of[tmp[k]] = k, labels.push_back(tmp[k]);
    lab_off.push_back((int32_t)labels.size());
    // in-arcs of state q >= 1: arcs 2(q-1), 2(q-1)+1;  out-arcs: its self loop 2q-1 (q >= 1), then 2q (q < L)
    in_ptr.push_back(0), out_ptr.push_back(0);
    for (int q = 0; q < Q; ++q) {
      in_ptr.push_back(2 * q);
      out_ptr.push_back(q < L ? 2 * q + 1 : A);
      if (q >= 1) out_arc.push_back(2 * q - 1);
      if (q < L) out_arc.push_back(2 * q);
    }
    ein_ptr.insert(ein_ptr.end(), (size_t)Q + 1, 0), eout_ptr.insert(eout_ptr.end(), (size_t)Q + 1, 0);
    // by-slot order (stable): the arcs of every position whose label has the slot, positions ascending
    const size_t sb = slot_ptr.size();
    slot_ptr.resize(sb + K + 1, 0);
    for (int l = 0; l < L; ++l) slot_ptr[sb + slot_of[y[l]] + 1] += 2;
    for (int k = 0; k < K; ++k) slot_ptr[sb + k + 1] += slot_ptr[sb + k];
    cnt.assign(slot_ptr.begin() + sb, slot_ptr.begin() + sb + K);
    const size_t ab = slot_arc.size();
    slot_arc.resize(ab + A);
    for (int l = 0; l < L; ++l) {
      int32_t& at = cnt[slot_of[y[l]]];
      slot_arc[ab + at] = 2 * l, slot_arc[ab + at + 1] = 2 * l + 1;
      at += 2;
    }
    for (int l = 1; l <= L; ++l) {
      const int32_t c = y[l - 1], slot = slot_of[c];
      const int32_t enter = l == 1 ? c : (1 + c) * C + y[l - 2];  // W[0,c] or W[1+c, prev]
      arc_src.push_back(l - 1), arc_dst.push_back(l), arc_slot.push_back(slot), arc_lab.push_back(c);
      arc_wid.push_back(enter), arc_orig.push_back(2 * (l - 1)), arc_w.push_back(0.f);
      arc_src.push_back(l), arc_dst.push_back(l), arc_slot.push_back(slot), arc_lab.push_back(c);
      arc_wid.push_back((1 + c) * C + c), arc_orig.push_back(2 * (l - 1) + 1), arc_w.push_back(0.f);
    }
    for (int k = 0; k < K; ++k) slot_of[tmp[k]] = -1;
    state_off.push_back(state_off.back() + Q);
    arc_off.push_back(arc_off.back() + A);
    eps_off.push_back(eps_off.back());
    max_states = std::max(max_states, Q), max_arcs = std::max(max_arcs, A);
    max_labels = std::max(max_labels, K), max_levels = std::max(max_levels, 1);
    return true;
  }

  // Appends the utterances of `o` (built independently, e.g. on another host thread) behind this builder's.
  void append(const Builder& o) {
    auto cat = [](auto& dst, const auto& src) { dst.insert(dst.end(), src.begin(), src.end()); };
    for (size_t k = 1; k < o.state_off.size(); ++k) {
      state_off.push_back(state_off.back() + (o.state_off[k] - o.state_off[k - 1]));
      arc_off.push_back(arc_off.back() + (o.arc_off[k] - o.arc_off[k - 1]));
      eps_off.push_back(eps_off.back() + (o.eps_off[k] - o.eps_off[k - 1]));
      lab_off.push_back((int32_t)labels.size() + o.lab_off[k]);
      lvl_off.push_back((int32_t)lvl_ptr.size() + o.lvl_off[k]);
    }
    cat(in_ptr, o.in_ptr), cat(out_ptr, o.out_ptr), cat(out_arc, o.out_arc);
    cat(ein_ptr, o.ein_ptr), cat(eout_ptr, o.eout_ptr), cat(eout_arc, o.eout_arc);
    cat(arc_src, o.arc_src), cat(arc_dst, o.arc_dst), cat(arc_slot, o.arc_slot), cat(arc_lab, o.arc_lab);
    cat(arc_wid, o.arc_wid), cat(arc_orig, o.arc_orig), cat(eps_src, o.eps_src), cat(eps_dst, o.eps_dst);
    cat(eps_wid, o.eps_wid), cat(eps_orig, o.eps_orig), cat(labels, o.labels), cat(lvl_ptr, o.lvl_ptr);
    cat(slot_ptr, o.slot_ptr), cat(slot_arc, o.slot_arc);
    cat(arc_w, o.arc_w), cat(eps_w, o.eps_w), cat(start_w, o.start_w), cat(accept_w, o.accept_w);
    max_states = std::max(max_states, o.max_states), max_arcs = std::max(max_arcs, o.max_arcs);
    max_eps = std::max(max_eps, o.max_eps), max_labels = std::max(max_labels, o.max_labels);
    max_levels = std::max(max_levels, o.max_levels);
  }

  wfl_lattice_host* finish(int B, int shared) {
    auto* h = new wfl_lattice_host();
    wfl_lattice_desc& d = h->desc;
    memset(&d, 0, sizeof(d));
    d.B = B, d.shared = shared;
    d.max_states = max_states, d.max_arcs = max_arcs, d.max_eps = max_eps;
    d.max_labels = pad_labels(max_labels), d.max_levels = max_levels;
    d.total_states = state_off.back(), d.total_arcs = arc_off.back(), d.total_eps = eps_off.back();
    d.total_labels = (int64_t)labels.size();
    {  // one allocation for each blob (every array is padded to a multiple of 4 elements)
      size_t ni = 0, nf = 0;
      for (const auto* v : {&state_off, &arc_off, &eps_off, &lab_off, &lvl_off, &in_ptr, &out_ptr, &out_arc, &ein_ptr,
                            &eout_ptr, &eout_arc, &arc_src, &arc_dst, &arc_slot, &arc_lab, &arc_wid, &eps_src, &eps_dst,
                            &eps_wid, &labels, &lvl_ptr, &arc_orig, &eps_orig, &slot_ptr, &slot_arc})
        ni += (v->size() + 3) & ~(size_t)3;
      for (const auto* v : {&arc_w, &eps_w, &start_w, &accept_w}) nf += (v->size() + 3) & ~(size_t)3;
      h->ints.reserve(ni), h->floats.reserve(nf);
    }
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

// Runs per_utt(b, builder, scratch arcs, start mask, accept mask) for b = 0..B-1 on the host thread pool in
// contiguous ranges (one Builder per range, merged in order): the packed batch is identical to a serial build.
struct PackScratch {
  std::vector<Arc> arcs;
  std::vector<uint8_t> st, ac;
};
wfl_lattice_host* build_batch(int B, int C, const std::function<bool(int, Builder&, PackScratch&)>& per_utt);
wfl_lattice_host* merge_direct(std::vector<Builder>& parts, int np, int B, int C, void* dst = nullptr, int64_t dst_bytes = 0,
                               int64_t reserve_floats = 0);

}  // namespace

// WFL_PACK_TRACE: per-phase host microseconds of the batch packers on stderr (scripts/pack_trace.py)
static bool pack_trace_on() {
  static const bool on = getenv("WFL_PACK_TRACE") != nullptr;
  return on;
}

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
  return build_batch(B, C, [&](int b, Builder& bld, PackScratch& sc) {
    auto& arcs = sc.arcs;
    auto &st = sc.st, &ac = sc.ac;
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
    return bld.add(S, st.data(), ac.data(), arcs);
  });
}

// The ASG force-alignment batch written straight into its final blobs: every array's size follows from the target
// lengths (Q = L + 1 states, 2L arcs, no epsilon arcs) and the number of distinct labels of each target, so one
// sizing pass and one filling pass replace per-utterance builders, their merge and the copy into the blobs.  The
// content is Builder::add_force_align's (== the generic add()'s for the same arcs: tests/test_host_library.py).
wfl_lattice_host* wfl_lattice_pack_asg_fal(const int32_t* targets, const int64_t* offsets, int B, int C) {
  if (B <= 0) {
    set_error("lattice_pack: empty batch");
    return nullptr;
  }
  const int64_t n = offsets[B] - offsets[0];
  if (n >= (1 << 18))  // long batches: the threaded builders
    return build_batch(B, C, [&](int b, Builder& bld, PackScratch&) {
      return bld.add_force_align(targets + offsets[b], (int)(offsets[b + 1] - offsets[b]));
    });
  const bool trace = pack_trace_on();
  const auto t0 = std::chrono::steady_clock::now();
  // pass 1: distinct labels of each target (sorted) and the slot of every position
  // (scratch kept per thread; bound to plain references once -- every access to a thread_local of a shared library
  // is a call into the TLS resolver)
  static thread_local std::vector<int32_t> t_labs, t_lab_cum, t_pos_slot, t_slot_of, t_cnt;
  std::vector<int32_t>&labs = t_labs, &lab_cum = t_lab_cum, &pos_slot = t_pos_slot, &slot_of = t_slot_of, &cnt = t_cnt;
  labs.clear(), lab_cum.assign(1, 0), pos_slot.resize((size_t)n);
  if ((int)slot_of.size() < C) slot_of.assign(C, -1);
  int max_L = 0, max_K = 0;
  for (int b = 0; b < B; ++b) {
    const int32_t* y = targets + offsets[b];
    const int L = (int)(offsets[b + 1] - offsets[b]);
    const size_t l0 = labs.size();
    for (int l = 0; l < L; ++l) {
      if (y[l] < 0 || y[l] >= C) {
        for (size_t k = l0; k < labs.size(); ++k) slot_of[labs[k]] = -1;
        set_error("pack_asg_fal: label %d outside [0,%d)", y[l], C);
        return nullptr;
      }
      if (slot_of[y[l]] < 0) slot_of[y[l]] = 0, labs.push_back(y[l]);
    }
    if (C <= 8 * L) {  // few classes: the marks, scanned in class order, ARE the sorted list
      size_t k = l0;
      for (int c = 0; c < C && k < labs.size(); ++c)
        if (slot_of[c] == 0) labs[k++] = c;
    } else {
      std::sort(labs.begin() + l0, labs.end());
    }
    const int K = (int)(labs.size() - l0);
    for (int k = 0; k < K; ++k) slot_of[labs[l0 + k]] = k;
    int32_t* ps = pos_slot.data() + (offsets[b] - offsets[0]);
    for (int l = 0; l < L; ++l) ps[l] = slot_of[y[l]];
    for (int k = 0; k < K; ++k) slot_of[labs[l0 + k]] = -1;
    lab_cum.push_back((int32_t)labs.size());
    max_L = std::max(max_L, L), max_K = std::max(max_K, K);
  }
  const int64_t S = n + B, A = 2 * n, NL = (int64_t)labs.size();
  auto* h = new wfl_lattice_host();
  wfl_lattice_desc& d = h->desc;
  memset(&d, 0, sizeof(d));
  d.B = B, d.shared = 0;
  d.max_states = max_L + 1, d.max_arcs = 2 * max_L, d.max_eps = 0, d.max_labels = pad_labels(max_K), d.max_levels = 1;
  d.total_states = S, d.total_arcs = A, d.total_eps = 0, d.total_labels = NL;
  int64_t ni = 0, nf = 0;
  auto lay = [](int64_t& cursor, int64_t& field, int64_t count) { field = cursor, cursor += (count + 3) & ~(int64_t)3; };
  lay(ni, d.state_off, B + 1), lay(ni, d.arc_off, B + 1), lay(ni, d.eps_off, B + 1), lay(ni, d.lab_off, B + 1);
  lay(ni, d.lvl_off, B + 1), lay(ni, d.in_ptr, S + B), lay(ni, d.out_ptr, S + B), lay(ni, d.out_arc, A);
  lay(ni, d.ein_ptr, S + B), lay(ni, d.eout_ptr, S + B), lay(ni, d.eout_arc, 0);
  lay(ni, d.arc_src, A), lay(ni, d.arc_dst, A), lay(ni, d.arc_slot, A), lay(ni, d.arc_lab, A), lay(ni, d.arc_wid, A);
  lay(ni, d.eps_src, 0), lay(ni, d.eps_dst, 0), lay(ni, d.eps_wid, 0);
  lay(ni, d.labels, NL), lay(ni, d.lvl_ptr, 2 * (int64_t)B), lay(ni, d.arc_orig, A), lay(ni, d.eps_orig, 0);
  lay(ni, d.slot_ptr, NL + B), lay(ni, d.slot_arc, A);
  lay(nf, d.arc_w, A), lay(nf, d.eps_w, 0), lay(nf, d.start_w, S), lay(nf, d.accept_w, S);
  d.int_words = ni, d.float_words = nf;
  const auto t1 = std::chrono::steady_clock::now();
  h->ints.assign((size_t)ni, 0), h->floats.assign((size_t)nf, 0.f);  // (arc_w, the epsilon CSRs and the padding stay 0)
  const auto t2 = std::chrono::steady_clock::now();
  int32_t* I = h->ints.data();
  float* F = h->floats.data();
  memcpy(I + d.labels, labs.data(), (size_t)NL * sizeof(int32_t));
  // pass 2
  int64_t s0 = 0, a0 = 0;
  for (int b = 0; b < B; ++b) {
    const int32_t* y = targets + offsets[b];
    const int L = (int)(offsets[b + 1] - offsets[b]), Q = L + 1, K = lab_cum[b + 1] - lab_cum[b];
    const int32_t* ps = pos_slot.data() + (offsets[b] - offsets[0]);
    I[d.state_off + b + 1] = (int32_t)(s0 + Q), I[d.arc_off + b + 1] = (int32_t)(a0 + 2 * L);
    I[d.lab_off + b + 1] = lab_cum[b + 1], I[d.lvl_off + b + 1] = 2 * (b + 1);
    I[d.lvl_ptr + 2 * b + 1] = Q;
    int32_t* ip = I + d.in_ptr + s0 + b;
    int32_t* op = I + d.out_ptr + s0 + b;
    int32_t* oa = I + d.out_arc + a0;
    float* sw = F + d.start_w + s0;
    float* aw = F + d.accept_w + s0;
    for (int q = 0; q < Q; ++q) {
      ip[q + 1] = 2 * q, op[q + 1] = q < L ? 2 * q + 1 : 2 * L;
      sw[q] = q == 0 ? 0.f : NEG, aw[q] = (q == L && L > 0) ? 0.f : NEG;
      if (q >= 1) *oa++ = 2 * q - 1;
      if (q < L) *oa++ = 2 * q;
    }
    int32_t* sp = I + d.slot_ptr + lab_cum[b] + b;
    for (int l = 0; l < L; ++l) sp[ps[l] + 1] += 2;
    for (int k = 0; k < K; ++k) sp[k + 1] += sp[k];
    cnt.assign(sp, sp + K);
    int32_t* sa = I + d.slot_arc + a0;
    int32_t *src = I + d.arc_src + a0, *dst = I + d.arc_dst + a0, *slt = I + d.arc_slot + a0, *lab = I + d.arc_lab + a0;
    int32_t *wid = I + d.arc_wid + a0, *org = I + d.arc_orig + a0;
    for (int l = 0; l < L; ++l) {
      const int32_t c = y[l], k = ps[l];
      int32_t& at = cnt[k];
      sa[at] = 2 * l, sa[at + 1] = 2 * l + 1, at += 2;
      src[2 * l] = l, dst[2 * l] = l + 1, src[2 * l + 1] = l + 1, dst[2 * l + 1] = l + 1;
      slt[2 * l] = slt[2 * l + 1] = k, lab[2 * l] = lab[2 * l + 1] = c;
      wid[2 * l] = l == 0 ? c : (1 + c) * C + y[l - 1];  // W[0,c] or W[1+c, prev]
      wid[2 * l + 1] = (1 + c) * C + c;
      org[2 * l] = 2 * l, org[2 * l + 1] = 2 * l + 1;
    }
    s0 += Q, a0 += 2 * L;
  }
  if (trace) {
    auto us = [](auto a, auto b) { return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1000.0; };
    fprintf(stderr, "[wfl pack_asg_fal] slots %.0f us, alloc %.0f us, fill %.0f us\n", us(t0, t1), us(t1, t2),
            us(t2, std::chrono::steady_clock::now()));
  }
  return h;
}

wfl_lattice_host* wfl_lattice_pack_stc(const int32_t* targets, const int64_t* offsets, int B, int star_idx,
                                       float log_prob, int C) {
  const int32_t BLANK = 0;  // stc.py:13
  return build_batch(B, C, [&](int b, Builder& bld, PackScratch& sc) {
    auto& arcs = sc.arcs;
    auto &st = sc.st, &ac = sc.ac;
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
    return bld.add(Q, st.data(), ac.data(), arcs);
  });
}

// ------------------------------------------------------------------------------------------------
// Batch-parallel host side of the Transducer (gtn.parallel_for over process(b), transducer.py:296,327)
// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// A persistent pool of host threads: parallel_for(n, fn) runs fn(0..n-1) on the pool plus the calling thread
// and returns when all are done.  Created on first use, never destroyed (threads die with the process); a
// forked child starts with a fresh pool.  Concurrent callers are serialised.
// Every worker sleeps on its OWN semaphore and the work is handed out through atomics: with one shared condition
// variable the woken threads queued up on its mutex and a batch of 64 x 80 us jobs took 0.7 ms on 32 threads
// (measured on the 256-core host of the MI355X box).
class HostPool {
 public:
  explicit HostPool(int nthreads) : nworkers_(nthreads) {
    sem_init(&done_, 0, 0);
    for (int i = 0; i < nthreads; ++i) std::thread([this, i] { worker(i); }).detach();
  }
  // Wakes the sleeping workers without giving them anything to do: they poll for spin_ns_ and find the job that the
  // caller is about to submit without the futex round trip (the operator calls this when it starts preparing a batch).
  void wake() {
    std::unique_lock<std::mutex> run(run_mu_, std::try_to_lock);
    if (!run.owns_lock() || nworkers_ == 0 || spin_ns_ <= 0) return;
    count_ += 1;
    gen_.store(count_ << 8, std::memory_order_release);  // (0 participants)
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen_), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
  }

  void parallel_for(int n, const std::function<void(int)>& fn) {
    std::lock_guard<std::mutex> run(run_mu_);
    const int k = std::min(nworkers_, std::max(n - 1, 0));  // workers that take part in this job
    if (k == 0) {  // nothing to share: no generation, nobody woken
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    fn_ = &fn, n_ = n;
    next_.store(0, std::memory_order_relaxed);
    running_.store(k + 1, std::memory_order_relaxed);
    // ONE system call wakes every sleeping worker (they sleep on the generation word): posting a semaphore per
    // worker cost the calling thread ~5 us each, 160 us before the last of 32 workers had even been told.
    // The word carries the job's participant count in its low byte: a worker decides whether it takes part from
    // the SAME atomic load that showed it the generation -- a worker that wakes late (after its generation's job is
    // over and while the next one is being set up) still sees the old word, hence the old k, and does not touch
    // fn_ / n_ / next_, which only participants of the published generation may read.
    count_ += 1;
    gen_.store((count_ << 8) | (uint32_t)k, std::memory_order_release);
    if (k > 0) syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen_), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    drain();
    while (sem_wait(&done_) != 0) {
    }
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_acq_rel);
      if (i >= n_) break;
      (*fn_)(i);
    }
    if (running_.fetch_sub(1, std::memory_order_acq_rel) == 1) sem_post(&done_);  // the last one out
  }
  void worker(int id) {
    uint32_t seen = 0;
    for (;;) {
      uint32_t g;
      // Stay hot for a moment before sleeping (WFL_HOST_SPIN_US, default 100; 0: sleep at once).  A batch is packed in
      // two parallel phases a few microseconds apart (build, merge): workers that went to sleep after the first are
      // woken again for the second, and the merge then runs on one thread (see build_batch: it is only shared out
      // when the pool is hot).  Same-box A/B over the fresh-target steps of cfg4, three rounds each: 0.79 / 0.79 /
      // 1.06 ms without, 0.64-0.69 ms with 50, 100 or 200 us; cfg3 unchanged (0.518).  1000 us gained nothing more
      // and put polling workers on the caller's sibling hyperthreads between steps.  The build itself stays at
      // ~350 us for 64 alignment graphs on 32 threads against 60 us each alone: the graph algebra allocates.
      if (spin_ns_ > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0; (g = gen_.load(std::memory_order_acquire)) == seen; ++it) {
          __builtin_ia32_pause();
          if ((it & 63) == 63 &&
              std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > spin_ns_)
            break;
        }
      }
      while ((g = gen_.load(std::memory_order_acquire)) == seen)
        syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen_), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
      seen = g;
      // (a job cannot finish before each of its k workers has drained, so a participant never misses a generation;
      // workers beyond k may skip some, which is all they would have done with them)
      if (id < (int)(g & 0xffu)) drain();
    }
  }
  std::mutex run_mu_;
  const int nworkers_;
  sem_t done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<uint32_t> gen_{0};
  std::atomic<int> next_{0}, running_{0};
  int n_ = 0;
  uint32_t count_ = 0;  // generation counter (upper 24 bits of gen_), advanced under run_mu_
  long long spin_ns_ = 100000;  // how long a worker polls for the next job before it sleeps (WFL_HOST_SPIN_US, microseconds)
 public:
  bool hot() const { return spin_ns_ > 0; }
 private:
};

std::mutex g_pool_mu;
HostPool* g_pool = nullptr;
int g_pool_threads = 0;

void pool_after_fork_child() {  // the parent's threads do not exist in the child: start over (leaks one object)
  new (&g_pool_mu) std::mutex();
  g_pool = nullptr, g_pool_threads = 0;
}

HostPool& host_pool() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (!g_pool) {
    static bool hooked = false;
    if (!hooked) pthread_atfork(nullptr, nullptr, pool_after_fork_child), hooked = true;
    const unsigned hw = std::thread::hardware_concurrency();
    // + the calling thread.  Not one per core: waking a sleeping thread costs microseconds and the jobs are ~0.1 ms
    // each -- measured on the 256-core host (scratch/pool_sweep.py): 24..48 threads are the sweet spot for a batch
    // of 64 utterances (0.6 ms against 4.7 ms serial), 64 and 128 are slower
    g_pool_threads = (int)std::max(1u, std::min(hw ? hw - 1 : 7u, 31u));
    if (const char* e = getenv("WFL_HOST_THREADS")) {  // total threads incl. the caller (tuning / tests)
      const int n = atoi(e);
      if (n >= 1) g_pool_threads = std::min(n - 1, 255);
    }
    g_pool = new HostPool(g_pool_threads);
  }
  return *g_pool;
}

wfl_lattice_host* build_batch(int B, int C, const std::function<bool(int, Builder&, PackScratch&)>& per_utt) {
  if (B <= 0) {
    set_error("lattice_pack: empty batch");
    return nullptr;
  }
  // ranges of at least 8 utterances: below that a Builder's fixed cost outweighs the parallelism
  HostPool& pool = host_pool();
  const int ranges = std::max(1, std::min(g_pool_threads + 1, B / 8));
  std::vector<Builder> parts(ranges);
  std::vector<std::string> errors(ranges);
  std::atomic<int> failed{0};
  auto run = [&](int r) {
    PackScratch sc;
    parts[r].C = C;
    const int lo = (int)((int64_t)B * r / ranges), hi = (int)((int64_t)B * (r + 1) / ranges);
    for (int b = lo; b < hi; ++b)
      if (!per_utt(b, parts[r], sc)) {
        errors[r] = wfl_last_error();
        failed.store(1);
        return;
      }
  };
  if (ranges == 1)
    run(0);
  else
    pool.parallel_for(ranges, run);
  if (failed.load()) {
    for (auto& e : errors)
      if (!e.empty()) {
        set_error("%s", e.c_str());
        break;
      }
    return nullptr;
  }
  if (ranges == 1) return parts[0].finish(B, 0);
  return merge_direct(parts, (int)parts.size(), B, C);
}

// Builders of consecutive utterance ranges -> the final blobs, every bulk array copied ONCE, straight to its place
// (the cumulative offset tables are a few hundred integers).
// The layout is finish()'s: the same arrays in the same order, each padded to a multiple of four elements.
// `dst` (optional, 16-byte aligned, dst_bytes large): the blobs go there instead of into the handle, laid out as the
// operator layer uploads them -- [floats | reserve_floats floats for the caller | pad to 16 B | ints] -- if they fit.
wfl_lattice_host* merge_direct(std::vector<Builder>& parts, int np, int B, int C, void* dst, int64_t dst_bytes,
                               int64_t reserve_floats) {
  using IV = std::vector<int32_t> Builder::*;
  using FV = std::vector<float> Builder::*;
  static const IV bulk_i[] = {&Builder::in_ptr,  &Builder::out_ptr,  &Builder::out_arc,  &Builder::ein_ptr, &Builder::eout_ptr,
                              &Builder::eout_arc, &Builder::arc_src, &Builder::arc_dst,  &Builder::arc_slot, &Builder::arc_lab,
                              &Builder::arc_wid, &Builder::eps_src,  &Builder::eps_dst,  &Builder::eps_wid, &Builder::labels,
                              &Builder::lvl_ptr, &Builder::arc_orig, &Builder::eps_orig, &Builder::slot_ptr, &Builder::slot_arc};
  static const FV bulk_f[] = {&Builder::arc_w, &Builder::eps_w, &Builder::start_w, &Builder::accept_w};
  constexpr int NI = sizeof(bulk_i) / sizeof(bulk_i[0]), NF = sizeof(bulk_f) / sizeof(bulk_f[0]);
  Builder tab;  // the cumulative tables and the maxima
  tab.C = C;
  std::vector<std::vector<int64_t>> ipos(NI, std::vector<int64_t>(np + 1, 0)), fpos(NF, std::vector<int64_t>(np + 1, 0));
  int64_t n_labels = 0, n_lvl = 0;
  for (int p = 0; p < np; ++p) {
    const Builder& o = parts[p];
    for (size_t k = 1; k < o.state_off.size(); ++k) {
      tab.state_off.push_back(tab.state_off.back() + (o.state_off[k] - o.state_off[k - 1]));
      tab.arc_off.push_back(tab.arc_off.back() + (o.arc_off[k] - o.arc_off[k - 1]));
      tab.eps_off.push_back(tab.eps_off.back() + (o.eps_off[k] - o.eps_off[k - 1]));
      tab.lab_off.push_back((int32_t)n_labels + o.lab_off[k]);
      tab.lvl_off.push_back((int32_t)n_lvl + o.lvl_off[k]);
    }
    n_labels += (int64_t)o.labels.size(), n_lvl += (int64_t)o.lvl_ptr.size();
    for (int f = 0; f < NI; ++f) ipos[f][p + 1] = ipos[f][p] + (int64_t)(o.*bulk_i[f]).size();
    for (int f = 0; f < NF; ++f) fpos[f][p + 1] = fpos[f][p] + (int64_t)(o.*bulk_f[f]).size();
    tab.max_states = std::max(tab.max_states, o.max_states), tab.max_arcs = std::max(tab.max_arcs, o.max_arcs);
    tab.max_eps = std::max(tab.max_eps, o.max_eps), tab.max_labels = std::max(tab.max_labels, o.max_labels);
    tab.max_levels = std::max(tab.max_levels, o.max_levels);
  }
  auto* h = new wfl_lattice_host();
  wfl_lattice_desc& d = h->desc;
  memset(&d, 0, sizeof(d));
  d.B = B, d.shared = 0;
  d.max_states = tab.max_states, d.max_arcs = tab.max_arcs, d.max_eps = tab.max_eps;
  d.max_labels = pad_labels(tab.max_labels), d.max_levels = tab.max_levels;
  d.total_states = tab.state_off.back(), d.total_arcs = tab.arc_off.back(), d.total_eps = tab.eps_off.back();
  d.total_labels = n_labels;
  auto pad4 = [](int64_t n) { return (n + 3) & ~(int64_t)3; };
  // descriptor slots in finish()'s order: five tables, then the bulk arrays
  int64_t* const tab_off[] = {&d.state_off, &d.arc_off, &d.eps_off, &d.lab_off, &d.lvl_off};
  const std::vector<int32_t>* const tabs[] = {&tab.state_off, &tab.arc_off, &tab.eps_off, &tab.lab_off, &tab.lvl_off};
  int64_t* const bulk_off[] = {&d.in_ptr,  &d.out_ptr, &d.out_arc, &d.ein_ptr, &d.eout_ptr, &d.eout_arc, &d.arc_src,
                               &d.arc_dst, &d.arc_slot, &d.arc_lab, &d.arc_wid, &d.eps_src, &d.eps_dst,  &d.eps_wid,
                               &d.labels,  &d.lvl_ptr, &d.arc_orig, &d.eps_orig, &d.slot_ptr, &d.slot_arc};
  int64_t* const bulk_foff[] = {&d.arc_w, &d.eps_w, &d.start_w, &d.accept_w};
  int64_t ni = 0;
  for (int t = 0; t < 5; ++t) *tab_off[t] = ni, ni += pad4((int64_t)tabs[t]->size());
  for (int f = 0; f < NI; ++f) *bulk_off[f] = ni, ni += pad4(ipos[f][np]);
  int64_t nf = 0;
  for (int f = 0; f < NF; ++f) *bulk_foff[f] = nf, nf += pad4(fpos[f][np]);
  d.int_words = ni, d.float_words = nf;
  int32_t* I;
  float* F;
  const int64_t off_i = (4 * (nf + reserve_floats) + 15) & ~(int64_t)15;
  if (dst && off_i + 4 * std::max<int64_t>(ni, 1) <= dst_bytes) {
    F = static_cast<float*>(dst), I = reinterpret_cast<int32_t*>(static_cast<char*>(dst) + off_i);
    h->external_ints_offset = off_i;
    // (the gaps that pad every array to four elements: zero, like the blobs the handle would own)
    for (int t = 0; t < 5; ++t)
      for (int64_t k = (int64_t)tabs[t]->size(); k < pad4((int64_t)tabs[t]->size()); ++k) I[*tab_off[t] + k] = 0;
    for (int f = 0; f < NI; ++f)
      for (int64_t k = ipos[f][np]; k < pad4(ipos[f][np]); ++k) I[*bulk_off[f] + k] = 0;
    for (int f = 0; f < NF; ++f)
      for (int64_t k = fpos[f][np]; k < pad4(fpos[f][np]); ++k) F[*bulk_foff[f] + k] = 0.f;
    memset(static_cast<char*>(dst) + 4 * (nf + reserve_floats), 0, (size_t)(off_i - 4 * (nf + reserve_floats)));
  } else {
    h->ints.assign((size_t)ni, 0), h->floats.assign((size_t)nf, 0.f);
    I = h->ints.data(), F = h->floats.data();
  }
  for (int t = 0; t < 5; ++t) memcpy(I + *tab_off[t], tabs[t]->data(), tabs[t]->size() * sizeof(int32_t));
  auto copy_part = [&](int p) {
    const Builder& o = parts[p];
    for (int f = 0; f < NI; ++f) {
      const auto& src = o.*bulk_i[f];
      if (!src.empty()) memcpy(I + *bulk_off[f] + ipos[f][p], src.data(), src.size() * sizeof(int32_t));
    }
    for (int f = 0; f < NF; ++f) {
      const auto& src = o.*bulk_f[f];
      if (!src.empty()) memcpy(F + *bulk_foff[f] + fpos[f][p], src.data(), src.size() * sizeof(float));
    }
  };
  // (serial unless the pool's workers are kept polling: ~2 MB of memcpy takes 50 us here, a second pass over a
  // sleeping pool 140 us in wake-ups alone)
  if (np >= 16 && host_pool().hot())
    host_pool().parallel_for(np, copy_part);
  else
    for (int p = 0; p < np; ++p) copy_part(p);
  return h;
}

struct GraphOwner {  // frees an intermediate graph at scope exit
  wfl_graph* g;
  explicit GraphOwner(wfl_graph* p) : g(p) {}
  ~GraphOwner() { wfl_graph_free(g); }
  GraphOwner(const GraphOwner&) = delete;
  GraphOwner& operator=(const GraphOwner&) = delete;
};

// transducer.py:265-281 for one target: all frame-level alignments of all decompositions of the target into
// tokens, optionally intersected with the transition model (then `wid` = arc of the transition model behind each
// arc, the index of its learnable weight).  Returns false with the thread's error set.
#ifdef WFL_PROFILE_HOST
static std::atomic<long long> g_prof[8];
struct ProfDump { ~ProfDump() { for (int i = 0; i < 8; ++i) fprintf(stderr, "prof[%d] = %.3f ms\n", i, g_prof[i].load() / 1e6); } } g_prof_dump;
#define PROF_T0 auto _t = std::chrono::steady_clock::now();
#define PROF(i) { auto _n = std::chrono::steady_clock::now(); g_prof[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(_n - _t).count(); _t = _n; }
#else
#define PROF_T0
#define PROF(i)
#endif
bool alignment_acceptor(const wfl_graph* tokens, const wfl_graph* lexicon, const wfl_graph* transitions,
                        const int32_t* target, int len, int C, Builder& out) {
  PROF_T0
  // tokens_target = remove(project_output(compose(chain(target), lexicon))): all decompositions into tokens
  wfl_graph* tt = wfl::lexicon_decompose(lexicon, target, len);
  if (!tt) {  // not a make_lexicon_graph-shaped lexicon: the generic graph algebra
    wfl_graph chain;  // make_chain_graph (transducer.py:23-29)
    chain.start.assign(len + 1, 0), chain.accept.assign(len + 1, 0);
    chain.start[0] = 1, chain.accept[len] = 1;
    for (int i = 0; i < len; ++i)
      chain.src.push_back(i), chain.dst.push_back(i + 1), chain.il.push_back(target[i]), chain.ol.push_back(target[i]),
          chain.w.push_back(0.f);
    GraphOwner c1(wfl_graph_compose(&chain, lexicon, nullptr, nullptr));
    if (!c1.g) return false;
    c1.g->il = c1.g->ol;  // project_output in place (c1 is ours)
    tt = wfl_graph_remove(c1.g, WFL_EPSILON, WFL_EPSILON, nullptr);
    if (!tt) return false;
  }
  GraphOwner tokens_target(tt);
  PROF(1)
  // alignments = project_input(remove(compose(tokens, tokens_target))): written down directly for the benchmark's token
  // graph (wfl::token_alignments), the generic graph algebra otherwise
  wfl_graph* direct = wfl::token_alignments(tokens, tokens_target.g);
  PROF(2)
  if (!direct) {
    GraphOwner c2(wfl_graph_compose(tokens, tokens_target.g, nullptr, nullptr));
    if (!c2.g) return false;
    direct = wfl_graph_remove(c2.g, WFL_EPSILON, WFL_EPSILON, nullptr);
    if (!direct) return false;
    direct->ol = direct->il;  // project_input
  }
  GraphOwner ali(direct);
  PROF(3)
  const wfl_graph* fin = ali.g;
  int32_t* prov = nullptr;
  wfl_graph* with_trans = nullptr;
  if (transitions) {
    with_trans = wfl_graph_compose(transitions, ali.g, &prov, nullptr);
    if (!with_trans) return false;
    fin = with_trans;
  }
  std::vector<Arc> arcs;
  const int64_t m = fin->num_arcs();
  arcs.reserve(m);
  for (int64_t a = 0; a < m; ++a)
    arcs.push_back({fin->src[a], fin->dst[a], fin->il[a], prov ? prov[a] : -1, (int32_t)a, transitions ? 0.f : fin->w[a]});
  out.C = C;
  PROF(4)
  const bool ok = out.add(fin->num_nodes(), fin->start.data(), fin->accept.data(), arcs);
  PROF(5)
  if (prov) free(prov);
  if (with_trans) wfl_graph_free(with_trans);
  return ok;
}

}  // namespace

extern "C" {

wfl_lattice_host* wfl_transducer_pack_batch(const wfl_graph* tokens, const wfl_graph* lexicon,
                                            const wfl_graph* transitions, const int32_t* targets,
                                            const int64_t* offsets, int B, int C, int nthreads) {
  return wfl_transducer_pack_batch_into(tokens, lexicon, transitions, targets, offsets, B, C, nthreads, nullptr, 0, 0);
}

wfl_lattice_host* wfl_transducer_pack_batch_into(const wfl_graph* tokens, const wfl_graph* lexicon,
                                                 const wfl_graph* transitions, const int32_t* targets,
                                                 const int64_t* offsets, int B, int C, int nthreads, void* dst,
                                                 int64_t dst_bytes, int64_t reserve_floats) {
  if (!tokens || !lexicon || !targets || !offsets || B <= 0) {
    set_error("transducer_pack_batch: bad arguments");
    return nullptr;
  }
  // build the shared operands' label-sorted adjacency once, before the threads ask for it
  tokens->out_sorted(true), lexicon->out_sorted(false);
  if (transitions) transitions->out_sorted(true);
  const bool trace = pack_trace_on();
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  // The per-utterance builders are kept across batches: their vectors were grown by the pool's threads, and handing
  // ~2 MB of other arenas' memory back to malloc from this thread costs more than building the lattices does
  // (measured: 330 us of free() per batch of 64 against 390 us of parallel build).  A concurrent second caller
  // simply works on builders of its own.
  static std::vector<Builder> cache;
  static std::mutex cache_mu;
  std::unique_lock<std::mutex> cache_lock(cache_mu, std::try_to_lock);
  std::vector<Builder> own;
  std::vector<Builder>& parts = cache_lock.owns_lock() ? cache : own;
  if ((int)parts.size() < B) parts.resize(B);
  for (int b = 0; b < B; ++b) parts[b].reset(C);
  std::vector<std::string> errors(B);
  std::atomic<int> failed{0};
  auto one = [&](int b) {
    if (!alignment_acceptor(tokens, lexicon, transitions, targets + offsets[b], (int)(offsets[b + 1] - offsets[b]), C,
                            parts[b])) {
      errors[b] = wfl_last_error();
      failed.store(1);
    }
  };
  if (nthreads == 1 || B == 1) {
    for (int b = 0; b < B; ++b) one(b);
  } else {
    host_pool().parallel_for(B, one);
  }
  if (failed.load()) {
    for (int b = 0; b < B; ++b)
      if (!errors[b].empty()) {
        set_error("transducer_pack_batch: utterance %d: %s", b, errors[b].c_str());
        break;
      }
    return nullptr;
  }
  auto t1 = now();
  wfl_lattice_host* h = merge_direct(parts, B, B, C, dst, dst_bytes, reserve_floats);
  if (trace) {
    auto t2 = now();
    auto t3 = now();
    auto us = [](auto a, auto b) { return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1000.0; };
    fprintf(stderr, "[wfl pack] build %.0f us, merge %.0f us, free %.0f us\n", us(t0, t1), us(t1, t2), us(t2, t3));
  }
  return h;
}

// Transducer.viterbi's decode stage for a whole batch (transducer.py:221-232: the body of process(b) behind the best
// frame path, under gtn.parallel_for).  The token graphs make_token_graph builds are decoded directly
// (wfl::token_decode); anything else goes through compose / viterbi_path per utterance on the host pool.
int wfl_transducer_decode_batch(const wfl_graph* tokens, const int32_t* labels, const int64_t* offsets, int B,
                                int32_t* out, int64_t out_capacity, int64_t* out_offsets, int nthreads) {
  if (!tokens || !offsets || !out_offsets || B < 0 || (!labels && B > 0 && offsets[B] > offsets[0]) || (!out && out_capacity > 0)) {
    set_error("transducer_decode_batch: bad arguments");
    return WFL_ERR_INVALID;
  }
  for (int b = 0; b < B; ++b)
    if (offsets[b + 1] < offsets[b]) {
      set_error("transducer_decode_batch: offsets must not decrease");
      return WFL_ERR_INVALID;
    }
  int ntok = 0;
  const bool direct = wfl::token_graph_kind(tokens, &ntok) >= 0;
  std::vector<std::vector<int32_t>> parts(B);
  std::vector<std::string> errors(B);
  std::atomic<int> failed{0};
  auto generic = [&](int b) {
    const int32_t* lab = labels + offsets[b];
    const int64_t len = offsets[b + 1] - offsets[b];
    wfl_graph chain;  // make_chain_graph (transducer.py:23-29)
    chain.start.assign(len + 1, 0), chain.accept.assign(len + 1, 0);
    chain.start[0] = 1;
    if (len > 0) chain.accept[len] = 1;
    for (int64_t i = 0; i < len; ++i)
      chain.src.push_back((int32_t)i), chain.dst.push_back((int32_t)i + 1), chain.il.push_back(lab[i]), chain.ol.push_back(lab[i]),
          chain.w.push_back(0.f);
    GraphOwner composed(wfl_graph_compose(&chain, tokens, nullptr, nullptr));
    GraphOwner best(composed.g ? wfl_graph_viterbi_path(composed.g) : nullptr);
    if (!best.g) {
      errors[b] = wfl_last_error();
      failed.store(1);
      return;
    }
    // remove(project_output(path)).labels_to_list(): the path's output labels without the epsilons
    for (int64_t a = 0; a < best.g->num_arcs(); ++a)
      if (best.g->ol[a] != WFL_EPSILON) parts[b].push_back(best.g->ol[a]);
  };
  auto one = [&](int b) {
    if (direct && wfl::token_decode(tokens, labels + offsets[b], offsets[b + 1] - offsets[b], parts[b])) return;
    generic(b);
  };
  if (direct) {
    // a pass over the labels: microseconds per utterance, not worth waking anybody (out-of-alphabet sequences fall
    // through to the graph algebra one by one)
    for (int b = 0; b < B; ++b) one(b);
  } else {
    tokens->out_sorted(false);  // built once, before the threads ask for it
    if (nthreads == 1 || B == 1) {
      for (int b = 0; b < B; ++b) one(b);
    } else {
      host_pool().parallel_for(B, one);
    }
  }
  if (failed.load()) {
    for (int b = 0; b < B; ++b)
      if (!errors[b].empty()) {
        set_error("transducer_decode_batch: utterance %d: %s", b, errors[b].c_str());
        break;
      }
    return WFL_ERR_INVALID;
  }
  int64_t total = 0;
  for (int b = 0; b < B; ++b) {
    out_offsets[b] = total;
    total += (int64_t)parts[b].size();
  }
  out_offsets[B] = total;
  if (total > out_capacity) {
    set_error("transducer_decode_batch: %lld labels do not fit the output buffer (%lld)", (long long)total, (long long)out_capacity);
    return WFL_ERR_INVALID;
  }
  for (int b = 0; b < B; ++b)
    if (!parts[b].empty()) memcpy(out + out_offsets[b], parts[b].data(), parts[b].size() * sizeof(int32_t));
  return WFL_OK;
}

void wfl_host_pool_wake(void) { host_pool().wake(); }

void wfl_lattice_host_free(wfl_lattice_host* h) { delete h; }
int64_t wfl_lattice_host_external(const wfl_lattice_host* h) { return h ? h->external_ints_offset : -1; }
const wfl_lattice_desc* wfl_lattice_host_desc(const wfl_lattice_host* h) { return &h->desc; }
const int32_t* wfl_lattice_host_ints(const wfl_lattice_host* h) { return h->ints.data(); }
const float* wfl_lattice_host_floats(const wfl_lattice_host* h) { return h->floats.data(); }

}  // extern "C"
