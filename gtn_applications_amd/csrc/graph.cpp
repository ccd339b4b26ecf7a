// Host WFST library of libwfl.so: the graph type and the graph functions the criteria call
// (compose/intersect, remove, project, viterbi_path on small graphs, equal/isomorphic, text I/O).
// Replaces the corresponding entry points of the external `gtn` library; the
// call sites are cited in include/wfl.h.  Pure host code, never touches the GPU.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <iterator>
#include <functional>
#include <limits>
#include <numeric>
#include <sstream>
#include <unordered_map>

#include "common.h"

namespace wfl {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

static void build_sorted(const wfl_graph& g, bool by_ol, Adjacency& adj) {
  const int n = g.num_nodes();
  const int64_t m = g.num_arcs();
  adj.ptr.assign(n + 1, 0);
  for (int64_t a = 0; a < m; ++a) adj.ptr[g.src[a] + 1]++;
  for (int i = 0; i < n; ++i) adj.ptr[i + 1] += adj.ptr[i];
  adj.idx.resize(m);
  std::vector<int64_t> fill(adj.ptr.begin(), adj.ptr.end() - 1);
  for (int64_t a = 0; a < m; ++a) adj.idx[fill[g.src[a]]++] = (int32_t)a;
  const std::vector<int32_t>& key = by_ol ? g.ol : g.il;
  for (int i = 0; i < n; ++i)
    std::stable_sort(adj.idx.begin() + adj.ptr[i], adj.idx.begin() + adj.ptr[i + 1],
                     [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  // label tables (see Adjacency): nodes with at least 32 out-arcs whose labels span at most 4 x their number
  adj.lab_lo.assign(n, 0), adj.lab_w.assign(n, 0), adj.tab_base.assign(n, 0), adj.tab.clear();
  for (int i = 0; i < n; ++i) {
    const int64_t b = adj.ptr[i], e = adj.ptr[i + 1];
    if (e - b < 32) continue;
    const int32_t lo = key[adj.idx[b]], hi = key[adj.idx[e - 1]];  // (sorted; epsilon = -1 simply widens the range by one)
    const int64_t width = (int64_t)hi - lo + 1;
    if (width > 4 * (e - b)) continue;
    adj.lab_lo[i] = lo, adj.lab_w[i] = (int32_t)width, adj.tab_base[i] = (int64_t)adj.tab.size();
    adj.tab.resize(adj.tab.size() + (size_t)width, -1);
    int32_t* t = adj.tab.data() + adj.tab_base[i];
    for (int64_t k = e - 1; k >= b; --k) t[key[adj.idx[k]] - lo] = (int32_t)(k - b);  // (downwards: the FIRST arc wins)
  }
}

// A graph is lexicon-shaped (what make_lexicon_graph builds, transducer.py:61-75) if node 0 is its only start and
// only accept node and every arc lies on a simple path 0 -> n1 -> ... -> 0 whose inner nodes have exactly one in- and
// one out-arc, whose input labels are not epsilon and whose output labels are epsilon except on the last arc.
static std::shared_ptr<LexTrie> build_lex_trie(const wfl_graph& g) {
  const int n = g.num_nodes();
  const int64_t m = g.num_arcs();
  if (n < 1 || !g.start[0] || !g.accept[0]) return nullptr;
  for (int i = 1; i < n; ++i)
    if (g.start[i] || g.accept[i]) return nullptr;
  std::vector<int32_t> out_arc(n, -1), out_deg(n, 0), in_deg(n, 0);
  for (int64_t a = 0; a < m; ++a) {
    if (g.il[a] == WFL_EPSILON) return nullptr;
    out_deg[g.src[a]]++, in_deg[g.dst[a]]++;
    if (g.src[a] != 0) out_arc[g.src[a]] = (int32_t)a;
  }
  for (int i = 1; i < n; ++i)
    if (out_deg[i] != 1 || in_deg[i] != 1) return nullptr;
  // spellings in arc order of their first arc (keeps the relative order of the entries)
  struct Node {
    std::vector<std::pair<int32_t, int32_t>> kids;  // (label, node)
    std::vector<LexTrie::Term> terms;
  };
  std::vector<Node> nodes(1);
  int64_t used = 0;
  for (int64_t a0 = 0; a0 < m; ++a0) {
    if (g.src[a0] != 0) continue;
    int cur = 0;
    float w = 0.f;
    int64_t a = a0;
    for (int steps = 0;; ++steps) {
      if (steps > n) return nullptr;
      ++used;
      w += g.w[a];
      int next = -1;
      for (auto& kv : nodes[cur].kids)
        if (kv.first == g.il[a]) next = kv.second;
      if (next < 0) {
        next = (int)nodes.size();
        nodes[cur].kids.emplace_back(g.il[a], next);
        nodes.emplace_back();
      }
      cur = next;
      if (g.dst[a] == 0) {
        nodes[cur].terms.push_back({g.ol[a], w});
        break;
      }
      if (g.ol[a] != WFL_EPSILON) return nullptr;
      a = out_arc[g.dst[a]];
      if (a < 0) return nullptr;
    }
  }
  if (used != m) return nullptr;
  auto t = std::make_shared<LexTrie>();
  t->child_ptr.push_back(0), t->term_ptr.push_back(0);
  for (auto& nd : nodes) {
    std::sort(nd.kids.begin(), nd.kids.end());
    for (auto& kv : nd.kids) t->child_label.push_back(kv.first), t->child_node.push_back(kv.second);
    t->child_ptr.push_back((int32_t)t->child_label.size());
    for (auto& tm : nd.terms) t->terms.push_back(tm);
    t->term_ptr.push_back((int32_t)t->terms.size());
  }
  return t;
}

wfl_graph* lexicon_decompose(const wfl_graph* lexicon, const int32_t* target, int len) {
  std::shared_ptr<LexTrie> trie;
  {
    std::lock_guard<std::mutex> lock(lexicon->mu);
    if (lexicon->lex_state == 0) {
      lexicon->lex_trie = build_lex_trie(*lexicon);
      lexicon->lex_state = lexicon->lex_trie ? 1 : -1;
    }
    if (lexicon->lex_state < 0) return nullptr;
    trie = lexicon->lex_trie;
  }
  const LexTrie& t = *trie;
  struct A {
    int32_t s, d, lab;
    float w;
  };
  std::vector<A> arcs;
  std::vector<uint8_t> reach(len + 1, 0), co(len + 1, 0);
  reach[0] = 1;
  for (int i = 0; i < len; ++i) {
    if (!reach[i]) continue;
    int node = 0;
    for (int j = i; j < len; ++j) {
      const int32_t* lo = t.child_label.data() + t.child_ptr[node];
      const int32_t* hi = t.child_label.data() + t.child_ptr[node + 1];
      const int32_t* it = std::lower_bound(lo, hi, target[j]);
      if (it == hi || *it != target[j]) break;
      node = t.child_node[it - t.child_label.data()];
      for (int k = t.term_ptr[node]; k < t.term_ptr[node + 1]; ++k) {
        arcs.push_back({i, j + 1, t.terms[k].olabel, t.terms[k].w});
        reach[j + 1] = 1;
      }
    }
  }
  co[len] = 1;
  for (size_t k = arcs.size(); k-- > 0;)  // arcs are ordered by source position: one reverse pass settles it
    if (co[arcs[k].d]) co[arcs[k].s] = 1;
  auto* out = new wfl_graph();
  std::vector<int32_t> id(len + 1, -1);
  for (int i = 0; i <= len; ++i)
    if (reach[i] && co[i]) {
      id[i] = out->num_nodes();
      out->start.push_back(i == 0), out->accept.push_back(i == len);
    }
  for (const A& a : arcs)
    if (id[a.s] >= 0 && id[a.d] >= 0) {
      out->src.push_back(id[a.s]), out->dst.push_back(id[a.d]);
      out->il.push_back(a.lab), out->ol.push_back(a.lab), out->w.push_back(a.w);
    }
  return out;
}

// Is `g` exactly what make_token_graph(N, "optional", False) builds?  Node 0 start + accept (idle), nodes 1..N the tokens
// (accept), node N + 1 the blank; arcs: 0 -> blank [N : eps], blank -> 0 [eps : eps], then per token i:
// 0 -> i+1 [i : i], i+1 -> i+1 [i : eps], i+1 -> blank [N : eps], i+1 -> j+1 [j : j] for every j != i.
static int token_graph_shape(const wfl_graph& g) {
  const int64_t nodes = g.num_nodes(), arcs = g.num_arcs();
  const int64_t N = nodes - 2;
  if (N < 1 || arcs != 2 + N * (3 + (N - 1))) return -1;
  for (int64_t n = 0; n < nodes; ++n)
    if (g.start[n] != (n == 0) || g.accept[n] != (n <= N)) return -1;
  auto is = [&](int64_t a, int64_t s, int64_t d, int64_t il, int64_t ol) {
    return g.src[a] == s && g.dst[a] == d && g.il[a] == il && g.ol[a] == ol && g.w[a] == 0.f;
  };
  const int64_t B = N + 1;
  if (!is(0, 0, B, N, WFL_EPSILON) || !is(1, B, 0, WFL_EPSILON, WFL_EPSILON)) return -1;
  int64_t a = 2;
  for (int64_t i = 0; i < N; ++i) {
    if (!is(a, 0, i + 1, i, i) || !is(a + 1, i + 1, i + 1, i, WFL_EPSILON) || !is(a + 2, i + 1, B, N, WFL_EPSILON)) return -1;
    a += 3;
    for (int64_t j = 0; j < N; ++j)
      if (j != i) {
        if (!is(a, i + 1, j + 1, j, j)) return -1;
        ++a;
      }
  }
  return (int)N;
}

// Which make_token_graph(N tokens, blank, allow_repeats) (transducer.py:78-123) is `g`, arc for arc?  Returns the mode
// (kTokNoneRepeats ...) and N, or -1.  All four graphs the factory can build transduce a frame-label sequence the same
// way wherever they accept it -- collapse repeated labels, drop the blank N -- which is what token_decode relies on.
static int token_graph_mode(const wfl_graph& g, int* n_tokens) {
  const int64_t nodes = g.num_nodes(), arcs = g.num_arcs();
  auto is = [&](int64_t a, int64_t s, int64_t d, int64_t il, int64_t ol) {
    return a < arcs && g.src[a] == s && g.dst[a] == d && g.il[a] == il && g.ol[a] == ol && g.w[a] == 0.f;
  };
  auto flags = [&](int64_t N, bool blank, bool tok_accept) {
    if (nodes != N + 1 + (blank ? 1 : 0)) return false;
    for (int64_t n = 0; n < nodes; ++n) {
      const bool st = n == 0, ac = n == 0 || (n <= N && tok_accept);
      if ((g.start[n] != 0) != st || (g.accept[n] != 0) != ac) return false;
    }
    return true;
  };
  // blank "none", repeats allowed: N + 1 nodes, per token  0 -> i+1 [i : i], i+1 -> i+1 [i : eps], i+1 -> 0 [eps : eps]
  if (nodes >= 2 && arcs == 3 * (nodes - 1)) {
    const int64_t N = nodes - 1;
    bool ok = flags(N, false, true);
    for (int64_t i = 0; ok && i < N; ++i)
      ok = is(3 * i, 0, i + 1, i, i) && is(3 * i + 1, i + 1, i + 1, i, WFL_EPSILON) && is(3 * i + 2, i + 1, 0, WFL_EPSILON, WFL_EPSILON);
    if (ok) return *n_tokens = (int)N, kTokNoneRepeats;
  }
  if (nodes >= 3 && arcs == 2 + 3 * (nodes - 2)) {
    const int64_t N = nodes - 2, B = N + 1;
    if (is(0, 0, B, N, WFL_EPSILON) && is(1, B, 0, WFL_EPSILON, WFL_EPSILON)) {
      // blank "optional", repeats allowed: 0 -> i+1 [i : i], i+1 -> i+1 [i : eps], i+1 -> 0 [eps : eps]
      bool ok = flags(N, true, true);
      for (int64_t i = 0; ok && i < N; ++i)
        ok = is(2 + 3 * i, 0, i + 1, i, i) && is(3 + 3 * i, i + 1, i + 1, i, WFL_EPSILON) && is(4 + 3 * i, i + 1, 0, WFL_EPSILON, WFL_EPSILON);
      if (ok) return *n_tokens = (int)N, kTokOptionalRepeats;
      // blank "forced", repeats allowed: blank -> i+1 [i : i], i+1 -> i+1 [i : eps], i+1 -> blank [N : eps]
      ok = flags(N, true, false);
      for (int64_t i = 0; ok && i < N; ++i)
        ok = is(2 + 3 * i, B, i + 1, i, i) && is(3 + 3 * i, i + 1, i + 1, i, WFL_EPSILON) && is(4 + 3 * i, i + 1, B, N, WFL_EPSILON);
      if (ok) return *n_tokens = (int)N, kTokForcedRepeats;
    }
  }
  const int N = token_graph_shape(g);  // blank "optional", no repeats
  if (N > 0) return *n_tokens = N, kTokOptionalNoRepeats;
  return -1;
}

int token_graph_kind(const wfl_graph* tokens, int* n_tokens) {
  std::lock_guard<std::mutex> lock(tokens->mu);
  if (tokens->tok_mode == 0) {
    int n = 0;
    const int m = token_graph_mode(*tokens, &n);
    tokens->tok_mode = m < 0 ? -1 : 1 + m;
    tokens->tok_n = n;
  }
  *n_tokens = tokens->tok_n;
  return tokens->tok_mode < 0 ? -1 : tokens->tok_mode - 1;
}

// remove(project_output(viterbi_path(compose(chain(labels), tokens)))) (transducer.py:221-229) for one frame-label
// sequence, written down directly for the token graphs of make_token_graph.  With those (all weights zero) every
// accepted sequence has exactly one path with the fewest output labels -- a token's self-loop wherever a label repeats
// -- so the shortest decoding the reference picks is: collapse repeats, drop blanks; and the graphs differ only in what
// they accept.  Returns false if `tokens` is none of them or a label is outside its alphabet (the caller composes).
bool token_decode(const wfl_graph* tokens, const int32_t* labels, int64_t n, std::vector<int32_t>& out) {
  int N = 0;
  const int mode = token_graph_kind(tokens, &N);
  if (mode < 0) return false;
  const int32_t blank = mode == kTokNoneRepeats ? -2 : N;  // (no blank label in the alphabet of the first graph)
  for (int64_t i = 0; i < n; ++i)
    if (labels[i] < 0 || (labels[i] >= N && labels[i] != blank)) return false;
  out.clear();
  if (n == 0) return true;  // (a chain without arcs has no accepting node: no path, no labels)
  if (mode == kTokForcedRepeats) {
    // accepted iff the sequence starts and ends with a blank and tokens are separated by blanks
    bool ok = labels[0] == blank && labels[n - 1] == blank;
    for (int64_t i = 1; ok && i < n; ++i)
      ok = labels[i] == blank || labels[i - 1] == blank || labels[i] == labels[i - 1];
    if (!ok) return true;  // no accepting path: viterbi_path gives the empty graph
  }
  int32_t prev = -3;
  for (int64_t i = 0; i < n; ++i) {
    if (labels[i] != blank && labels[i] != prev) out.push_back(labels[i]);
    prev = labels[i];
  }
  return true;
}

wfl_graph* token_alignments(const wfl_graph* tokens, const wfl_graph* tt) {
  int N;
  {
    std::lock_guard<std::mutex> lock(tokens->mu);
    if (tokens->tok_state == 0) tokens->tok_state = token_graph_shape(*tokens);
    N = tokens->tok_state;
  }
  if (N < 0) return nullptr;
  const int M = tt->num_nodes();
  const int64_t A = tt->num_arcs();
  int u0 = -1;
  for (int u = 0; u < M; ++u)
    if (tt->start[u]) {
      if (u0 >= 0) return nullptr;
      u0 = u;
    }
  if (u0 < 0) return nullptr;
  for (int64_t a = 0; a < A; ++a)
    if (tt->il[a] < 0 || tt->il[a] >= N || tt->il[a] != tt->ol[a]) return nullptr;  // (an acceptor over the tokens)
  // out-arcs of tokens_target by node (counting sort: a few hundred arcs)
  std::vector<int32_t> ptr(M + 1, 0), order(A);
  for (int64_t a = 0; a < A; ++a) ++ptr[tt->src[a] + 1];
  for (int u = 0; u < M; ++u) ptr[u + 1] += ptr[u];
  {
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t a = 0; a < A; ++a) order[fill[tt->src[a]]++] = (int32_t)a;
  }
  // states: 0 = (idle, u0) | 1 + u = (blank, u) | then one per distinct (token k, node v) some arc of tokens_target
  // enters v with; `tok_of[a]` = the state arc a of tokens_target leads to
  auto* out = new wfl_graph();
  out->start.assign(1 + M, 0), out->accept.assign(1 + M, 0);
  out->start[0] = 1, out->accept[0] = tt->accept[u0];
  for (int u = 0; u < M; ++u) out->accept[1 + u] = tt->accept[u];
  std::vector<int32_t> tok_of(A), in_ptr(M + 1, 0), in_order(A);
  for (int64_t a = 0; a < A; ++a) ++in_ptr[tt->dst[a] + 1];
  for (int v = 0; v < M; ++v) in_ptr[v + 1] += in_ptr[v];
  {
    std::vector<int32_t> fill(in_ptr.begin(), in_ptr.end() - 1);
    for (int64_t a = 0; a < A; ++a) in_order[fill[tt->dst[a]]++] = (int32_t)a;
  }
  std::vector<int32_t> tok_label, tok_node;  // per token state
  for (int v = 0; v < M; ++v)
    for (int i = in_ptr[v]; i < in_ptr[v + 1]; ++i) {  // (a node has a handful of in-arcs: quadratic is fine)
      const int32_t a = in_order[i];
      int32_t id = -1;
      for (int j = in_ptr[v]; j < i; ++j)
        if (tt->il[in_order[j]] == tt->il[a]) {
          id = tok_of[in_order[j]];
          break;
        }
      if (id < 0) {
        id = out->num_nodes();
        out->start.push_back(0), out->accept.push_back(tt->accept[v]);
        tok_label.push_back(tt->il[a]), tok_node.push_back(v);
      }
      tok_of[a] = id;
    }
  {  // (an upper bound of the arc count, so that no vector grows arc by arc)
    size_t bound = 1 + (size_t)(ptr[u0 + 1] - ptr[u0]);
    for (int u = 0; u < M; ++u) bound += 1 + (size_t)(ptr[u + 1] - ptr[u]);
    for (size_t t = 0; t < tok_node.size(); ++t) bound += 2 + (size_t)(ptr[tok_node[t] + 1] - ptr[tok_node[t]]);
    for (auto* v : {&out->src, &out->dst, &out->il, &out->ol}) v->reserve(bound);
    out->w.reserve(bound);
  }
  auto arc = [&](int32_t s, int32_t d, int32_t lab, float w) {
    out->src.push_back(s), out->dst.push_back(d), out->il.push_back(lab), out->ol.push_back(lab), out->w.push_back(w);
  };
  auto leave = [&](int32_t s, int u, int32_t not_token) {  // blank, then every token arc of tokens_target out of u
    arc(s, 1 + u, N, 0.f);
    for (int i = ptr[u]; i < ptr[u + 1]; ++i) {
      const int32_t a = order[i];
      if (tt->il[a] != not_token) arc(s, tok_of[a], tt->il[a], tt->w[a]);
    }
  };
  leave(0, u0, -1);
  for (int u = 0; u < M; ++u) leave(1 + u, u, -1);
  for (size_t t = 0; t < tok_label.size(); ++t) {
    const int32_t s = 1 + M + (int32_t)t;
    arc(s, s, tok_label[t], 0.f);  // one more frame of the token
    leave(s, tok_node[t], tok_label[t]);
  }
  return out;
}

}  // namespace wfl

const wfl::Adjacency& wfl_graph::out_sorted(bool by_olabel) const {
  std::lock_guard<std::mutex> lock(mu);
  if (by_olabel) {
    if (!out_by_ol_ok) {
      wfl::build_sorted(*this, true, out_by_ol);
      out_by_ol_ok = true;
    }
    return out_by_ol;
  }
  if (!out_by_il_ok) {
    wfl::build_sorted(*this, false, out_by_il);
    out_by_il_ok = true;
  }
  return out_by_il;
}

using wfl::set_error;

extern "C" {

const char* wfl_last_error(void) { return wfl::g_err.c_str(); }
int wfl_version(void) { return 1; }
void wfl_free(void* p) { free(p); }

wfl_graph* wfl_graph_new(void) { return new wfl_graph(); }
void wfl_graph_free(wfl_graph* g) { delete g; }

wfl_graph* wfl_graph_clone(const wfl_graph* g) {
  if (!g) return nullptr;
  auto* o = new wfl_graph();
  o->start = g->start, o->accept = g->accept;
  o->src = g->src, o->dst = g->dst, o->il = g->il, o->ol = g->ol, o->w = g->w;
  o->sort_mode = g->sort_mode;
  return o;
}

int wfl_graph_add_node(wfl_graph* g, int start, int accept) {
  g->start.push_back(start != 0);
  g->accept.push_back(accept != 0);
  g->invalidate();
  return g->num_nodes() - 1;
}

int wfl_graph_add_arc(wfl_graph* g, int src, int dst, int ilabel, int olabel, float weight) {
  const int n = g->num_nodes();
  if (src < 0 || src >= n || dst < 0 || dst >= n) {
    set_error("add_arc: node out of range (src=%d dst=%d nodes=%d)", src, dst, n);
    return -1;
  }
  g->src.push_back(src), g->dst.push_back(dst), g->il.push_back(ilabel), g->ol.push_back(olabel);
  g->w.push_back(weight);
  g->invalidate();
  g->sort_mode = -1;
  return (int)(g->num_arcs() - 1);
}

int wfl_graph_add_nodes(wfl_graph* g, int n, const uint8_t* start, const uint8_t* accept) {
  for (int i = 0; i < n; ++i) {
    g->start.push_back(start ? start[i] != 0 : 0);
    g->accept.push_back(accept ? accept[i] != 0 : 0);
  }
  g->invalidate();
  return WFL_OK;
}

int wfl_graph_add_arcs(wfl_graph* g, int64_t n, const int32_t* src, const int32_t* dst, const int32_t* ilabel,
                       const int32_t* olabel, const float* weight) {
  const int nn = g->num_nodes();
  for (int64_t a = 0; a < n; ++a)
    if (src[a] < 0 || src[a] >= nn || dst[a] < 0 || dst[a] >= nn) {
      set_error("add_arcs: node out of range at arc %lld", (long long)a);
      return WFL_ERR_INVALID;
    }
  g->src.insert(g->src.end(), src, src + n);
  g->dst.insert(g->dst.end(), dst, dst + n);
  g->il.insert(g->il.end(), ilabel, ilabel + n);
  if (olabel)
    g->ol.insert(g->ol.end(), olabel, olabel + n);
  else
    g->ol.insert(g->ol.end(), ilabel, ilabel + n);
  if (weight)
    g->w.insert(g->w.end(), weight, weight + n);
  else
    g->w.insert(g->w.end(), (size_t)n, 0.f);
  g->invalidate();
  g->sort_mode = -1;
  return WFL_OK;
}

int wfl_graph_num_nodes(const wfl_graph* g) { return g->num_nodes(); }
int64_t wfl_graph_num_arcs(const wfl_graph* g) { return g->num_arcs(); }

int wfl_graph_get(const wfl_graph* g, uint8_t* start, uint8_t* accept, int32_t* src, int32_t* dst, int32_t* ilabel,
                  int32_t* olabel, float* weight) {
  const size_t n = g->start.size(), m = g->src.size();
  if (start && n) memcpy(start, g->start.data(), n);
  if (accept && n) memcpy(accept, g->accept.data(), n);
  if (src && m) memcpy(src, g->src.data(), m * 4);
  if (dst && m) memcpy(dst, g->dst.data(), m * 4);
  if (ilabel && m) memcpy(ilabel, g->il.data(), m * 4);
  if (olabel && m) memcpy(olabel, g->ol.data(), m * 4);
  if (weight && m) memcpy(weight, g->w.data(), m * 4);
  return WFL_OK;
}

int wfl_graph_set_weights(wfl_graph* g, const float* w) {
  if (g->num_arcs()) memcpy(g->w.data(), w, g->num_arcs() * sizeof(float));
  return WFL_OK;
}

int wfl_graph_arc_sort(wfl_graph* g, int olabel) {
  g->sort_mode = olabel ? 1 : 0;
  (void)g->out_sorted(olabel != 0);  // build the index compose will use
  return WFL_OK;
}

// ---------------------------------------------------------------------------------------------
// compose
// ---------------------------------------------------------------------------------------------
wfl_graph* wfl_graph_compose(const wfl_graph* g1, const wfl_graph* g2, int32_t** prov_first, int32_t** prov_second) {
  if (!g1 || !g2) {
    set_error("compose: null graph");
    return nullptr;
  }
  const wfl::Adjacency& A1 = g1->out_sorted(true);   // by olabel
  const wfl::Adjacency& A2 = g2->out_sorted(false);  // by ilabel
  const int64_t n2 = g2->num_nodes();
  // (n1, n2) -> composed node: open-addressing table (linear probing, power-of-two capacity); the per-utterance
  // compositions of the Transducer visit a few thousand pairs each and std::unordered_map's node allocations
  // were most of their cost
  std::vector<int64_t> tab_key(1024, -1);
  std::vector<int32_t> tab_val(1024, 0);
  size_t tab_mask = 1023, tab_used = 0;
  std::vector<std::pair<int32_t, int32_t>> pairs;  // node -> (n1, n2)
  pairs.reserve(512);
  struct TArc {
    int32_t s, d, il, ol, a1, a2;
    float w;
  };
  std::vector<TArc> arcs;
  arcs.reserve(1024);
  std::vector<int32_t> queue;  // FIFO: consumed through `qhead`
  queue.reserve(512);
  size_t qhead = 0;
  auto slot_of_key = [&](int64_t key) {
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
    size_t i = (size_t)(h >> 20) & tab_mask;
    while (tab_key[i] != -1 && tab_key[i] != key) i = (i + 1) & tab_mask;
    return i;
  };
  // Dead-end pruning: a pair that is not accepting and has no way forward would only be trimmed again at the end,
  // together with every arc into it.  Most pairs of the per-utterance compositions are of this kind (the reference's
  // lexicon graph is an unshared trie: ~100 word pieces start with the same letter and all but a few die on their
  // second letter), so they are recognised BEFORE a node, a queue entry and an arc are spent on them.  The
  // surviving nodes and arcs keep their relative order: the result is identical to composing first and trimming after.
  auto can_move = [&](int32_t a, int32_t b) -> bool {
    if (g1->accept[a] && g2->accept[b]) return true;
    const int32_t* x = A1.idx.data() + A1.ptr[a];
    const int32_t* xe = A1.idx.data() + A1.ptr[a + 1];
    const int32_t* y = A2.idx.data() + A2.ptr[b];
    const int32_t* ye = A2.idx.data() + A2.ptr[b + 1];
    if (x == xe && y == ye) return false;
    if (x != xe && g1->ol[*x] == WFL_EPSILON) return true;  // (sorted: epsilon = -1 comes first)
    if (y != ye && g2->il[*y] == WFL_EPSILON) return true;
    if (x == xe || y == ye) return false;
    if ((xe - x) * 8 < (ye - y) || (ye - y) * 8 < (xe - x)) {
      const bool small_first = (xe - x) < (ye - y);
      const int32_t *sm = small_first ? x : y, *sme = small_first ? xe : ye;
      const std::vector<int32_t>& skey = small_first ? g1->ol : g2->il;
      const std::vector<int32_t>& lkey = small_first ? g2->il : g1->ol;
      const wfl::Adjacency& LA = small_first ? A2 : A1;
      const int ln = small_first ? b : a;
      for (; sm != sme; ++sm) {
        const int32_t* end;
        if (LA.find(ln, skey[*sm], lkey, end)) return true;
      }
      return false;
    }
    while (x != xe && y != ye) {
      const int32_t lx = g1->ol[*x], ly = g2->il[*y];
      if (lx == ly) return true;
      if (lx < ly)
        ++x;
      else
        ++y;
    }
    return false;
  };
  // returns the node of pair (a, b), creating it on first sight; -1 for a dead end (remembered in the table)
  auto get_node = [&](int32_t a, int32_t b, bool force) -> int32_t {
    const int64_t key = (int64_t)a * n2 + b;
    size_t i = slot_of_key(key);
    if (tab_key[i] == key) return tab_val[i];
    const bool alive = force || can_move(a, b);
    const int32_t id = alive ? (int32_t)pairs.size() : -1;
    tab_key[i] = key, tab_val[i] = id;
    ++tab_used;
    if (alive) {
      pairs.emplace_back(a, b);
      queue.push_back(id);
    }
    if (tab_used * 2 > tab_mask) {  // keep the load factor below 1/2
      std::vector<int64_t> ok;
      std::vector<int32_t> ov;
      ok.swap(tab_key), ov.swap(tab_val);
      tab_mask = tab_mask * 4 + 3;
      tab_key.assign(tab_mask + 1, -1), tab_val.assign(tab_mask + 1, 0);
      for (size_t k = 0; k < ok.size(); ++k)
        if (ok[k] != -1) {
          const size_t j = slot_of_key(ok[k]);
          tab_key[j] = ok[k], tab_val[j] = ov[k];
        }
    }
    return id;
  };
  for (int a = 0; a < g1->num_nodes(); ++a)
    if (g1->start[a])
      for (int b = 0; b < g2->num_nodes(); ++b)
        if (g2->start[b]) get_node(a, b, true);
  while (qhead < queue.size()) {
    const int32_t cur = queue[qhead++];
    const int32_t a = pairs[cur].first, b = pairs[cur].second;
    const int32_t* x = A1.idx.data() + A1.ptr[a];
    const int32_t* xe = A1.idx.data() + A1.ptr[a + 1];
    const int32_t* y = A2.idx.data() + A2.ptr[b];
    const int32_t* ye = A2.idx.data() + A2.ptr[b + 1];
    // epsilon on the first graph's output: advance first alone
    for (; x != xe && g1->ol[*x] == WFL_EPSILON; ++x) {
      const int32_t d = get_node(g1->dst[*x], b, false);
      if (d >= 0) arcs.push_back({cur, d, g1->il[*x], WFL_EPSILON, *x, -1, g1->w[*x]});
    }
    // epsilon on the second graph's input: advance second alone
    const int32_t* y0 = y;
    for (; y != ye && g2->il[*y] == WFL_EPSILON; ++y) {
      const int32_t d = get_node(a, g2->dst[*y], false);
      if (d >= 0) arcs.push_back({cur, d, WFL_EPSILON, g2->ol[*y], -1, *y, g2->w[*y]});
    }
    (void)y0;
    // label matches: both ranges are sorted by the matching label
    const int64_t nx = xe - x, ny = ye - y;
    auto emit = [&](const int32_t* xa, const int32_t* ya) {
      const int32_t d = get_node(g1->dst[*xa], g2->dst[*ya], false);
      if (d >= 0) arcs.push_back({cur, d, g1->il[*xa], g2->ol[*ya], *xa, *ya, g1->w[*xa] + g2->w[*ya]});
    };
    if (nx == 0 || ny == 0) continue;
    if (nx * 8 < ny || ny * 8 < nx) {
      // iterate the small side, binary search the large side
      const bool small_first = nx < ny;
      const int32_t *s = small_first ? x : y, *se = small_first ? xe : ye;
      const int32_t *l = small_first ? y : x, *le = small_first ? ye : xe;
      const std::vector<int32_t>& skey = small_first ? g1->ol : g2->il;
      const std::vector<int32_t>& lkey = small_first ? g2->il : g1->ol;
      const wfl::Adjacency& LA = small_first ? A2 : A1;
      const int ln = small_first ? b : a;
      (void)l;
      for (; s != se; ++s) {
        const int32_t lab = skey[*s];
        const int32_t* end;
        const int32_t* lo = LA.find(ln, lab, lkey, end);
        if (!lo) continue;
        for (; lo != le && lo != end && lkey[*lo] == lab; ++lo) small_first ? emit(s, lo) : emit(lo, s);
      }
    } else {
      while (x != xe && y != ye) {
        const int32_t lx = g1->ol[*x], ly = g2->il[*y];
        if (lx < ly)
          ++x;
        else if (ly < lx)
          ++y;
        else {
          const int32_t* y2 = y;
          for (; y2 != ye && g2->il[*y2] == lx; ++y2) emit(x, y2);
          ++x;
        }
      }
    }
  }
  // trim to co-accessible states
  const int32_t np = (int32_t)pairs.size();
  std::vector<uint8_t> coacc(np, 0);
  {
    std::vector<int64_t> rptr(np + 1, 0);
    for (auto& t : arcs) rptr[t.d + 1]++;
    for (int i = 0; i < np; ++i) rptr[i + 1] += rptr[i];
    std::vector<int32_t> rsrc(arcs.size());
    std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
    for (auto& t : arcs) rsrc[fill[t.d]++] = t.s;
    std::vector<int32_t> stack;
    for (int i = 0; i < np; ++i)
      if (g1->accept[pairs[i].first] && g2->accept[pairs[i].second]) coacc[i] = 1, stack.push_back(i);
    while (!stack.empty()) {
      const int32_t v = stack.back();
      stack.pop_back();
      for (int64_t k = rptr[v]; k < rptr[v + 1]; ++k)
        if (!coacc[rsrc[k]]) coacc[rsrc[k]] = 1, stack.push_back(rsrc[k]);
    }
  }
  auto* out = new wfl_graph();
  std::vector<int32_t> newid(np, -1);
  for (int i = 0; i < np; ++i)
    if (coacc[i]) {
      newid[i] = out->num_nodes();
      out->start.push_back(g1->start[pairs[i].first] && g2->start[pairs[i].second]);
      out->accept.push_back(g1->accept[pairs[i].first] && g2->accept[pairs[i].second]);
    }
  std::vector<int32_t> p1, p2;
  for (auto& t : arcs)
    if (coacc[t.s] && coacc[t.d]) {
      out->src.push_back(newid[t.s]), out->dst.push_back(newid[t.d]);
      out->il.push_back(t.il), out->ol.push_back(t.ol), out->w.push_back(t.w);
      p1.push_back(t.a1), p2.push_back(t.a2);
    }
  auto give = [](int32_t** dstp, const std::vector<int32_t>& v) {
    if (!dstp) return;
    *dstp = (int32_t*)malloc(std::max<size_t>(1, v.size()) * sizeof(int32_t));
    if (!v.empty()) memcpy(*dstp, v.data(), v.size() * sizeof(int32_t));
  };
  give(prov_first, p1);
  give(prov_second, p2);
  return out;
}

// ---------------------------------------------------------------------------------------------
// remove / project
// ---------------------------------------------------------------------------------------------
wfl_graph* wfl_graph_remove(const wfl_graph* g, int ilabel, int olabel, int32_t** prov) {
  const int n = g->num_nodes();
  const int64_t m = g->num_arcs();
  auto match = [&](int64_t a) { return g->il[a] == ilabel && g->ol[a] == olabel; };
  std::vector<uint8_t> keep(g->start.begin(), g->start.end());
  for (int64_t a = 0; a < m; ++a)
    if (!match(a)) keep[g->dst[a]] = 1;
  // insertion-order adjacency (keeps the relative arc order of the input)
  std::vector<int64_t> ptr(n + 1, 0);
  for (int64_t a = 0; a < m; ++a) ptr[g->src[a] + 1]++;
  for (int i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
  std::vector<int32_t> idx(m);
  {
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t a = 0; a < m; ++a) idx[fill[g->src[a]]++] = (int32_t)a;
  }
  auto* out = new wfl_graph();
  std::vector<int32_t> newid(n, -1);
  for (int i = 0; i < n; ++i)
    if (keep[i]) {
      newid[i] = out->num_nodes();
      out->start.push_back(g->start[i]);
      out->accept.push_back(0);
    }
  std::vector<int32_t> pv;
  std::vector<int32_t> seen_stamp(n, -1);
  std::deque<int32_t> queue;
  for (int i = 0; i < n; ++i) {
    if (!keep[i]) continue;
    queue.clear();
    queue.push_back(i);
    seen_stamp[i] = i;
    while (!queue.empty()) {
      const int32_t r = queue.front();
      queue.pop_front();
      if (g->accept[r]) out->accept[newid[i]] = 1;
      for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) {
        const int32_t a = idx[k];
        if (match(a)) {
          if (seen_stamp[g->dst[a]] != i) seen_stamp[g->dst[a]] = i, queue.push_back(g->dst[a]);
        } else {
          out->src.push_back(newid[i]), out->dst.push_back(newid[g->dst[a]]);
          out->il.push_back(g->il[a]), out->ol.push_back(g->ol[a]), out->w.push_back(g->w[a]);
          pv.push_back(a);
        }
      }
    }
  }
  if (prov) {
    *prov = (int32_t*)malloc(std::max<size_t>(1, pv.size()) * sizeof(int32_t));
    if (!pv.empty()) memcpy(*prov, pv.data(), pv.size() * sizeof(int32_t));
  }
  return out;
}

wfl_graph* wfl_graph_project(const wfl_graph* g, int output) {
  wfl_graph* o = wfl_graph_clone(g);
  if (output)
    o->il = o->ol;
  else
    o->ol = o->il;
  o->sort_mode = -1;
  return o;
}

// ---------------------------------------------------------------------------------------------
// viterbi_path on a host graph (DAG)
// ---------------------------------------------------------------------------------------------
wfl_graph* wfl_graph_viterbi_path(const wfl_graph* g) {
  const int n = g->num_nodes();
  const int64_t m = g->num_arcs();
  std::vector<int64_t> ptr(n + 1, 0);
  std::vector<int32_t> indeg(n, 0);
  for (int64_t a = 0; a < m; ++a) ptr[g->src[a] + 1]++, indeg[g->dst[a]]++;
  for (int i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
  std::vector<int32_t> idx(m);
  {
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t a = 0; a < m; ++a) idx[fill[g->src[a]]++] = (int32_t)a;
  }
  const float NEG = -std::numeric_limits<float>::infinity();
  std::vector<double> score(n, NEG);
  std::vector<int32_t> nout(n, 0), back(n, -1);
  std::deque<int32_t> queue;
  for (int i = 0; i < n; ++i) {
    if (g->start[i]) score[i] = 0;
    if (!indeg[i]) queue.push_back(i);
  }
  int visited = 0;
  while (!queue.empty()) {
    const int32_t v = queue.front();
    queue.pop_front();
    ++visited;
    for (int64_t k = ptr[v]; k < ptr[v + 1]; ++k) {
      const int32_t a = idx[k], d = g->dst[a];
      float wa = g->w[a];
      if (wa != wa) wa = NEG;  // NaN arc == impossible arc (NaN policy)
      const double s = score[v] + wa;
      const int32_t no = nout[v] + (g->ol[a] != WFL_EPSILON);
      if (s > score[d] || (s == score[d] && s > NEG && no < nout[d])) {
        score[d] = s, nout[d] = no, back[d] = a;
      }
      if (--indeg[d] == 0) queue.push_back(d);
    }
  }
  if (visited != n) {
    set_error("viterbi_path: graph has a cycle");
    return nullptr;
  }
  int best = -1;
  for (int i = 0; i < n; ++i)
    if (g->accept[i] && score[i] > NEG &&
        (best < 0 || score[i] > score[best] || (score[i] == score[best] && nout[i] < nout[best])))
      best = i;
  auto* out = new wfl_graph();
  if (best < 0) return out;
  std::vector<int32_t> path;
  for (int v = best; back[v] >= 0; v = g->src[back[v]]) path.push_back(back[v]);
  std::reverse(path.begin(), path.end());
  out->start.push_back(1), out->accept.push_back(path.empty());
  for (size_t k = 0; k < path.size(); ++k) {
    out->start.push_back(0), out->accept.push_back(k + 1 == path.size());
    const int32_t a = path[k];
    out->src.push_back((int32_t)k), out->dst.push_back((int32_t)k + 1);
    out->il.push_back(g->il[a]), out->ol.push_back(g->ol[a]), out->w.push_back(g->w[a]);
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// equal / isomorphic
// ---------------------------------------------------------------------------------------------
wfl_graph* wfl_graph_token_alignments(const wfl_graph* tokens, const wfl_graph* tokens_target) {
  if (!tokens || !tokens_target) return nullptr;
  return wfl::token_alignments(tokens, tokens_target);
}

int wfl_graph_equal(const wfl_graph* a, const wfl_graph* b) {
  if (a->num_nodes() != b->num_nodes() || a->num_arcs() != b->num_arcs()) return 0;
  if (a->start != b->start || a->accept != b->accept) return 0;
  auto key = [](const wfl_graph* g) {
    std::vector<std::tuple<int, int, int, int, float>> v;
    for (int64_t k = 0; k < g->num_arcs(); ++k) v.emplace_back(g->src[k], g->dst[k], g->il[k], g->ol[k], g->w[k]);
    std::sort(v.begin(), v.end());
    return v;
  };
  return key(a) == key(b);
}

int wfl_graph_isomorphic(const wfl_graph* a, const wfl_graph* b) {
  const int n = a->num_nodes();
  if (n != b->num_nodes() || a->num_arcs() != b->num_arcs()) return 0;
  using Sig = std::tuple<int, int, std::vector<std::tuple<int, int, float, int>>, std::vector<std::tuple<int, int, float>>>;
  auto sigs = [](const wfl_graph* g) {
    std::vector<Sig> s(g->num_nodes());
    for (int i = 0; i < g->num_nodes(); ++i) std::get<0>(s[i]) = g->start[i], std::get<1>(s[i]) = g->accept[i];
    for (int64_t k = 0; k < g->num_arcs(); ++k) {
      std::get<2>(s[g->src[k]]).emplace_back(g->il[k], g->ol[k], g->w[k], g->src[k] == g->dst[k]);
      std::get<3>(s[g->dst[k]]).emplace_back(g->il[k], g->ol[k], g->w[k]);
    }
    for (auto& x : s) std::sort(std::get<2>(x).begin(), std::get<2>(x).end()), std::sort(std::get<3>(x).begin(), std::get<3>(x).end());
    return s;
  };
  auto sa = sigs(a), sb = sigs(b);
  {
    auto x = sa, y = sb;
    std::sort(x.begin(), x.end()), std::sort(y.begin(), y.end());
    if (x != y) return 0;
  }
  auto adj = [](const wfl_graph* g) {
    std::vector<std::vector<int32_t>> out(g->num_nodes()), in(g->num_nodes());
    for (int64_t k = 0; k < g->num_arcs(); ++k) out[g->src[k]].push_back((int32_t)k), in[g->dst[k]].push_back((int32_t)k);
    return std::make_pair(out, in);
  };
  auto [aout, ain] = adj(a);
  auto [bout, bin] = adj(b);
  std::vector<int32_t> map(n, -1), inv(n, -1);
  using Key = std::tuple<int, int, int, float>;
  auto consistent = [&](int u, int v) {
    for (int dir = 0; dir < 2; ++dir) {
      std::vector<Key> l, r;
      for (int32_t k : (dir ? ain : aout)[u]) {
        const int o = dir ? a->src[k] : a->dst[k];
        const int mo = (o == u) ? v : map[o];
        if (mo >= 0) l.emplace_back(mo, a->il[k], a->ol[k], a->w[k]);
      }
      for (int32_t k : (dir ? bin : bout)[v]) {
        const int o = dir ? b->src[k] : b->dst[k];
        if (o == v || inv[o] >= 0) r.emplace_back(o, b->il[k], b->ol[k], b->w[k]);
      }
      std::sort(l.begin(), l.end()), std::sort(r.begin(), r.end());
      if (l != r) return false;
    }
    return true;
  };
  std::function<bool(int)> solve = [&](int u) {
    if (u == n) return true;
    for (int v = 0; v < n; ++v) {
      if (inv[v] >= 0 || sa[u] != sb[v] || !consistent(u, v)) continue;
      map[u] = v, inv[v] = u;
      if (solve(u + 1)) return true;
      map[u] = -1, inv[v] = -1;
    }
    return false;
  };
  return solve(0) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// text I/O (format pinned by tests/trans_backoff_test.txt)
// ---------------------------------------------------------------------------------------------
wfl_graph* wfl_graph_loadtxt(const char* path) {
  std::ifstream in(path);
  if (!in) {
    set_error("loadtxt: cannot open %s", path);
    return nullptr;
  }
  std::vector<std::string> lines;
  for (std::string ln; std::getline(in, ln);) lines.push_back(ln);
  while (!lines.empty() && lines.back().find_first_not_of(" \t\r\n") == std::string::npos) lines.pop_back();
  if (lines.size() < 2) {
    set_error("loadtxt: %s needs a start line and an accept line", path);
    return nullptr;
  }
  auto ints = [](const std::string& s) {
    std::vector<int> v;
    std::istringstream ss(s);
    for (int x; ss >> x;) v.push_back(x);
    return v;
  };
  const auto starts = ints(lines[0]), accepts = ints(lines[1]);
  struct A {
    int s, d, il, ol;
    float w;
  };
  std::vector<A> arcs;
  int maxn = -1;
  for (int v : starts) maxn = std::max(maxn, v);
  for (int v : accepts) maxn = std::max(maxn, v);
  for (size_t k = 2; k < lines.size(); ++k) {
    std::istringstream ss(lines[k]);
    A a{0, 0, 0, 0, 0.f};
    if (!(ss >> a.s >> a.d >> a.il)) continue;
    if (!(ss >> a.ol)) a.ol = a.il;
    std::string wtok;  // strtof, not operator>>: "-inf" / "nan" are legal weights (hard constraints)
    a.w = (ss >> wtok) ? std::strtof(wtok.c_str(), nullptr) : 0.f;
    arcs.push_back(a);
    maxn = std::max(maxn, std::max(a.s, a.d));
  }
  auto* g = new wfl_graph();
  g->start.assign(maxn + 1, 0), g->accept.assign(maxn + 1, 0);
  for (int v : starts) g->start[v] = 1;
  for (int v : accepts) g->accept[v] = 1;
  for (auto& a : arcs) g->src.push_back(a.s), g->dst.push_back(a.d), g->il.push_back(a.il), g->ol.push_back(a.ol), g->w.push_back(a.w);
  return g;
}

int wfl_graph_savetxt(const wfl_graph* g, const char* path) {
  std::ofstream out(path);
  if (!out) {
    set_error("savetxt: cannot open %s", path);
    return WFL_ERR_INVALID;
  }
  bool first = true;
  for (int i = 0; i < g->num_nodes(); ++i)
    if (g->start[i]) out << (first ? "" : " ") << i, first = false;
  out << "\n";
  first = true;
  for (int i = 0; i < g->num_nodes(); ++i)
    if (g->accept[i]) out << (first ? "" : " ") << i, first = false;
  out << "\n";
  out.precision(9);
  for (int64_t k = 0; k < g->num_arcs(); ++k)
    out << g->src[k] << " " << g->dst[k] << " " << g->il[k] << " " << g->ol[k] << " " << g->w[k] << "\n";
  return WFL_OK;
}

// ---- gtn.save / gtn.load (utils.py:261 reads config["transitions"] with gtn.load; build_transitions.py:221
// writes it with gtn.save).  gtn is not vendored, so the layout is restated from gtn's published utils.cpp and is
// UNPINNED: four int32 counts (num_nodes + the numbers of start nodes, accept nodes and arcs), the start ids, the
// accept ids, then per arc {src, dst, ilabel, olabel : int32, weight : float32}, little endian.  The order of the
// three trailing counts is not something this tree can check against gtn, so the reader accepts the two
// plausible orders and picks the one that is consistent with the file size and with every id being in range;
// a file that fits neither is rejected loudly instead of being mis-parsed.
static wfl_graph* parse_binary(const std::vector<char>& buf, const char* path) {
  const int64_t size = (int64_t)buf.size();
  if (size < 16) {
    wfl::set_error("load: %s is too short for a binary graph header", path);
    return nullptr;
  }
  int32_t h[4];
  std::memcpy(h, buf.data(), 16);
  const int64_t n = h[0];
  // (num_start, num_accept, num_arcs) candidates: {nodes, start, accept, arcs} and {nodes, arcs, start, accept}
  const int64_t cand[2][3] = {{h[1], h[2], h[3]}, {h[2], h[3], h[1]}};
  for (int c = 0; c < 2; ++c) {
    const int64_t ns = cand[c][0], na = cand[c][1], m = cand[c][2];
    if (n < 0 || ns < 0 || na < 0 || m < 0 || ns > n || na > n) continue;
    if (16 + 4 * (ns + na) + 20 * m != size) continue;
    const char* p = buf.data() + 16;
    std::vector<int32_t> st(ns), ac(na);
    std::memcpy(st.data(), p, 4 * ns), p += 4 * ns;
    std::memcpy(ac.data(), p, 4 * na), p += 4 * na;
    bool ok = true;
    for (int32_t v : st) ok &= v >= 0 && v < n;
    for (int32_t v : ac) ok &= v >= 0 && v < n;
    if (!ok) continue;
    auto* g = new wfl_graph();
    g->start.assign(n, 0), g->accept.assign(n, 0);
    for (int32_t v : st) g->start[v] = 1;
    for (int32_t v : ac) g->accept[v] = 1;
    g->src.resize(m), g->dst.resize(m), g->il.resize(m), g->ol.resize(m), g->w.resize(m);
    for (int64_t k = 0; k < m && ok; ++k, p += 20) {
      int32_t a[4];
      std::memcpy(a, p, 16);
      std::memcpy(&g->w[k], p + 16, 4);
      g->src[k] = a[0], g->dst[k] = a[1], g->il[k] = a[2], g->ol[k] = a[3];
      ok &= a[0] >= 0 && a[0] < n && a[1] >= 0 && a[1] < n && a[2] >= -1 && a[3] >= -1;
    }
    if (ok) return g;
    delete g;
  }
  wfl::set_error("load: %s is neither gtn text nor a consistent gtn binary graph (size %lld, header %d %d %d %d); "
                 "re-export it with gtn.savetxt", path, (long long)size, h[0], h[1], h[2], h[3]);
  return nullptr;
}

wfl_graph* wfl_graph_load(const char* path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) {
    wfl::set_error("load: cannot open %s", path);
    return nullptr;
  }
  std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  // text files hold digits, signs, dots, exponents and white space only
  bool text = !buf.empty();
  for (char ch : buf)
    if (!(std::isdigit((unsigned char)ch) || std::isspace((unsigned char)ch) || (ch != 0 && std::strchr("+-.eEinfa", ch)))) {
      text = false;
      break;
    }
  if (text) return wfl_graph_loadtxt(path);
  return parse_binary(buf, path);
}

int wfl_graph_save(const wfl_graph* g, const char* path) {
  std::ofstream out(path, std::ios::binary);
  if (!out) {
    wfl::set_error("save: cannot open %s", path);
    return WFL_ERR_INVALID;
  }
  std::vector<int32_t> st, ac;
  for (int i = 0; i < g->num_nodes(); ++i) {
    if (g->start[i]) st.push_back(i);
    if (g->accept[i]) ac.push_back(i);
  }
  const int32_t h[4] = {(int32_t)g->num_nodes(), (int32_t)st.size(), (int32_t)ac.size(), (int32_t)g->num_arcs()};
  out.write((const char*)h, 16);
  out.write((const char*)st.data(), 4 * st.size());
  out.write((const char*)ac.data(), 4 * ac.size());
  for (int64_t k = 0; k < g->num_arcs(); ++k) {
    const int32_t a[4] = {g->src[k], g->dst[k], g->il[k], g->ol[k]};
    out.write((const char*)a, 16);
    out.write((const char*)&g->w[k], 4);
  }
  return out ? WFL_OK : WFL_ERR_RUNTIME;
}

}  // extern "C"
