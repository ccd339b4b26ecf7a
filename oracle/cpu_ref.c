/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called from gtn_applications_amd/.
 * Used by tests/ (checked against the float64 oracle) and by bench.py's `cpu_baseline` leg.
 *
 * Plain-C, float32, graph-faithful CPU restatement of the reference's CTC path
 *     negate(forward_score(intersect(g_emissions, g_criterion)))  +  gtn.backward
 * (criterions/ctc.py:15-29 create_ctc_graph, ctc.py:38-65 forward, ctc.py:72-87 backward), batch
 * threaded over utterances like gtn.parallel_for (ctc.py:65,83).
 *
 * It does what the gtn C++ library does for this call chain (library absent from /root/reference,
 * un-pinned -- requirements.txt:1, README.md:11 -- so this restates its published algorithm):
 *   1. build the per-utterance CTC label graph as an arc list;
 *   2. intersect it with the T-step emissions chain, MATERIALISING the composed lattice: nodes
 *      (t, s), one arc per (frame, label-graph arc), trimmed to nodes that are both accessible and
 *      co-accessible (gtn.intersect semantics);
 *   3. forward_score: one log-add per lattice arc in topological order;
 *   4. backward: reverse topological sweep computing arc posteriors, scattered into the [T, C]
 *      emissions gradient.
 * The only liberty taken: lattice nodes are indexed densely as t*S+s instead of through gtn's hash
 * maps, which makes this baseline FASTER than the real library -- a conservative denominator for
 * any speed-up quoted against it ("kind": "port" in bench.py).
 *
 * Round 2 adds the same graph-faithful treatment for the other two timed workloads:
 *   oracle_asg_cpu      criterions/asg.py:84-185 -- per sample: create_transitions_graph (C + C^2 arcs, rebuilt per
 *                       sample like asg.py:103), create_force_align_graph, intersect(fal, transitions) by generic
 *                       label matching, then forward_score / backward over intersect(., emissions) for the
 *                       numerator and over intersect(emissions, transitions) (T*C^2 arcs) for the denominator;
 *   oracle_lattice_cpu  criterions/transducer.py:283,321-336 -- forward_score(intersect(emissions, alignments)) and
 *                       its backward for caller-supplied epsilon-free alignment acceptors, with the log_softmax of
 *                       transducer.py:186-187 (and its backward) around it.  The per-sample graph algebra that builds
 *                       the acceptor (transducer.py:265-276) is NOT part of this port: one more reason it is a
 *                       conservative (fast) stand-in for the real CPU path.
 * For these two the composed lattice is swept frame by frame without storing per-arc index arrays (node (t, s) =
 * t*S + s, arc (t, a)): the same arithmetic as gtn's sweep over the materialised lattice, minus its memory traffic.
 *
 * Build: make -C oracle   (gcc -O3 -pthread -shared).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG (-INFINITY)

static inline float logadd(float a, float b) {
  if (a == NEG) return b;
  if (b == NEG) return a;
  const float m = a > b ? a : b;
  return m + log1pf(expf(-fabsf(a - b)));
}

typedef struct {
  int src, dst, label;
} garc_t;

/* one utterance: returns loss = -log Z and (optionally) accumulates d(loss)/dx * scale into grad */
static float ctc_one(const float* x, int T, int C, const int* y, int L, int blank, float scale, float* grad) {
  const int S = 2 * L + 1;
  /* 1. label graph (ctc.py:15-29) */
  garc_t* ga = (garc_t*)malloc(sizeof(garc_t) * 3 * (size_t)S);
  int A = 0;
  for (int s = 0; s < S; ++s) {
    const int lab = (s & 1) ? y[(s - 1) / 2] : blank;
    ga[A++] = (garc_t){s, s, lab};
    if (s > 0) ga[A++] = (garc_t){s - 1, s, lab};
    if ((s & 1) && s > 1 && lab != y[(s - 1) / 2 - 1]) ga[A++] = (garc_t){s - 2, s, lab};
  }
  /* 2. intersect with the emissions chain: node (t, s) = t*S + s, t = 0..T */
  const size_t n_nodes = (size_t)(T + 1) * S;
  unsigned char* reach = (unsigned char*)calloc(n_nodes, 1);
  unsigned char* coreach = (unsigned char*)calloc(n_nodes, 1);
  reach[0] = 1; /* (0, start state 0) */
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < A; ++a)
      if (reach[(size_t)t * S + ga[a].src]) reach[(size_t)(t + 1) * S + ga[a].dst] = 1;
  coreach[(size_t)T * S + S - 1] = 1;
  if (S >= 2) coreach[(size_t)T * S + S - 2] = 1;
  for (int t = T - 1; t >= 0; --t)
    for (int a = 0; a < A; ++a)
      if (coreach[(size_t)(t + 1) * S + ga[a].dst]) coreach[(size_t)t * S + ga[a].src] = 1;
  /* materialise the surviving arcs (topologically ordered by construction: grouped by frame) */
  size_t cap = (size_t)T * A, n_arcs = 0;
  int* asrc = (int*)malloc(sizeof(int) * (cap ? cap : 1));
  int* adst = (int*)malloc(sizeof(int) * (cap ? cap : 1));
  int* aemi = (int*)malloc(sizeof(int) * (cap ? cap : 1)); /* index into x: t*C + label */
  float* aw = (float*)malloc(sizeof(float) * (cap ? cap : 1));
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < A; ++a) {
      const size_t u = (size_t)t * S + ga[a].src, v = (size_t)(t + 1) * S + ga[a].dst;
      if (reach[u] && coreach[u] && reach[v] && coreach[v]) {
        asrc[n_arcs] = (int)u, adst[n_arcs] = (int)v, aemi[n_arcs] = t * C + ga[a].label;
        float w = x[aemi[n_arcs]];
        aw[n_arcs] = (w != w) ? NEG : w; /* NaN weight == impossible arc */
        ++n_arcs;
      }
    }
  /* 3. forward_score */
  float* score = (float*)malloc(sizeof(float) * n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) score[i] = NEG;
  score[0] = 0.f;
  for (size_t k = 0; k < n_arcs; ++k) score[adst[k]] = logadd(score[adst[k]], score[asrc[k]] + aw[k]);
  float z = NEG;
  z = logadd(z, score[(size_t)T * S + S - 1]);
  if (S >= 2) z = logadd(z, score[(size_t)T * S + S - 2]);
  /* 4. backward: d(logZ)/d(arc) by reverse sweep */
  if (grad && z != NEG) {
    float* ng = (float*)calloc(n_nodes, sizeof(float));
    ng[(size_t)T * S + S - 1] = expf(score[(size_t)T * S + S - 1] - z);
    if (S >= 2) ng[(size_t)T * S + S - 2] = expf(score[(size_t)T * S + S - 2] - z);
    for (size_t k = n_arcs; k-- > 0;) {
      const float g = ng[adst[k]];
      if (g == 0.f || score[adst[k]] == NEG) continue;
      const float ag = expf(score[asrc[k]] + aw[k] - score[adst[k]]) * g;
      ng[asrc[k]] += ag;
      grad[aemi[k]] += -ag * scale; /* loss = -logZ */
    }
    free(ng);
  }
  free(score), free(aw), free(aemi), free(adst), free(asrc), free(coreach), free(reach), free(ga);
  return -z;
}

typedef struct {
  const float* x;
  int B, T, C, blank;
  const int* targets;
  const int64_t* offsets;
  const float* scale;
  float* losses;
  float* grad;
  int next;
  pthread_mutex_t mu;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    const int b = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (b >= j->B) break;
    const size_t off = (size_t)b * j->T * j->C;
    if (j->grad) memset(j->grad + off, 0, sizeof(float) * (size_t)j->T * j->C);
    j->losses[b] = ctc_one(j->x + off, j->T, j->C, j->targets + j->offsets[b], (int)(j->offsets[b + 1] - j->offsets[b]),
                           j->blank, j->scale ? j->scale[b] : 1.f, j->grad ? j->grad + off : NULL);
  }
  return NULL;
}

/* x [B,T,C] host float32; losses [B]; grad [B,T,C] or NULL; scale [B] or NULL multiplies the gradient
 * of utterance b (scale_b / B of ctc.py:81,87).  Returns 0. */
int oracle_ctc_cpu(const float* x, int B, int T, int C, const int* targets, const int64_t* offsets, int blank,
                   const float* scale, int nthreads, float* losses, float* grad) {
  job_t j = {x, B, T, C, blank, targets, offsets, scale, losses, grad, 0, PTHREAD_MUTEX_INITIALIZER};
  if (nthreads < 1) nthreads = 1;
  if (nthreads > B) nthreads = B;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &j);
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th);
  return 0;
}


/* ================================================================================================
 * generic epsilon-free acceptor (x) emissions chain
 * ============================================================================================== */
typedef struct {
  int S, A;          /* nodes, arcs */
  int *src, *dst, *lab;
  float* w;          /* arc weights (may be NULL = 0) */
  unsigned char *start, *accept;
} acc_t;

static void acc_free(acc_t* g) {
  free(g->src), free(g->dst), free(g->lab), free(g->w), free(g->start), free(g->accept);
}

/* forward_score(intersect(emissions, g)) and, if gx / gw are given, the backward pass scaled by `scale`
 * (d score/dx accumulated into gx [T,C], d score/d arc weight into gw [A]).  Returns the score (log Z). */
static float lattice_one(const float* x, int T, int C, const acc_t* g, float scale, float* gx, float* gw) {
  const int S = g->S, A = g->A;
  const size_t n = (size_t)(T + 1) * S;
  unsigned char* reach = (unsigned char*)calloc(n, 1);
  unsigned char* co = (unsigned char*)calloc(n, 1);
  for (int s = 0; s < S; ++s) reach[s] = g->start[s];
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < A; ++a)
      if (reach[(size_t)t * S + g->src[a]]) reach[(size_t)(t + 1) * S + g->dst[a]] = 1;
  for (int s = 0; s < S; ++s) co[(size_t)T * S + s] = g->accept[s];
  for (int t = T - 1; t >= 0; --t)
    for (int a = 0; a < A; ++a)
      if (co[(size_t)(t + 1) * S + g->dst[a]]) co[(size_t)t * S + g->src[a]] = 1;
  float* score = (float*)malloc(sizeof(float) * n);
  for (size_t i = 0; i < n; ++i) score[i] = NEG;
  for (int s = 0; s < S; ++s)
    if (g->start[s]) score[s] = 0.f;
  for (int t = 0; t < T; ++t) {
    const float* xt = x + (size_t)t * C;
    for (int a = 0; a < A; ++a) {
      const size_t u = (size_t)t * S + g->src[a], v = (size_t)(t + 1) * S + g->dst[a];
      if (!(reach[u] && co[u] && co[v])) continue; /* trimmed away by gtn.intersect */
      float w = xt[g->lab[a]] + (g->w ? g->w[a] : 0.f);
      if (w != w) w = NEG;
      score[v] = logadd(score[v], score[u] + w);
    }
  }
  float z = NEG;
  for (int s = 0; s < S; ++s)
    if (g->accept[s]) z = logadd(z, score[(size_t)T * S + s]);
  if ((gx || gw) && z != NEG) {
    float* ng = (float*)calloc(n, sizeof(float));
    for (int s = 0; s < S; ++s)
      if (g->accept[s] && score[(size_t)T * S + s] != NEG) ng[(size_t)T * S + s] = expf(score[(size_t)T * S + s] - z);
    for (int t = T - 1; t >= 0; --t) {
      const float* xt = x + (size_t)t * C;
      for (int a = A - 1; a >= 0; --a) {
        const size_t u = (size_t)t * S + g->src[a], v = (size_t)(t + 1) * S + g->dst[a];
        if (!(reach[u] && co[u] && co[v])) continue;
        const float gv = ng[v];
        if (gv == 0.f || score[v] == NEG) continue;
        float w = xt[g->lab[a]] + (g->w ? g->w[a] : 0.f);
        if (w != w) w = NEG;
        const float ag = expf(score[u] + w - score[v]) * gv;
        ng[u] += ag;
        if (gx) gx[(size_t)t * C + g->lab[a]] += ag * scale;
        if (gw) gw[a] += ag * scale;
      }
    }
    free(ng);
  }
  free(score), free(co), free(reach);
  return z;
}

/* gtn.intersect of two small epsilon-free acceptors by label matching (reachable pairs only);
 * `from2[k]` = arc of g2 behind result arc k (its weight is the one that is learned: asg.py:62-66) */
static void intersect_small(const acc_t* g1, const acc_t* g2, acc_t* out, int** from2) {
  const int S1 = g1->S, S2 = g2->S;
  int* id = (int*)malloc(sizeof(int) * (size_t)S1 * S2);
  for (int i = 0; i < S1 * S2; ++i) id[i] = -1;
  int* queue = (int*)malloc(sizeof(int) * (size_t)S1 * S2);
  int qh = 0, qt = 0, cap = 16, A = 0;
  int *src = (int*)malloc(sizeof(int) * cap), *dst = (int*)malloc(sizeof(int) * cap), *lab = (int*)malloc(sizeof(int) * cap);
  int* f2 = (int*)malloc(sizeof(int) * cap);
  float* w = (float*)malloc(sizeof(float) * cap);
  for (int a = 0; a < S1; ++a)
    for (int b = 0; b < S2; ++b)
      if (g1->start[a] && g2->start[b]) id[a * S2 + b] = qt, queue[qt++] = a * S2 + b;
  while (qh < qt) {
    const int p = queue[qh++], n1 = p / S2, n2 = p % S2;
    for (int a1 = 0; a1 < g1->A; ++a1) {
      if (g1->src[a1] != n1) continue;
      for (int a2 = 0; a2 < g2->A; ++a2) {
        if (g2->src[a2] != n2 || g2->lab[a2] != g1->lab[a1]) continue;
        const int q = g1->dst[a1] * S2 + g2->dst[a2];
        if (id[q] < 0) id[q] = qt, queue[qt++] = q;
        if (A == cap) {
          cap *= 2;
          src = (int*)realloc(src, sizeof(int) * cap), dst = (int*)realloc(dst, sizeof(int) * cap);
          lab = (int*)realloc(lab, sizeof(int) * cap), f2 = (int*)realloc(f2, sizeof(int) * cap);
          w = (float*)realloc(w, sizeof(float) * cap);
        }
        src[A] = id[p], dst[A] = id[q], lab[A] = g1->lab[a1], f2[A] = a2;
        w[A] = (g1->w ? g1->w[a1] : 0.f) + (g2->w ? g2->w[a2] : 0.f);
        ++A;
      }
    }
  }
  out->S = qt, out->A = A, out->src = src, out->dst = dst, out->lab = lab, out->w = w;
  out->start = (unsigned char*)calloc(qt ? qt : 1, 1), out->accept = (unsigned char*)calloc(qt ? qt : 1, 1);
  for (int i = 0; i < qt; ++i) {
    const int n1 = queue[i] / S2, n2 = queue[i] % S2;
    out->start[i] = g1->start[n1] && g2->start[n2];
    out->accept[i] = g1->accept[n1] && g2->accept[n2];
  }
  *from2 = f2;
  free(queue), free(id);
}

/* ---- ASG ------------------------------------------------------------------------------------------ */
typedef struct {
  const float *x, *W;
  int B, T, C;
  const int* targets;
  const int64_t* offsets;
  const float* scale; /* scale_b / B */
  float *losses, *gx, *gW; /* gW: [nthreads][(C+1)*C] partial sums */
  int next, nthreads;
  pthread_mutex_t mu;
} asg_job_t;

static void asg_one(asg_job_t* j, int b, float* gW) {
  const int C = j->C, T = j->T;
  const int* y = j->targets + j->offsets[b];
  const int L = (int)(j->offsets[b + 1] - j->offsets[b]);
  /* create_transitions_graph (asg.py:54-69), rebuilt per sample as the reference does (asg.py:103) */
  acc_t tr;
  tr.S = C + 1, tr.A = C + C * C;
  tr.src = (int*)malloc(sizeof(int) * tr.A), tr.dst = (int*)malloc(sizeof(int) * tr.A);
  tr.lab = (int*)malloc(sizeof(int) * tr.A), tr.w = (float*)malloc(sizeof(float) * tr.A);
  tr.start = (unsigned char*)calloc(tr.S, 1), tr.accept = (unsigned char*)calloc(tr.S, 1);
  tr.start[0] = 1;
  int a = 0;
  for (int i = 0; i < C; ++i) tr.accept[i + 1] = 1, tr.src[a] = 0, tr.dst[a] = i + 1, tr.lab[a] = i, tr.w[a] = j->W[a], ++a;
  for (int i = 0; i < C; ++i)
    for (int k = 0; k < C; ++k) tr.src[a] = k + 1, tr.dst[a] = i + 1, tr.lab[a] = i, tr.w[a] = j->W[a], ++a;
  /* create_force_align_graph (asg.py:72-81) */
  acc_t fal;
  fal.S = L + 1, fal.A = 2 * L;
  fal.src = (int*)malloc(sizeof(int) * (fal.A + 1)), fal.dst = (int*)malloc(sizeof(int) * (fal.A + 1));
  fal.lab = (int*)malloc(sizeof(int) * (fal.A + 1)), fal.w = NULL;
  fal.start = (unsigned char*)calloc(fal.S, 1), fal.accept = (unsigned char*)calloc(fal.S, 1);
  fal.start[0] = 1, fal.accept[L] = 1;
  for (int l = 1; l <= L; ++l) {
    fal.src[2 * l - 2] = l - 1, fal.dst[2 * l - 2] = l, fal.lab[2 * l - 2] = y[l - 1];
    fal.src[2 * l - 1] = l, fal.dst[2 * l - 1] = l, fal.lab[2 * l - 1] = y[l - 1];
  }
  acc_t num;
  int* from_tr;
  intersect_small(&fal, &tr, &num, &from_tr);
  const size_t off = (size_t)b * T * C;
  float* gx = j->gx ? j->gx + off : NULL;
  const float sc = j->scale[b];
  float* gnum = (float*)calloc(num.A ? num.A : 1, sizeof(float));
  float* gtr = gW ? (float*)calloc(tr.A, sizeof(float)) : NULL;
  const float fal_score = lattice_one(j->x + off, T, C, &num, -sc, gx, gW ? gnum : NULL);
  const float fcc_score = lattice_one(j->x + off, T, C, &tr, sc, gx, gtr);
  j->losses[b] = fcc_score - fal_score;
  if (gW) {
    for (int k = 0; k < tr.A; ++k) gW[k] += gtr[k]; /* arc id == row-major index into W */
    for (int k = 0; k < num.A; ++k) gW[from_tr[k]] += gnum[k];
    free(gtr);
  }
  free(gnum), free(from_tr);
  acc_free(&num), acc_free(&fal), acc_free(&tr);
}

typedef struct {
  asg_job_t* job;
  int tid;
} asg_arg_t;

static void* asg_worker(void* arg) {
  asg_arg_t* a = (asg_arg_t*)arg;
  asg_job_t* j = a->job;
  float* gW = j->gW ? j->gW + (size_t)a->tid * (j->C + 1) * j->C : NULL;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    const int b = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (b >= j->B) break;
    if (j->gx) memset(j->gx + (size_t)b * j->T * j->C, 0, sizeof(float) * (size_t)j->T * j->C);
    asg_one(j, b, gW);
  }
  return NULL;
}

/* x [B,T,C], W [(C+1),C]; scale[b] = scale_b / B; losses [B] unscaled (FCC - FAL); gx [B,T,C] or NULL; gW [(C+1),C]
 * or NULL (summed over the batch). */
int oracle_asg_cpu(const float* x, const float* W, int B, int T, int C, const int* targets, const int64_t* offsets,
                   const float* scale, int nthreads, float* losses, float* gx, float* gW) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > B) nthreads = B;
  const size_t nW = (size_t)(C + 1) * C;
  float* part = gW ? (float*)calloc(nW * nthreads, sizeof(float)) : NULL;
  asg_job_t j = {x, W, B, T, C, targets, offsets, scale, losses, gx, part, 0, nthreads, PTHREAD_MUTEX_INITIALIZER};
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  asg_arg_t* args = (asg_arg_t*)malloc(sizeof(asg_arg_t) * (size_t)nthreads);
  for (int i = 0; i < nthreads; ++i) args[i].job = &j, args[i].tid = i, pthread_create(&th[i], NULL, asg_worker, &args[i]);
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  if (gW) {
    memset(gW, 0, sizeof(float) * nW);
    for (int i = 0; i < nthreads; ++i)
      for (size_t k = 0; k < nW; ++k) gW[k] += part[(size_t)i * nW + k];
    free(part);
  }
  free(args), free(th);
  return 0;
}

/* ---- caller-supplied acceptors (Transducer numerator) --------------------------------------------------- */
typedef struct {
  const float* x;
  int B, T, C, log_softmax;
  const int64_t *node_off, *arc_off; /* [B+1] */
  const int *src, *dst, *lab;
  const unsigned char *start, *accept;
  const float* scale;
  float *losses, *gx;
  int next;
  pthread_mutex_t mu;
} lat_job_t;

static void* lat_worker(void* arg) {
  lat_job_t* j = (lat_job_t*)arg;
  const int T = j->T, C = j->C;
  float* lp = (float*)malloc(sizeof(float) * (size_t)T * C);
  for (;;) {
    pthread_mutex_lock(&j->mu);
    const int b = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (b >= j->B) break;
    const size_t off = (size_t)b * T * C;
    const float* xb = j->x + off;
    if (j->log_softmax) { /* transducer.py:186-187 */
      for (int t = 0; t < T; ++t) {
        float m = NEG, s = 0.f;
        for (int c = 0; c < C; ++c) m = xb[(size_t)t * C + c] > m ? xb[(size_t)t * C + c] : m;
        for (int c = 0; c < C; ++c) s += expf(xb[(size_t)t * C + c] - m);
        const float lse = m + logf(s);
        for (int c = 0; c < C; ++c) lp[(size_t)t * C + c] = xb[(size_t)t * C + c] - lse;
      }
      xb = lp;
    }
    acc_t g;
    g.S = (int)(j->node_off[b + 1] - j->node_off[b]), g.A = (int)(j->arc_off[b + 1] - j->arc_off[b]);
    g.src = (int*)(j->src + j->arc_off[b]), g.dst = (int*)(j->dst + j->arc_off[b]), g.lab = (int*)(j->lab + j->arc_off[b]);
    g.w = NULL, g.start = (unsigned char*)(j->start + j->node_off[b]), g.accept = (unsigned char*)(j->accept + j->node_off[b]);
    float* gx = j->gx ? j->gx + off : NULL;
    if (gx) memset(gx, 0, sizeof(float) * (size_t)T * C);
    j->losses[b] = -lattice_one(xb, T, C, &g, -j->scale[b], gx, NULL);
    if (gx && j->log_softmax) /* backward of log_softmax: g - softmax * sum(g) per frame */
      for (int t = 0; t < T; ++t) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += gx[(size_t)t * C + c];
        for (int c = 0; c < C; ++c) gx[(size_t)t * C + c] -= expf(lp[(size_t)t * C + c]) * s;
      }
  }
  free(lp);
  return NULL;
}

int oracle_lattice_cpu(const float* x, int B, int T, int C, int log_softmax, const int64_t* node_off,
                       const int64_t* arc_off, const int* src, const int* dst, const int* lab,
                       const unsigned char* start, const unsigned char* accept, const float* scale, int nthreads,
                       float* losses, float* gx) {
  lat_job_t j = {x, B, T, C, log_softmax, node_off, arc_off, src, dst, lab, start, accept, scale, losses, gx, 0,
                 PTHREAD_MUTEX_INITIALIZER};
  if (nthreads < 1) nthreads = 1;
  if (nthreads > B) nthreads = B;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, lat_worker, &j);
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th);
  return 0;
}
