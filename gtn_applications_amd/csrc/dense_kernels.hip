// Dense (fully connected) transition engine for gfx950: the ASG denominator
//   forward_score(intersect(emissions, transitions))            criterions/asg.py:114
// its gradient (asg.py:158-168) and the ASG Viterbi decode (asg.py:217-226), for the transitions
// graph of asg.py:54-69:  W[0,i] = start->i,  W[1+i, j] = score(prev j -> cur i).
//
//   alpha_0[i] = x[0,i] + W[0,i] ;  alpha_t[i] = x[t,i] + LSE_j (alpha_{t-1}[j] + W[1+i,j])
//   beta_{T-1}[j] = 0            ;  beta_t[j]  = LSE_i (W[1+i,j] + x[t+1,i] + beta_{t+1}[i])
//
// One workgroup per (utterance, direction); W lives in LDS with an odd leading dimension (bank-
// conflict-free both row- and column-wise), the C*C log-add terms of a frame are split over all
// 256 lanes (R = 256/C partial reductions per state, merged through LDS).  The C*C-per-frame
// transition posteriors of the gradient are accumulated in registers (pairs strided over the
// workgroup, W held in registers too) and written as per-workgroup partial sums that a second
// kernel reduces: deterministic, no atomics.  This path is VALU/transcendental-bound (B*T*C^2
// exp per sweep), not HBM-bound; see DESIGN.md.
//
// The log semiring normally does not run those kernels at all: section "probability-domain sweeps"
// below turns the per-frame LSE over C*C terms into a C x C matrix-vector product in fp32 FMAs
// (transition matrix exponentiated once, alpha/beta kept as scaled probabilities with an exact
// power-of-two renormalisation per frame) and checks, per utterance, that nothing left the fp32
// range; utterances that fail the check are recomputed by the log-domain kernels above.
#include <type_traits>

#include "device_common.h"

namespace wfl {

struct DenseLds {
  float* W;    // [(C+1) * ldw]
  void* a0;    // [C] state vector: double in the log semiring, float in the tropical one (DenseVal)
  void* a1;    // [C]
  float* x0;   // [C]
  float* x1;   // [C]
  void* pm;    // [R*C] partial max (DenseVal)
  float* ps;   // [R*C] partial sum (or arg for tropical)
  float* red;  // [2][32] (reductions; the frames' maxima, double-buffered)
};
// State values of the generic dense sweep.  Log semiring: DOUBLE in LDS -- the scores reach thousands at T = 1000, where a
// float resolves 2.4e-4, and that resolution was paid at every add of the recursion and again in alpha + beta - log Z
// (measured 1.0 .. 2.5e-4 on the posteriors).  What goes to HBM stays float: the frame's vector RELATIVE to the largest
// state of the frame before (a double per frame in the workspace, DenseWs::M) -- values of a few tens at most, and the
// gradient kernel adds the two frame references and log Z (DenseWs::z2, a double) in double.  Tropical: float throughout
// (sums and comparisons of the caller's floats: ties stay ties), raw scores.
template <int SR>
struct DenseVal {
  using type = float;
};
template <>
struct DenseVal<WFL_SEMIRING_LOG> {
  using type = double;
};

__device__ __forceinline__ float blk_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float blk_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}

// DIR 0: alpha sweep, DIR 1: beta sweep.  SR tropical only for DIR 0 (writes back-pointers).
template <int SR, int DIR>
__device__ void dense_chain(const DenseLds& L, const float* __restrict__ x, int b, int T, int C, int ldw,
                            float* __restrict__ out, int32_t* __restrict__ bptr, float* __restrict__ logz,
                            double* __restrict__ Mref, double* __restrict__ z64) {
  using VT = typename DenseVal<SR>::type;
  constexpr bool kLog = SR == WFL_SEMIRING_LOG;
  VT* const a0 = reinterpret_cast<VT*>(L.a0);
  VT* const a1 = reinterpret_cast<VT*>(L.a1);
  VT* const pmv = reinterpret_cast<VT*>(L.pm);
  const int tid = threadIdx.x, NT = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6, nwv = (NT + 63) >> 6;
  double ref = 0.0;  // what the frame's stored values are relative to (log semiring): the previous frame's largest state
  const int R = C <= NT ? NT / C : 1;       // partial reductions per state
  const int span = (C + R - 1) / R;         // terms per partial
  const float* xb = x + (int64_t)b * T * C;
  float* ob = out + (int64_t)b * T * C;
  const int t0 = DIR == 0 ? 0 : T - 1;
  VT* cur = (t0 & 1) ? a1 : a0;
  {
    float top = WFL_NEG_INF;
    for (int i = tid; i < C; i += NT) {
      const float v = DIR == 0 ? nan_to_neg(xb[i]) + L.W[i] : 0.f;
      cur[i] = (VT)v;
      ob[(int64_t)t0 * C + i] = v;  // (relative to ref = 0)
      top = fmaxf(top, v);
      if (SR == WFL_SEMIRING_TROPICAL) bptr[(int64_t)b * T * C + i] = -1;
    }
    if (kLog) {
      top = wave_max(top);
      if (lane == 0) L.red[(t0 & 1) * 32 + wv] = top;
      if (tid == 0 && Mref) Mref[t0] = 0.0;
    }
  }
  if (DIR == 1 && T > 0)  // the beta sweep consumes x[t+1]; stage row T-1 for the first step
    for (int i = tid; i < C; i += NT) (((T - 1) & 1) ? L.x1 : L.x0)[i] = nan_to_neg(xb[(int64_t)(T - 1) * C + i]);
  if (DIR == 0 && T > 1)
    for (int i = tid; i < C; i += NT) L.x1[i] = nan_to_neg(xb[(int64_t)C + i]);
  // The emission row of step s + 1 reaches LDS at the end of step s; it is REQUESTED a step earlier still (registers
  // `pre`: requested during step s - 1, written during step s), and the per-frame barriers order LDS traffic only --
  // neither the row's round trip nor the acknowledgement of the frame's own stores (scores, back-pointers) sits on
  // the frame-to-frame path (with __syncthreads and a row requested in the step that writes it, a frame of the
  // max-plus sweep took ~2.6 us at C = 82: 0.65 of the 0.8 ms of a bigram Transducer.viterbi).
  // (one (state, slice) per thread and a slice of at most kSpanRegs terms: its weights stay in registers)
  constexpr int kSpanRegs = 64;
  const bool wregs = C <= NT && span <= kSpanRegs;  // block-uniform
  const int wj0 = (tid / C) * span;
  float wr[kSpanRegs];
#pragma unroll
  for (int jj = 0; jj < kSpanRegs; ++jj) {
    wr[jj] = WFL_NEG_INF;  // (a slice's tail beyond C: never the maximum, exp(-inf) = 0)
    if (wregs && tid < R * C && jj < span && wj0 + jj < C)
      wr[jj] = DIR == 0 ? L.W[(1 + tid % C) * ldw + wj0 + jj] : L.W[(1 + wj0 + jj) * ldw + tid % C];
  }
  auto row_needed_by = [&](int step) { return DIR == 0 ? step : T - step; };  // (= tx of that step)
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  if (T > 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + j * NT;
      if (i < C) pre[j] = xb[(int64_t)row_needed_by(2) * C + i];
    }
  }
  __syncthreads();
  for (int step = 1; step < T; ++step) {
    const int t = DIR == 0 ? step : T - 1 - step;       // slot being produced
    const int tf = DIR == 0 ? t - 1 : t + 1;            // slot read
    const VT* from = (tf & 1) ? a1 : a0;
    VT* to = (t & 1) ? a1 : a0;
    const int tx = DIR == 0 ? t : t + 1;                // emissions row used by this step
    const float* xr = (tx & 1) ? L.x1 : L.x0;
    const int txn = DIR == 0 ? t + 1 : t;               // row the next step needs
    const bool has_next = step + 1 < T;
    float pre2[4] = {0.f, 0.f, 0.f, 0.f};
    if (step + 2 < T) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + j * NT;
        if (i < C) pre2[j] = xb[(int64_t)row_needed_by(step + 2) * C + i];
      }
    }
    // ---- partial reductions
    if (wregs) {
      // the thread's slice of W in registers (wr, loaded once): a frame only reads the previous vector from LDS, the
      // same address across the 82-odd threads of a slice (a broadcast) -- reading W itself every frame, rows a
      // stride of C apart, bank-conflicted a 28-term slice into ~6000 cycles (measured: 2.6 us per frame at C = 82)
      // (the slice's terms at compile-time offsets from ONE base address, NO branch per term -- behind a branch every
      // LDS read was waited for on its own: a0 / a1 / x0 / x1 are padded by kSpanRegs zeros, and a term beyond the
      // slice carries wr = -inf: never the maximum, exp(-inf) = 0.  The trip count is the slice rounded up to 16 / 32 / 64.)
      auto partials = [&](auto bucket) {
        constexpr int S = decltype(bucket)::value;
        const VT* fp = from + wj0;
        const float* xp = xr + wj0;
        VT m = WFL_NEG_INF;
        int am = -1;
#pragma unroll
        for (int jj = 0; jj < S; ++jj) {
          const VT v = DIR == 0 ? fp[jj] + (VT)wr[jj] : (VT)wr[jj] + (VT)xp[jj] + fp[jj];
          if (v > m) m = v, am = jj;
        }
        if (am >= 0) am += wj0;
        float sum = 0.f;
        if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF) {
#pragma unroll
          for (int jj = 0; jj < S; ++jj) {
            const VT v = DIR == 0 ? fp[jj] + (VT)wr[jj] : (VT)wr[jj] + (VT)xp[jj] + fp[jj];
            sum += fast_exp((float)(v - m));
          }
        }
        pmv[tid] = m;
        L.ps[tid] = SR == WFL_SEMIRING_LOG ? sum : __int_as_float(am);
      };
      if (tid < R * C) {
        if (span <= 16)
          partials(std::integral_constant<int, 16>{});
        else if (span <= 32)
          partials(std::integral_constant<int, 32>{});
        else
          partials(std::integral_constant<int, kSpanRegs>{});
      }
    } else
    for (int idx = tid; idx < R * C; idx += NT) {
      const int s = idx % C, r = idx / C;  // s: state being produced, r: which slice of the other index
      const int j0 = r * span, j1 = min(C, j0 + span);
      VT m = WFL_NEG_INF;
      int am = -1;
      for (int j = j0; j < j1; ++j) {
        const VT v = DIR == 0 ? from[j] + (VT)L.W[(1 + s) * ldw + j] : (VT)L.W[(1 + j) * ldw + s] + (VT)xr[j] + from[j];
        if (v > m) m = v, am = j;
      }
      float sum = 0.f;
      if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF)
        for (int j = j0; j < j1; ++j) {
          const VT v = DIR == 0 ? from[j] + (VT)L.W[(1 + s) * ldw + j] : (VT)L.W[(1 + j) * ldw + s] + (VT)xr[j] + from[j];
          sum += fast_exp((float)(v - m));
        }
      pmv[idx] = m;
      L.ps[idx] = SR == WFL_SEMIRING_LOG ? sum : __int_as_float(am);
    }
    lds_barrier();
    // (the next step's emission row goes to LDS, and the row after it moves up, BEFORE this frame's stores are issued:
    // the counter the loads are waited on is shared with the stores, and behind them the wait was a store round trip)
    if (has_next) {
      float* xn = (txn & 1) ? L.x1 : L.x0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + j * NT;
        if (i < C) xn[i] = nan_to_neg(pre[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = pre2[j];
    // ---- merge partials, add the emission (alpha only), publish
    if (kLog) {  // this frame's reference: the largest state of the frame before (its waves' maxima, left before the last barrier)
      float rf = L.red[(tf & 1) * 32];
      for (int q = 1; q < nwv; ++q) rf = fmaxf(rf, L.red[(tf & 1) * 32 + q]);
      ref = (rf > WFL_NEG_INF && rf < __builtin_inff()) ? (double)rf : 0.0;
      if (tid == 0 && Mref) Mref[t] = ref;
    }
    float top = WFL_NEG_INF;
    for (int s = tid; s < C; s += NT) {
      VT m = pmv[s];
      int am = SR == WFL_SEMIRING_LOG ? 0 : __float_as_int(L.ps[s]);
      for (int r = 1; r < R; ++r) {
        const VT v = pmv[r * C + s];
        if (v > m) {
          m = v;
          if (SR != WFL_SEMIRING_LOG) am = __float_as_int(L.ps[r * C + s]);
        }
      }
      VT val = m;
      if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF) {
        float sum = 0.f;
        for (int r = 0; r < R; ++r) sum += L.ps[r * C + s] * fast_exp((float)(pmv[r * C + s] - m));
        val = m + (VT)lse_log((double)sum);  // (sum >= 1: it contains the largest partial's own sum)
      }
      if (DIR == 0) val += (VT)xr[s];
      to[s] = val;
      ob[(int64_t)t * C + s] = (float)(val - (VT)ref);
      top = fmaxf(top, (float)val);
      if (SR == WFL_SEMIRING_TROPICAL) bptr[((int64_t)b * T + t) * C + s] = am;
    }
    if (kLog) {
      top = wave_max(top);
      if (lane == 0) L.red[(t & 1) * 32 + wv] = top;
    }
    lds_barrier();
  }
  if (DIR == 0 && logz && T > 0) {
    const VT* fin = ((T - 1) & 1) ? a1 : a0;
    float m = WFL_NEG_INF;
    for (int i = tid; i < C; i += NT) m = fmaxf(m, (float)fin[i]);
    m = blk_max(m, L.red);
    double z = m;
    if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF) {
      float s = 0.f;
      for (int i = tid; i < C; i += NT) s += fast_exp((float)(fin[i] - (VT)m));
      s = blk_sum(s, L.red);
      z = (double)m + log((double)s);
    }
    if (tid == 0) {
      logz[b] = (float)z;
      if (kLog && z64) *z64 = z;  // ln Z for the log-domain gradient (DenseWs::z2 holds log2 Z for the others)
    }
  }
}

template <int SR>
__global__ void __launch_bounds__(256)
    dense_chain_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C, int ldw,
                       float* __restrict__ alpha, float* __restrict__ beta, int32_t* __restrict__ bptr,
                       float* __restrict__ logz, const int32_t* __restrict__ only_flagged, double* __restrict__ Mws,
                       double* __restrict__ z2ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, NT = blockDim.x;
  if (only_flagged && !(only_flagged[2 * b] | only_flagged[2 * b + 1])) return;  // served by the fast sweep
  const int R = C <= NT ? NT / C : 1;
  DenseLds L;
  constexpr int kPad = 64;  // (dense_chain reads up to kSpanRegs terms past the vectors' ends: zeros)
  // (the arrays that hold doubles in the log semiring first: 8-byte aligned whatever (C+1) * ldw is)
  double* pd = (double*)smem;
  L.a0 = pd, pd += C + kPad;
  L.a1 = pd, pd += C + kPad;
  L.pm = pd, pd += (size_t)R * C;
  float* p = (float*)pd;
  L.W = p, p += (size_t)(C + 1) * ldw;
  L.x0 = p, p += C + kPad;
  L.x1 = p, p += C + kPad;
  for (int i = tid; i < kPad; i += NT) L.x0[C + i] = L.x1[C + i] = 0.f;
  // (the state vectors: zeros throughout -- their padding is read whether they hold C doubles or C floats)
  for (int i = tid; i < 2 * (C + kPad); i += NT) ((float*)L.a0)[i] = ((float*)L.a1)[i] = 0.f;
  L.ps = p, p += (size_t)R * C;
  L.red = p;
  for (int i = tid; i < (C + 1) * C; i += NT) L.W[(i / C) * ldw + (i % C)] = nan_to_neg(W[i]);
  __syncthreads();
  // (log semiring: the frames' references go to DenseWs::M [B][2][T], ln Z to DenseWs::z2 -- see DenseVal)
  if (dir == 0)
    dense_chain<SR, 0>(L, x, b, T, C, ldw, alpha, bptr, logz, Mws ? Mws + (int64_t)b * 2 * T : nullptr, z2ws ? z2ws + b : nullptr);
  else
    dense_chain<WFL_SEMIRING_LOG, 1>(L, x, b, T, C, ldw, beta, nullptr, nullptr, Mws ? Mws + (int64_t)b * 2 * T + T : nullptr, nullptr);
}

static size_t dense_chain_lds(int C, int ldw) {
  const int R = C <= 256 ? 256 / C : 1;
  // (a0, a1, pm sized for doubles)
  return 4 * ((size_t)(C + 1) * ldw + 6 * ((size_t)C + 64) + 3 * (size_t)R * C + 64) + 64;
}

// ------------------------------------------------------------------------------------------------
// ASG.viterbi (asg.py:211-236) for up to 256 classes: the max-plus sweep with the transition matrix in REGISTERS and
// no back-pointers, then a back-trace that re-derives the one back-pointer per frame it needs.
//
// dense_chain_kernel<tropical> above keeps W in LDS slices, tracks the arg max term by term (five instructions a term),
// merges partial maxima behind a second barrier and stores C back-pointers per frame: 2.3 us per frame at C = 100 --
// 2.3 ms per call at the ASG benchmark's shape, five times the training step it is called beside (train.py:279).  Here
// a state's row of W is split over the two lanes of a pair (SPAN = ceil(C / 2) registers each), a frame is SPAN / 4
// 16-byte reads of the previous vector (two addresses per wave: the pair's halves), SPAN / 2 packed adds and as many
// three-operand maxima, one DPP exchange inside the pair, one LDS write, one barrier.  The arithmetic is the old
// kernel's, operation for operation -- (alpha[j] + W[s][j]) rounded, the maximum, + x[t][s] -- so the stored vectors
// are bit-identical, and so is the path: the back-trace takes  arg max_j alpha[t-1][j] + W[cur][j]  over the STORED
// vector with the lowest j winning ties (what the old sweep's strict comparisons in ascending j did).
// ------------------------------------------------------------------------------------------------
typedef float vit_v2f __attribute__((ext_vector_type(2)));
constexpr int kVitPrefetch = 8;  // emission rows in flight per thread (their loads share a counter with the frames' stores)
template <int SPAN, int R>  // R lanes per state (adjacent: 2 or 4), SPAN previous states per lane
__global__ void __launch_bounds__(512)
    dense_viterbi_sweep_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C,
                               float* __restrict__ alpha) {
  static_assert(SPAN % 4 == 0 && SPAN <= 128, "16-byte reads of the previous vector; the slice lives in registers");
  static_assert(R == 2 || R == 4, "the lanes of a state meet inside a quad");
  // the two vectors, each with R * SPAN readable entries (entries >= C: zeros, met by weights of -inf)
  __shared__ __attribute__((aligned(16))) float av[2][R * SPAN];
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int s = tid / R, r = tid & (R - 1), j0 = r * SPAN;
  const bool owner = r == 0 && s < C;
  const float* xb = x + (int64_t)b * T * C;
  float* ob = alpha + (int64_t)b * T * C;
  const int sc = min(s, C - 1);  // (clamped: every thread loads, no branch around a load -- behind one, each is a round trip)
  vit_v2f w[SPAN / 2];
  {
    const float* wrow = W + (int64_t)(1 + sc) * C;
#pragma unroll
    for (int q = 0; q < SPAN / 2; ++q) {
      const int ja = j0 + 2 * q, jb = ja + 1;
      w[q].x = wrow[min(ja, C - 1)], w[q].y = wrow[min(jb, C - 1)];
    }
#pragma unroll
    for (int q = 0; q < SPAN / 2; ++q) {
      const int ja = j0 + 2 * q, jb = ja + 1;
      w[q].x = (s < C && ja < C) ? nan_to_neg(w[q].x) : WFL_NEG_INF;
      w[q].y = (s < C && jb < C) ? nan_to_neg(w[q].y) : WFL_NEG_INF;
    }
  }
  for (int i = tid; i < 2 * R * SPAN; i += NT) (&av[0][0])[i] = 0.f;
  __syncthreads();
  if (owner) {
    const float v = nan_to_neg(xb[s]) + nan_to_neg(W[s]);
    av[0][s] = v;
    ob[s] = v;
  }
  // One frame: `xv` is x[t][s] (raw).
  auto frame = [&](int t, float xv) {
    const float4* from = reinterpret_cast<const float4*>(&av[(t - 1) & 1][j0]);
    float m0 = WFL_NEG_INF, m1 = WFL_NEG_INF;
#pragma unroll
    for (int q = 0; q < SPAN / 4; ++q) {
      const float4 a = from[q];
      const vit_v2f lo = vit_v2f{a.x, a.y} + w[2 * q], hi = vit_v2f{a.z, a.w} + w[2 * q + 1];
      m0 = __builtin_fmaxf(__builtin_fmaxf(m0, lo.x), lo.y);
      m1 = __builtin_fmaxf(__builtin_fmaxf(m1, hi.x), hi.y);
    }
    float m = __builtin_fmaxf(m0, m1);
    // the rest of the state's row: the other lanes of the pair / quad (quad_perm [1, 0, 3, 2], then [2, 3, 0, 1])
    m = __builtin_fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, true)));
    if (R == 4) m = __builtin_fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, true)));
    const float val = m + nan_to_neg(xv);
    if (owner) {
      av[t & 1][s] = val;
      ob[(int64_t)t * C + s] = val;
    }
    lds_barrier();
  };
  // Emission rows: two register sets of kVitPrefetch rows, used alternately -- a set is requested while the other one's
  // frames run and nothing is copied between them (a copy is a wait for the set's loads AND, the counter being shared
  // and in order, for every store issued before them: a store round trip per group).
  const float* xs = xb + sc;
  float xa[kVitPrefetch], xz[kVitPrefetch];
  int t = 1;
  const int tfull = 1 + ((T - 1) / (2 * kVitPrefetch)) * (2 * kVitPrefetch);  // frames [1, tfull): whole double groups
  if (t < tfull) {
#pragma unroll
    for (int k = 0; k < kVitPrefetch; ++k) xa[k] = xs[(int64_t)(t + k) * C];
  }
  __syncthreads();
  for (; t < tfull; t += 2 * kVitPrefetch) {
#pragma unroll
    for (int k = 0; k < kVitPrefetch; ++k) xz[k] = xs[(int64_t)(t + kVitPrefetch + k) * C];
#pragma unroll
    for (int k = 0; k < kVitPrefetch; ++k) frame(t + k, xa[k]);
#pragma unroll
    for (int k = 0; k < kVitPrefetch; ++k) xa[k] = xs[(int64_t)min(t + 2 * kVitPrefetch + k, T - 1) * C];
#pragma unroll
    for (int k = 0; k < kVitPrefetch; ++k) frame(t + kVitPrefetch + k, xz[k]);
  }
  for (; t < T; ++t) frame(t, xs[(int64_t)t * C]);  // (< 2 * kVitPrefetch frames)
}

// The path from the stored vectors: one workgroup per utterance; the rows of alpha travel to LDS in chunks (coalesced,
// all threads), wave 0 walks a chunk: lane l holds W[cur][l + 64 i] (re-read when the state changes: from LDS when the
// matrix fits beside the chunk, from L2 otherwise) and a step is <= 4 LDS reads, as many adds, a wave-wide maximum and
// the lowest lane / slot that attains it.  Two chunk buffers: waves 1 .. 3 fetch the next chunk while wave 0 walks this one.
constexpr int kVitChunkFloats = 4 * 1024;  // 16 KiB of stored vectors per chunk
template <bool WLDS, int NJ>  // NJ = ceil(C / 64): previous states per lane
__global__ void __launch_bounds__(256)
    dense_viterbi_backtrace_kernel(const float* __restrict__ alpha, const float* __restrict__ W, int B, int T, int C,
                                   int32_t* __restrict__ path) {
  extern __shared__ __attribute__((aligned(16))) char vsmem[];
  float* rowbuf = reinterpret_cast<float*>(vsmem);                    // [2][kVitChunkFloats + 256] (a padded row beyond the last)
  int32_t* walk = reinterpret_cast<int32_t*>(rowbuf + 2 * (kVitChunkFloats + 256));  // [rows per chunk] the chunk's stretch of the path
  int* cur_s = walk + 1024;
  float* wl = reinterpret_cast<float*>(cur_s + 4);                  // [C * C] (WLDS): transitions INTO state i at wl[i * C + j]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (b >= B || T <= 0) return;
  if (WLDS)
    for (int i = tid; i < C * C; i += 256) wl[i] = nan_to_neg(W[C + i]);
  if (tid < 64) {  // arg max of the last frame: the first maximum (lowest state)
    const float* fin = alpha + ((int64_t)b * T + (T - 1)) * C;
    float best = -__builtin_inff();
    int arg = 0x3fffffff;
    for (int i = tid; i < C; i += 64) {
      const float v = fin[i];
      if (v > best || arg == 0x3fffffff) best = v, arg = i;
    }
    const float m = wave_all_max(best);
    arg = -wave_all_max_int(-(best == m ? arg : 0x3fffffff));
    if (tid == 0) *cur_s = arg == 0x3fffffff ? 0 : arg;
  }
  const int rows_per_chunk = max(1, min(kVitChunkFloats / C, 1024));
  int32_t* out = path + (int64_t)b * T;
  const float* ab = alpha + (int64_t)b * T * C;
  int jc[NJ];  // this lane's previous states, clamped (a clamped slot's weight is -inf)
#pragma unroll
  for (int i = 0; i < NJ; ++i) jc[i] = min(lane + 64 * i, C - 1);
  // the walk from frame t1 down to t0 + 1 reads the vectors of frames t1 - 1 .. t0 (frame -1: none, the walk ends);
  // buffer k holds rows[k] = vector of frame t0 + k, k = 0 .. t1 - t0 - 1
  auto fetch = [&](int t1, float* rows, int from, int step) {
    const int t0 = max(t1 - rows_per_chunk, -1), first = max(t0, 0);
    for (int i = from; i < (t1 - first) * C; i += step) rows[(first - t0) * C + i] = ab[(int64_t)first * C + i];
  };
  fetch(T - 1, rowbuf, tid, 256);
  int which = 0;
  for (int t1 = T - 1; t1 >= 0; t1 -= rows_per_chunk, which ^= 1) {
    const int t0 = max(t1 - rows_per_chunk, -1);
    const int nrow = t1 - t0;  // frames t0 + 1 .. t1 get their label
    float* rows = rowbuf + which * (kVitChunkFloats + 256);
    __syncthreads();
    if (tid >= 64 && t0 >= 0) fetch(t0, rowbuf + (which ^ 1) * (kVitChunkFloats + 256), tid - 64, 192);
    if (tid < 64) {
      int cur = __builtin_amdgcn_readfirstlane(*cur_s);
      float wr[NJ], pv[NJ];
      int held = -1;
      // (the vector a step adds to is requested a step ahead: its address does not depend on the state)
#pragma unroll
      for (int i = 0; i < NJ; ++i) pv[i] = rows[max(t1 - 1 - t0, 0) * C + jc[i]];
      for (int t = t1; t > t0; --t) {
        if (lane == 0) walk[t - t0 - 1] = cur;
        if (t == 0) break;
        if (held != cur) {  // (wave-uniform)
#pragma unroll
          for (int i = 0; i < NJ; ++i) {
            const float raw = WLDS ? wl[cur * C + jc[i]] : nan_to_neg(W[(int64_t)(1 + cur) * C + jc[i]]);
            wr[i] = lane + 64 * i < C ? raw : WFL_NEG_INF;
          }
          held = cur;
        }
        float v[NJ], m = WFL_NEG_INF;
#pragma unroll
        for (int i = 0; i < NJ; ++i) v[i] = pv[i] + wr[i], m = vmax(m, v[i]);
#pragma unroll
        for (int i = 0; i < NJ; ++i) pv[i] = rows[max(t - 2 - t0, 0) * C + jc[i]];  // (the next step's; a spare read at the chunk's end)
        const float top = wave_all_max(m);
        // the lowest previous state that attains the maximum: slot by slot, the lowest lane of the first slot with a hit
        int nxt = 0;
#pragma unroll
        for (int i = NJ - 1; i >= 0; --i) {
          const unsigned long long hit = __ballot(v[i] == top);
          nxt = hit ? 64 * i + (int)__builtin_ctzll(hit) : nxt;
        }
        cur = top > WFL_NEG_INF ? nxt : 0;  // unreachable state (all -inf): keep the path well-formed
      }
      if (lane == 0) *cur_s = cur;
    }
    __syncthreads();
    for (int i = tid; i < nrow; i += 256) out[t0 + 1 + i] = walk[i];  // (the next chunk's walk starts behind the loop's first barrier)
  }
}

// ------------------------------------------------------------------------------------------------
// gradient
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ void __launch_bounds__(256)
    dense_grad_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C,
                      const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ logz,
                      const float* __restrict__ coef, const float* __restrict__ coef_w, const float* __restrict__ gout,
                      int accumulate, const float* __restrict__ addend, float* __restrict__ dx,
                      float* __restrict__ partial, int rows_per_block, const int32_t* __restrict__ only_flagged,
                      const double* __restrict__ Mws, const double* __restrict__ z2ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ap = (float*)smem;  // [C] alpha_{t-1} (relative to its frame's reference)
  float* xb = ap + C;        // [C] x_t + beta_t + (the references of alpha_{t-1} and beta_t - ln Z)
  const int b = blockIdx.y, tid = threadIdx.x, NT = 256;
  if (only_flagged && !(only_flagged[2 * b] | only_flagged[2 * b + 1])) return;
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = (coef ? coef[b] : 1.f) * g0;
  const float cw = (coef_w ? coef_w[b] : 1.f) * g0;
  const float zf = logz[b];
  const bool dead = !(zf > WFL_NEG_INF) || !(zf < __builtin_inff());
  // alpha, beta: floats relative to a double per frame (DenseWs::M, written by dense_chain), ln Z a double: the three
  // are thousands apart at T = 1000, their sum is formed in double and only then rounded
  const double* Ma = Mws + (int64_t)b * 2 * T;
  const double* Mb = Ma + T;
  const double zd = dead ? 0.0 : z2ws[b];
  const int t_begin = blockIdx.x * rows_per_block, t_end = min(T, t_begin + rows_per_block);
  const int npairs = C * C;
  const int64_t base = (int64_t)b * T * C;
  // NP * 256 transition pairs per pass over the block's frames (one pass up to C = 128; wider matrices stream the
  // frames again for the next slice of pairs -- a fallback, not a fast path)
  for (int pass0 = 0; pass0 < (partial ? npairs : 1); pass0 += NP * NT) {
  const bool first_pass = pass0 == 0;
  float acc[NP], wreg[NP];
  int pij[NP];  // (i << 16) | j
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = pass0 + tid + k * NT;
    acc[k] = 0.f;
    pij[k] = p < npairs ? ((p / C) << 16) | (p % C) : 0;
    wreg[k] = p < npairs ? nan_to_neg(W[C + p]) : WFL_NEG_INF;  // W[1+i, j]
  }
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    const float k_post = (float)(Ma[t] + Mb[t] - zd);                  // alpha_t + beta_t - ln Z
    const float k_pair = (float)(Ma[t > 0 ? t - 1 : 0] + Mb[t] - zd);  // alpha_{t-1} + ... + beta_t - ln Z
    for (int i = tid; i < C; i += NT) {
      const float a = alpha[base + (int64_t)t * C + i], be = beta[base + (int64_t)t * C + i];
      if (dx && first_pass) {
        const float post = dead ? 0.f : fast_exp(a + be + k_post);
        const int64_t o = base + (int64_t)t * C + i;
        dx[o] = (accumulate ? dx[o] : 0.f) + (addend ? g0 * addend[o] : 0.f) + cf * (post == post ? post : 0.f);
      }
      if (partial) {
        ap[i] = t > 0 ? alpha[base + (int64_t)(t - 1) * C + i] : WFL_NEG_INF;
        xb[i] = nan_to_neg(x[base + (int64_t)t * C + i]) + be + k_pair;
      }
    }
    if (!partial) continue;
    __syncthreads();
    if (t > 0 && !dead) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const float v = ap[pij[k] & 0xffff] + wreg[k] + xb[pij[k] >> 16];
        acc[k] += (v > WFL_NEG_INF) ? fast_exp(v) : 0.f;
      }
    }
  }
  if (partial) {
    float* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (int64_t)(C + 1) * C;
    // start arcs: posterior of frame 0 (only the block that owns t = 0)
    if (first_pass)
      for (int i = tid; i < C; i += NT) {
        float v = 0.f;
        if (t_begin == 0 && !dead) {
          const float p = fast_exp(alpha[base + i] + beta[base + i] + (float)(Ma[0] + Mb[0] - zd));
          v = (p == p) ? p * cw : 0.f;
        }
        dst[i] = v;
      }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int p = pass0 + tid + k * NT;
      if (p < npairs) dst[C + p] = acc[k] * cw;
    }
  }
  }  // pass
}

// dW[i] = (accumulate ? dW[i] : 0) + gout * addend[i] + sum_k partial[k][i]: 32 elements x 8 slices of the partials
// per workgroup, merged in LDS in a fixed order (deterministic)
__global__ void __launch_bounds__(256)
    dense_reduce_kernel(const float* __restrict__ partial, int nblk, int n, float* __restrict__ dW, int accumulate,
                        const float* __restrict__ addend, const float* __restrict__ gout) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  float s = 0.f;
  if (i < n) {
    const int per = (nblk + 7) / 8;
    const int k0 = g * per, k1 = min(nblk, k0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s0 += partial[(int64_t)k * n + i];
      s1 += partial[(int64_t)(k + 1) * n + i];
      s2 += partial[(int64_t)(k + 2) * n + i];
      s3 += partial[(int64_t)(k + 3) * n + i];
    }
    for (; k < k1; ++k) s0 += partial[(int64_t)k * n + i];
    s = (s0 + s1) + (s2 + s3);
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    float t = red[0][e];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += red[q][e];
    dW[i] = (accumulate ? dW[i] : 0.f) + (addend ? (gout ? gout[0] : 1.f) * addend[i] : 0.f) + t;
  }
}

// One workgroup per utterance.  Following the back-pointers is a chain of T dependent loads: from global memory that
// is a memory round trip per frame (measured: 0.5 ms of the 0.8 ms of a bigram Transducer.viterbi at T = 250), so
// the rows travel to LDS in chunks (coalesced, all threads) and one lane walks the chunk there (~50 cycles a step).
constexpr int kBtChunkWords = 12 * 1024;  // 48 KiB of back-pointers per chunk
__global__ void __launch_bounds__(256) dense_backtrace_kernel(const float* __restrict__ alpha, const int32_t* __restrict__ bptr,
                                                               int B, int T, int C, int32_t* __restrict__ path) {
  __shared__ int32_t rows[kBtChunkWords];
  __shared__ int32_t walk[1024];  // the chunk's stretch of the path, written out coalesced
  __shared__ int cur_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B || T <= 0) return;
  if (tid < 64) {  // arg max of the last frame: the first maximum (lowest state)
    const float* fin = alpha + ((int64_t)b * T + (T - 1)) * C;
    float best = -__builtin_inff();
    int arg = 0x3fffffff;
    for (int i = tid; i < C; i += 64) {
      const float v = fin[i];
      if (v > best || arg == 0x3fffffff) best = v, arg = i;
    }
    const float m = wave_all_max(best);
    arg = -wave_all_max_int(-(best == m ? arg : 0x3fffffff));
    if (tid == 0) cur_s = arg == 0x3fffffff ? 0 : arg;
  }
  const int rows_per_chunk = max(1, min(kBtChunkWords / C, 1024));
  int32_t* out = path + (int64_t)b * T;
  // frames (t0, t1]: rows t0 + 1 .. t1 of bptr are needed to step from t1 down to t0
  for (int t1 = T - 1; t1 >= 0; t1 -= rows_per_chunk) {
    const int t0 = max(t1 - rows_per_chunk, -1);  // the walk covers frames t1 .. t0 + 1
    const int nrow = t1 - t0;
    const int32_t* src = bptr + ((int64_t)b * T + (t0 + 1)) * C;
    if (C <= kBtChunkWords)
      for (int i = tid; i < nrow * C; i += 256) rows[i] = src[i];
    __syncthreads();
    if (tid == 0) {
      int cur = cur_s;
      for (int t = t1; t > t0; --t) {
        walk[t - t0 - 1] = cur;
        if (t > 0) {
          cur = C <= kBtChunkWords ? rows[(t - t0 - 1) * C + cur] : src[(int64_t)(t - t0 - 1) * C + cur];
          if (cur < 0) cur = 0;  // unreachable state (all -inf): keep the path well-formed
        }
      }
      cur_s = cur;
    }
    __syncthreads();
    for (int i = tid; i < nrow; i += 256) out[t0 + 1 + i] = walk[i];
    __syncthreads();
  }
}

// =================================================================================================
// Probability-domain sweeps (log semiring, C <= 192)
//
//   P[i][j]     = 2^((W[1+i][j] - rowmax_i) * log2 e)                     in (0, 1], rows of the thread
//   e_t[i]      = 2^((x[t,i] + rowmax_i) * log2 e - mx2_t)                mx2_t = max_i of the exponent
//   a~_t[i]     = e_t[i] * 2^-kk_{t-1} * sum_j P[i][j] a~_{t-1}[j]        (alpha; beta is the transpose)
//   alpha_t[i]  = a~_t[i] * 2^(E_t + M_t),   E_t = sum kk (int),  M_t = sum mx2 (double)
//
// kk is an integer exponent read off the vector itself that keeps it near 2^30: an exact scaling.  One
// workgroup per (utterance, direction), five waves.  Waves 0-3 (one per SIMD) run the matrix-vector product: a
// wave covers 32 states, and each state's dot product is split over TWO lanes (lane l and l ^ 16 take the two
// halves of the vector), so a lane keeps HALF a row (alpha) / column (beta) of P in registers and the halves meet
// in one cross-lane add.  The frame vector reaches the multipliers through DPP, not through LDS: every row of 16
// lanes holds 16 consecutive elements of its half of the vector per register (NCH <= 4 registers, filled by ONE
// ds_read_b128 per lane and frame from a transposed copy), and element k is broadcast inside the row by the
// multiply-add itself (v_fmac_f32_dpp row_newbcast:k).  Delivering every v[j] to every lane through LDS costs
// 128 x 128 x 4 B = 64 KiB of LDS return traffic per frame, ~512 cycles at 128 B/clk whatever the lane layout
// (measured: ~1050 cycles per frame for both the two-wave and the four-wave ds_read_b128 variants); the DPP form
// moves 4 KiB.  One barrier per frame; wave 4 runs ahead, turning emission rows into e_t / mx2_t in an LDS ring.
// Range check: every stored a~ / b~ must stay >= 2^-95 (and finite) and every finite transition
// within 2^-60 of its row maximum; otherwise the utterance is flagged and the log-domain kernels
// recompute it (flag[b][dir]).
// =================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kNormExp = 30;
constexpr float kFloor = 0x1p-95f;
constexpr float kHardGap = 60.f;

struct DenseWs {
  double* M;      // [B][2][T]  cumulative emission offsets (log2), indexed by frame
  double* z2;     // [B]        log2 Z
  int32_t* E;     // [B][2][T]  cumulative renormalisation exponents, indexed by frame
  float* mx2;     // [B][T]     per-frame emission offset (log2) for t >= 1
  float* wr2;     // [256]      row maxima of W[1:, :] in log2 units
  int32_t* flag;  // [B][2]     1: the fast sweep could not represent this utterance
};

__host__ __device__ inline DenseWs dense_ws_carve(void* ws, int B, int T) {
  DenseWs w;
  char* p = (char*)ws;
  w.M = (double*)p, p += (size_t)8 * B * 2 * T;
  w.z2 = (double*)p, p += (size_t)8 * B;
  w.E = (int32_t*)p, p += (size_t)4 * B * 2 * T;
  w.mx2 = (float*)p, p += (size_t)4 * B * T;
  w.wr2 = (float*)p, p += (size_t)4 * 256;
  w.flag = (int32_t*)p;
  return w;
}
static size_t dense_ws_bytes(int B, int T) {
  return (size_t)8 * B * 2 * T + (size_t)8 * B + (size_t)4 * B * 2 * T + (size_t)4 * B * T + 4 * 256 + (size_t)8 * B + 64;
}

// chain waves of a sweep: a wave covers 32 states -- four up to 128 classes (one per SIMD), one per 32 states beyond (160
// and 192 classes: five / six, two on some SIMDs; a lane then keeps up to 96 transition factors in registers)
template <int CP>
constexpr int dense_chain_waves() { return CP <= 128 ? 4 : CP / 32; }
template <int CP>
constexpr int dense_threads() { return (dense_chain_waves<CP>() + 1) * 64; }
template <int CP>
struct FastLds {
  static constexpr int H = CP / 2;                 // states per half of the vector (both halves equally full)
  static constexpr int NCH = (H + 15) / 16;        // 16-element chunks per half
  static constexpr int LASTN = H - 16 * (NCH - 1); // elements of the last chunk (the others are full)
  static constexpr int NC4 = (NCH + 3) / 4 * 4;    // ... rounded up to whole 16-byte reads
  static constexpr int NW = dense_chain_waves<CP>();
  float vecT[2][2][16][NC4];  // frame vector (ping-pong): [half][position in chunk][chunk]: a lane's NCH elements contiguous
  float eh[4][CP];         // ring of e_t rows staged by the helper wave
  float wr2[CP];
  float st2[CP];           // start weights W[0, :] in log2 units
  float wsum[NW];
  float sinv[2];           // 2^-kk of step n at [n & 1], written by the helper wave one interval ahead
  double mtot;
};

template <int CP, int DIR, bool PAIR>
__device__ __forceinline__ void dense_fast_sweep(const float* __restrict__ x, const float* __restrict__ W, int T, int C,
                                                 void* wsp, int B, float* __restrict__ out, float* __restrict__ logz,
                                                 FastLds<CP>& L) {
  constexpr int kDenseChainWaves = FastLds<CP>::NW, kDenseThreads = dense_threads<CP>();
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const DenseWs ws = dense_ws_carve(wsp, B, T);
  const float* xb = x + (int64_t)b * T * C;
  float* ob = out + (int64_t)b * T * C;
  double* Mb = ws.M + ((int64_t)b * 2 + DIR) * T;
  int32_t* Eb = ws.E + ((int64_t)b * 2 + DIR) * T;

  // ---- row maxima of the transition matrix (one wave per row), start weights.  Four rows per trip with their loads
  // (clamped addresses, no test around them) issued together: row by row, and the matrix below element by element
  // inside its bounds tests, the prologue was ~70 dependent L2 round trips -- a tenth of the sweep at T = 1000.
  for (int i0 = wave; i0 < CP; i0 += 4 * (kDenseChainWaves + 1)) {
    float a0[4], a1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ic = min(i0 + u * (kDenseChainWaves + 1), C - 1);
      a0[u] = W[(1 + ic) * C + min(lane, C - 1)];
      a1[u] = W[(1 + ic) * C + min(lane + 64, C - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * (kDenseChainWaves + 1);
      float m = fmaxf(lane < C ? a0[u] : WFL_NEG_INF, lane + 64 < C ? a1[u] : WFL_NEG_INF);
      if (C > 128)
        for (int j = lane + 128; j < C; j += 64) m = fmaxf(m, W[(1 + min(i, C - 1)) * C + j]);
      m = wave_all_max(m);
      if (lane == 0 && i < CP) L.wr2[i] = i < C ? m * kLog2e : 0.f;
    }
  }
  for (int i = tid; i < CP; i += kDenseThreads) L.st2[i] = i < C ? nan_to_neg(W[i]) * kLog2e : WFL_NEG_INF;
  __syncthreads();
  if (b == 0 && DIR == 0)
    for (int i = tid; i < C; i += kDenseThreads) ws.wr2[i] = L.wr2[i];

  // ---- chain waves: lane (wave, hh, qq, il) works on state q = 32 wave + 16 hh + il and on the half qq of the
  // vector: its half of the state's row (alpha) / column (beta) of P
  constexpr int H = FastLds<CP>::H, NCH = FastLds<CP>::NCH, LASTN = FastLds<CP>::LASTN;
  const int qq = (lane >> 4) & 1;                         // which half of the vector this lane multiplies
  const int il = lane & 15;
  const int q = 32 * wave + 16 * (lane >> 5) + il;        // state of a chain lane
  const bool owner = qq == 0;                              // the lane that finishes the state
  float P[16 * NCH];
  int hard = 0;
  for (int i = tid; i < 2 * 2 * 16 * FastLds<CP>::NC4; i += kDenseThreads) (&L.vecT[0][0][0][0])[i] = 0.f;  // (padding stays 0)
  if (wave < kDenseChainWaves) {
    // (all of the lane's transition scores first, from clamped addresses: P[] holds the raw scores until the second loop)
    const int qc = min(q, C - 1);
#pragma unroll
    for (int k = 0; k < 16 * NCH; ++k) {
      const int jc = min(qq * H + k, C - 1);
      P[k] = DIR == 0 ? W[(1 + qc) * C + jc] : W[(1 + jc) * C + qc];
    }
#pragma unroll
    for (int k = 0; k < 16 * NCH; ++k) {
      const int j = qq * H + k;
      float p = 0.f;
      if (k < H && q < C && j < C) {
        const float d = P[k] * kLog2e - (DIR == 0 ? L.wr2[q] : L.wr2[j]);
        hard |= !(d >= -kHardGap);  // -inf, NaN, +inf rows, or a dynamic range the floor check cannot vouch for
        p = __builtin_amdgcn_exp2f(d);
      }
      P[k] = p;
    }
  }
  if (__syncthreads_or(hard)) {
    if (tid == 0) ws.flag[2 * b + DIR] = 1;
    return;
  }

  // ---- helper wave state: a lane covers two states -- lane and lane + 64, or (PAIR: an even number of classes and
  // an 8-byte aligned tensor) 2 lane and 2 lane + 1, whose scores then arrive in ONE 8-byte load per row -- and, beyond
  // 128 classes, a third: 128 + lane
  // item r is the emission row the chain multiplies in at step r: frame r (alpha), frame T - r (beta)
  constexpr int NS = CP <= 128 ? 2 : 3;
  static_assert(CP <= 192, "states per helper lane");
  int ist[NS];
  ist[0] = PAIR ? 2 * lane : lane, ist[1] = PAIR ? 2 * lane + 1 : lane + 64;
  if constexpr (NS > 2) ist[2] = lane + 128;
  bool has[NS];
  float add[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) has[s] = ist[s] < CP, add[s] = L.wr2[has[s] ? ist[s] : 0];
  // Emission rows travel HBM -> registers kDepth items ahead of their use.  (The first version kept 4 in flight: with
  // ~1.7 us of load latency under load that alone pinned the sweep at latency / 4 = ~1000 cycles per frame, whatever
  // the matrix-vector product cost -- measured with three different product layouts.)
  // As deep as the 6-bit vmcnt counter allows: 31 rows of two loads, 48 rows of one (three states: 21 / 31).  With 16,
  // the sweep ran at 343 us alone but at 375 us next to the numerator's gradient kernel streaming 170 MB on the other
  // stream (the round trip grows past the 5.4 us that 16 frames cover); with 31: 358 us.
  constexpr int kDepth = NS == 2 ? (PAIR ? 48 : 31) : (PAIR ? 31 : 21);
  float raw[kDepth][NS];
  double mrun = 0.0;
  auto item_ok = [&](int r) { return DIR == 0 ? r < T : (r >= 1 && r < T); };
  auto issue = [&](int r, float (&dst)[NS]) {
    if (!item_ok(r)) return;
    const float* row = xb + (int64_t)(DIR == 0 ? r : T - r) * C;
#pragma unroll
    for (int s = 0; s < NS; ++s) dst[s] = ist[s] < C ? row[ist[s]] : WFL_NEG_INF;
  };
  // CHECKED = false: the caller guarantees 3 <= r and that the item exists (no branches: the main loop
  // must stay straight-line so that the loads of later items stay in flight across this one's use)
  auto stage = [&](int r, const float (&src)[NS], auto checked) {
    constexpr bool CHECKED = decltype(checked)::value;
    if (CHECKED && !item_ok(r)) {
      if (DIR == 1 && r == 0 && lane == 0) Mb[T - 1] = 0.0;
      return;
    }
    const bool first = CHECKED && DIR == 0 && r == 0;
    float sv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sv[s] = ist[s] < C ? fmaf(nan_to_neg(src[s]), kLog2e, first ? L.st2[ist[s]] : add[s]) : WFL_NEG_INF;
    float mloc = vmax(sv[0], sv[1]);
    if constexpr (NS > 2) mloc = vmax(mloc, sv[2]);
    const float m = wave_all_max(mloc);
    float* dst = L.eh[r & 3];
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (has[s]) dst[ist[s]] = ist[s] < C ? __builtin_amdgcn_exp2f(sv[s] - m) : 0.f;
    mrun += (double)m;
    if (lane == 0) {
      Mb[DIR == 0 ? r : T - 1 - r] = mrun;
      if (DIR == 0) ws.mx2[(int64_t)b * T + r] = m;
    }
  };
  auto issue_fast = [&](int r, float (&dst)[NS]) {  // straight-line: the row index is clamped, items past the end are unused
    const int rr = DIR == 0 ? min(r, T - 1) : max(T - r, 0);
    const float* row = xb + (int64_t)rr * C;
    if (PAIR) {  // (C is even: the last pair starts at C - 2)
      const float2 v = *reinterpret_cast<const float2*>(row + min(ist[0], C - 2));
      dst[0] = v.x, dst[1] = v.y;
    } else {
      dst[0] = row[ist[0] < C ? ist[0] : 0];
      dst[1] = row[ist[1] < C ? ist[1] : 0];
    }
    if constexpr (NS > 2) dst[2] = row[ist[2] < C ? ist[2] : 0];
  };
  // straight-line staging of item r >= 2 (no early return: later items' loads stay in flight across this one's
  // use); items past the end only skip their bookkeeping
  // The bookkeeping words of a frame (running offset M, the frame's offset mx2, the exponent sum E) are NOT stored
  // frame by frame: lane k of the helper keeps those of the k-th frame of a group of kDepth and the group is written
  // with one store per array.  A store per frame from the wave that also waits for the prefetched rows shrinks the
  // prefetch: loads and stores share the in-order vmcnt counter, the conditional stores are not counted by the
  // compiler along the shortest path, and its `s_waitcnt vmcnt(16)` then means "all but the last three frames'
  // operations" -- the rows were effectively requested three frames ahead, not sixteen (371 -> 337 us at cfg3).
  double gM = 0.0;
  float gm2 = 0.f;
  int gE = 0;
  auto stage_fast = [&](int r, const float (&src)[NS], int k) {
    const bool ok = r < T;
    float sv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sv[s] = ist[s] < C ? fmaf(nan_to_neg(src[s]), kLog2e, add[s]) : WFL_NEG_INF;
    float mloc = vmax(sv[0], sv[1]);
    if constexpr (NS > 2) mloc = vmax(mloc, sv[2]);
    const float m = wave_all_max(mloc);
    float* dst = L.eh[r & 3];
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (has[s]) dst[ist[s]] = ist[s] < C ? __builtin_amdgcn_exp2f(sv[s] - m) : 0.f;
    mrun += ok ? (double)m : 0.0;
    gM = lane == k ? mrun : gM;
    gm2 = lane == k ? m : gm2;
  };
  const std::true_type kChecked;

  // ---- chain wave state
  // The scale of step n is 2^-kk with kk the exponent of the largest of the first four elements of the
  // vector being multiplied, minus kNormExp: every lane (and the helper wave, which does the E_t
  // bookkeeping) derives it from the first ds_read_b128 it issues anyway -- no reduction and no
  // extra LDS round trip on the per-frame critical path (barrier -> broadcast reads + FMAs ->
  // ds_write).  Any power of two is exact; how well it centres the vector only matters for the
  // range check below.
  // range check of every stored value, kFloor <= val < 3e38 (NaN fails): the running minimum and maximum of the values'
  // BIT PATTERNS as signed integers -- positive floats order like their bits, NaN sorts above +inf, anything with the
  // sign bit below zero -- two instructions per frame, the comparison once after the sweep
  int vlo = 0x7f7fffff, vhi = 0;
  float last = 0.f;
  auto scale_exp = [&](const float4& v) {
    return __builtin_amdgcn_frexp_expf(vmax(vmax(v.x, v.y), vmax(v.z, v.w))) - kNormExp;
  };
  // where state i sits in the transposed two-halves layout of the frame vector
  auto vslot = [&](int cur, int i) -> float& {
    const int half = i >= H ? 1 : 0, pos = i - half * H;
    return L.vecT[cur][half][pos & 15][pos >> 4];
  };
  // the scale reference: four fixed elements of the vector (v[0], v[16], v[32], v[48]: one aligned 16-B read)
  auto scale_ref = [&](int cur) { return *reinterpret_cast<const float4*>(L.vecT[cur][0][0]); };
  constexpr int kVecBuf = 2 * 16 * FastLds<CP>::NC4;  // floats of one of the two frame-vector buffers
  float* const myslot = &vslot(0, q < CP ? q : 0);
  auto chain_step = [&](int n, int cur) {  // n >= 1, cur = (n - 1) & 1
    const float e = L.eh[(DIR == 0 ? n : n + 1) & 3][q < CP ? q : 0];  // beta at n = T-1: a stale row, unused
    const float4 vc4 = *reinterpret_cast<const float4*>(L.vecT[cur][qq][il]);  // this row's chunks of its half
    float4 vc8 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (NCH > 4) vc8 = *reinterpret_cast<const float4*>(&L.vecT[cur][qq][il][4]);
    const float inv = L.sinv[n & 1];
    const float vc[8] = {vc4.x, vc4.y, vc4.z, vc4.w, vc8.x, vc8.y, vc8.z, vc8.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // element k of the row's chunk, broadcast to the row's 16 lanes by the multiply-add's DPP source
    // (chunk c holds N valid elements: 16, or LASTN in the last one -- at C = 100 the halves are 52 long, so the
    // last chunk issues 4 multiply-adds instead of 16: 52 per frame, not 64, at 6.4 cycles each)
#define WFL_BC(c, k, N)                                                                            \
  if constexpr ((k) < (N))                                                                         \
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf"          \
                 : "+v"(acc[(k) & 3])                                                                  \
                 : "v"(vc[c]), "v"(P[16 * (c) + (k)]));
#define WFL_BC16(c, N)                                                                                                   \
  WFL_BC(c, 0, N) WFL_BC(c, 1, N) WFL_BC(c, 2, N) WFL_BC(c, 3, N) WFL_BC(c, 4, N) WFL_BC(c, 5, N) WFL_BC(c, 6, N)       \
  WFL_BC(c, 7, N) WFL_BC(c, 8, N) WFL_BC(c, 9, N) WFL_BC(c, 10, N) WFL_BC(c, 11, N) WFL_BC(c, 12, N) WFL_BC(c, 13, N)  \
  WFL_BC(c, 14, N) WFL_BC(c, 15, N)
    // (the chunk registers may have been moved by a VALU instruction just before: a DPP read of a VGPR needs two wait
    // states after a VALU write of it, and the hazard recogniser does not look into inline assembly)
    asm volatile("s_nop 1" ::: "memory");
    WFL_BC16(0, NCH == 1 ? LASTN : 16)
    if constexpr (NCH > 1) { WFL_BC16(1, NCH == 2 ? LASTN : 16) }
    if constexpr (NCH > 2) { WFL_BC16(2, NCH == 3 ? LASTN : 16) }
    if constexpr (NCH > 3) { WFL_BC16(3, NCH == 4 ? LASTN : 16) }
    if constexpr (NCH > 4) { WFL_BC16(4, NCH == 5 ? LASTN : 16) }
    if constexpr (NCH > 5) { WFL_BC16(5, NCH == 6 ? LASTN : 16) }
    static_assert(NCH <= 6, "chunks of the frame vector per lane");
#undef WFL_BC16
#undef WFL_BC
    float part = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    {  // + the other half of the dot product (lane ^ 16: same state).  v_permlane16_swap exchanges the odd rows of
       // 16 lanes of its first operand with the even rows of its second: given the same value twice, the two results
       // hold, for every lane, its own value and that of lane ^ 16 (in one order or the other) -- one VALU
       // instruction where __shfl_xor is a ds_bpermute round trip (~65 cycles on the per-frame critical path)
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      const v2u sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(part), __float_as_uint(part), false, false);
      part = __uint_as_float(sw.x) + __uint_as_float(sw.y);
    }
    const float y = inv * part;
    const float val = DIR == 0 ? e * y : y;
    last = val;
    if (owner && q < C) {  // (ONE predicated region per frame; the padding states C .. CP-1 of the vector stay zero)
      myslot[(cur ^ 1) * kVecBuf] = DIR == 0 ? val : e * y;  // vslot(cur ^ 1, q), its address kept across frames
      vlo = min(vlo, __float_as_int(val)), vhi = max(vhi, __float_as_int(val));
      ob[(int64_t)(DIR == 0 ? n : T - 1 - n) * C + q] = val;
    }
  };
  // helper side of the same bookkeeping: E_t = sum of the exponents applied up to step n
  // The scale of step n is PREDICTED an interval ahead by the helper wave and left in LDS as 2^-kk: the chain waves read
  // one word instead of a second ds_read_b128, three max, frexp and ldexp per frame.  With s_n the exponent (minus
  // kNormExp) of the vector step n multiplies -- what the helper can see during interval n -- the vector of step n+1
  // sits at s_n + g_n - kk_n, g_n the growth of the step in flight; the growth of the step before,
  // g_{n-1} = s_n - s_{n-1} + kk_{n-1}, stands in for it: kk_{n+1} = 2 s_n - s_{n-1} + kk_{n-1} - kk_n, which leaves
  // the vector off centre by g_n - g_{n-1} only (no accumulation; any power of two is exact).  Using s_n itself, one
  // frame stale, is a feedback loop with a delay of one step and gain one -- it does not damp and the vector wanders
  // out of the fp32 range (measured: every utterance flagged).  Step 1 is not scaled (kk_1 = 0).
  int ecum = 0, kcur = 0, kprev = 0, sprev = 0;
  auto helper_scale = [&](int n, int k) {
    ecum += kcur;  // kk_n, applied by the chain waves in this interval
    gE = lane == k ? ecum : gE;
    const int sn = scale_exp(scale_ref((n - 1) & 1));
    const int knext = 2 * sn - (n == 1 ? sn : sprev) + kprev - kcur;
    if (lane == 0) L.sinv[(n + 1) & 1] = __builtin_amdgcn_ldexpf(1.f, -knext);
    kprev = kcur, kcur = knext, sprev = sn;
  };

  // ---- prologue: items 0, 1 and 2 staged synchronously, items 3 .. kDepth+2 in flight (item r lives in raw[r % kDepth])
  if (wave == kDenseChainWaves) {
    if (lane == 0) L.sinv[1] = 1.f;  // (step 1)
    issue(0, raw[0]);
    issue(1, raw[1]);
    issue(2, raw[2]);
    stage(0, raw[0], kChecked);
    stage(1, raw[1], kChecked);
    stage(2, raw[2], kChecked);
#pragma unroll
    for (int r = 3; r < kDepth + 3; ++r) issue_fast(r, raw[r % kDepth]);
  }
  __syncthreads();
  // ---- intervals: interval n ends with barrier n; the chain performs step n, the helper stages
  // item n + 2 and issues the loads of item n + 6.  Role-specialised loops with the same barrier count.
  // five waves on four SIMDs: the helper shares one with a chain wave and must not take its issue slots
#ifndef WFL_DENSE_CHAIN_PRIO
#define WFL_DENSE_CHAIN_PRIO 3
#endif
  if (wave < kDenseChainWaves)
    __builtin_amdgcn_s_setprio(WFL_DENSE_CHAIN_PRIO);
  else
    __builtin_amdgcn_s_setprio(0);
  if (wave < kDenseChainWaves) {
    {  // interval 0
      float val, next;
      if (DIR == 0) {
        val = L.eh[0][q < CP ? q : 0];
        next = val;
      } else {
        val = q < C ? 1.f : 0.f;
        next = T > 1 ? L.eh[1][q < CP ? q : 0] * val : 0.f;
      }
      if (owner && q < C) {
        vlo = min(vlo, __float_as_int(val)), vhi = max(vhi, __float_as_int(val));
        ob[(int64_t)(DIR == 0 ? 0 : T - 1) * C + q] = val;
      }
      last = val;
      if (owner && q < CP) vslot(0, q) = next;
    }
    lds_barrier();
    int n = 1;
    for (; n + 1 < T; n += 2) {
      chain_step(n, 0);
      lds_barrier();
      chain_step(n + 1, 1);
      lds_barrier();
    }
    if (n < T) {
      chain_step(n, 0);
      lds_barrier();
    }
  } else {
    if (lane == 0) Eb[DIR == 0 ? 0 : T - 1] = 0;
    lds_barrier();  // interval 0
    // interval n: the helper stages item n + 2 (loaded kDepth intervals ago) and issues the loads of item
    // n + 2 + kDepth into the registers it just freed.  Unrolled by kDepth so that the register ring is indexed by
    // constants; the only branch is the loop exit.
    for (int n = 1; n < T; n += kDepth) {
#pragma unroll
      for (int k = 0; k < kDepth; ++k) {
        if (n + k < T) {
          helper_scale(n + k, k);
          stage_fast(n + k + 2, raw[(k + 3) % kDepth], k);
          issue_fast(n + k + 2 + kDepth, raw[(k + 3) % kDepth]);
          lds_barrier();
        }
      }
      // the group's words: lane k holds those of step n + k (E) and of item n + k + 2 (M, mx2)
      if (lane < kDepth && n + lane < T) {
        Eb[DIR == 0 ? n + lane : T - 1 - (n + lane)] = gE;
        const int r = n + lane + 2;
        if (r < T) {
          Mb[DIR == 0 ? r : T - 1 - r] = gM;
          if (DIR == 0) ws.mx2[(int64_t)b * T + r] = gm2;
        }
      }
    }
  }
  // ---- epilogue: range verdict; log Z from the last alpha vector
  const int any_bad = __syncthreads_or(!(vlo >= __float_as_int(kFloor) && vhi < __float_as_int(3.0e38f)));
  if (DIR == 0) {
    if (wave < kDenseChainWaves) {
      const float s = wave_all_sum((owner && q < C) ? last : 0.f);
      if (lane == 0) L.wsum[wave] = s;
    } else if (lane == 0) {
      L.mtot = mrun + (double)ecum;
    }
    __syncthreads();
    if (tid == 0) {
      float zs = (L.wsum[0] + L.wsum[1]) + (L.wsum[2] + L.wsum[3]);
#pragma unroll
      for (int k = 4; k < kDenseChainWaves; ++k) zs += L.wsum[k];
      const double z2 = L.mtot + (double)__builtin_amdgcn_logf(zs);
      ws.z2[b] = z2;
      logz[b] = (float)(z2 * 0.6931471805599453);
    }
  }
  if (tid == 0) ws.flag[2 * b + DIR] = any_bad ? 1 : 0;
}

template <int CP, bool PAIR>
__global__ void __launch_bounds__(dense_threads<CP>())
    dense_fast_chain_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C, void* wsp, int B,
                            float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ logz) {
  __shared__ __attribute__((aligned(16))) FastLds<CP> L;
  if (blockIdx.y == 0) {
    if (!beta && threadIdx.x == 0) dense_ws_carve(wsp, B, T).flag[2 * blockIdx.x + 1] = 0;
    dense_fast_sweep<CP, 0, PAIR>(x, W, T, C, wsp, B, alpha, logz, L);
  } else {
    dense_fast_sweep<CP, 1, PAIR>(x, W, T, C, wsp, B, beta, nullptr, L);
  }
}

// gradient of the fast sweeps.  Emission gradient: dx[t,i] = cf * a~_t[i] b~_t[i] 2^(La_t + Lb_t - z2).
// Transition gradient: dW[1+i][j] = P[i][j] * sum_t a~_{t-1}[j] * (e_t[i] b~_t[i]) * 2^(La_{t-1} + mx2_t
// + Lb_t - z2): an outer-product accumulation over frames, 8x8 register tile per thread, operands
// staged in LDS TS frames at a time; the power of two is split evenly over both operands so
// that neither leaves the fp32 range.  Per-workgroup partial sums, reduced by dense_reduce_kernel.
//
// The kernel is a stream over three [T, C] arrays with ~300 cycles of FMAs per frame, and only two or three
// workgroups fit the grid on a CU: unpipelined, every stage paid a full HBM round trip with nothing else to run
// (214 us at B=128, T=1000, C=100 against 40 us of traffic and 31 us of FMAs).  The loads of stage s+1 (and the
// per-frame scale words) are therefore issued into registers BEFORE the FMAs of stage s and consumed after them.
__device__ __forceinline__ float at_byte(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int CP, int TS>
__global__ void __launch_bounds__(((CP / 8) * (CP / 8) + 63) / 64 * 64)
    dense_fast_grad_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C, int B,
                           const float* __restrict__ alpha, const float* __restrict__ beta, const void* wsp,
                           const float* __restrict__ coef, const float* __restrict__ coef_w,
                           const float* __restrict__ gout, int accumulate, const float* __restrict__ addend,
                           float* __restrict__ dx, float* __restrict__ partial, int rows_per_block) {
  constexpr int G = CP / 8, NT = (G * G + 63) / 64 * 64, NI = (TS * CP + NT - 1) / NT;
  static_assert(TS <= 64, "the scale words of a stage are computed by the first wave");
  __shared__ __attribute__((aligned(16))) float A[TS][CP];
  __shared__ __attribute__((aligned(16))) float U[TS][CP];
  __shared__ float sc[2][TS][4];
  __shared__ float wr2[CP], s0[CP];
  const int b = blockIdx.y, tid = threadIdx.x;
  const DenseWs ws = dense_ws_carve(const_cast<void*>(wsp), B, T);
  if (ws.flag[2 * b] | ws.flag[2 * b + 1]) return;  // recomputed by the log-domain kernels
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = (coef ? coef[b] : 1.f) * g0;
  const float cw = (coef_w ? coef_w[b] : 1.f) * g0;
  const double z2 = ws.z2[b];
  const double *Ma = ws.M + (int64_t)b * 2 * T, *Mb = Ma + T;
  const int32_t *Ea = ws.E + (int64_t)b * 2 * T, *Eb = Ea + T;
  const float* mx2 = ws.mx2 + (int64_t)b * T;
  const int t_begin = blockIdx.x * rows_per_block, t_end = min(T, t_begin + rows_per_block);
  const int64_t base = (int64_t)b * T * C;
  for (int i = tid; i < CP; i += NT) wr2[i] = i < C ? ws.wr2[i] : 0.f, s0[i] = 0.f;
  const int ti = tid / G, tj = tid - ti * G;
  const bool active = tid < G * G && partial;
  // the (row, column) of this thread's NI staged elements never change: only the stage's first frame does
  int er[NI], ei[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int idx = tid + k * NT;
    er[k] = idx / CP, ei[k] = idx - er[k] * CP;
    if (idx >= TS * CP || ei[k] >= C) er[k] = -1, ei[k] = 0;
  }
  // registers of the stage in flight.  issue() only LOADS (branch-free, from clamped addresses): any arithmetic on
  // a loaded value here would put its s_waitcnt in front of the FMAs the loads are meant to fly under.  Addresses are
  // the utterance's (uniform) base pointers plus 32-bit BYTE offsets (scalar base + vector offset addressing), built
  // with 24-bit multiplies.  With 64-bit per-element addresses the kernel sat at 242 registers and the allocator
  // recycled the destinations of loads still in flight as address temporaries (even the dead upper half of a
  // v_mad_u64_u32 result shares a register with a load in flight): a wait for a full round trip after every few
  // loads of the "prefetch" -- measured 163 -> 127 us for this kernel + the reduction at cfg3.
  float ra[NI], rb[NI], rx[NI], rp[NI], rd[NI], re[NI];
  double q_ma = 0., q_mb = 0., q_mp = 0.;  // lane r < TS: scale words of frame t0 + r (and of the frame before it)
  int32_t q_ea = 0, q_eb = 0, q_ep = 0;
  float q_m2 = 0.f;
  const float* const ab = alpha + base;
  const float* const bb = beta + base;
  const float* const xb = x + base;
  float* const dxb = dx ? dx + base : nullptr;
  const float* const prev_dx = (dx && accumulate) ? dx + base : nullptr;
  const float* const addb = addend ? addend + base : nullptr;
  auto issue = [&](int t0) {
    if (tid < TS) {
      const int t = min(t0 + tid, T - 1), tp = max(t - 1, 0);
      q_ma = Ma[t], q_ea = Ea[t], q_mb = Mb[t], q_eb = Eb[t], q_m2 = mx2[t], q_mp = Ma[tp], q_ep = Ea[tp];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int t = min(t0 + max(er[k], 0), t_end - 1);
      const unsigned o = 4u * (__umul24((unsigned)t, (unsigned)C) + (unsigned)ei[k]);
      ra[k] = at_byte(ab, o), rb[k] = at_byte(bb, o), rx[k] = at_byte(xb, o);
      rp[k] = at_byte(ab, t > 0 ? o - 4u * (unsigned)C : o);
      if (prev_dx) rd[k] = at_byte(prev_dx, o);
      if (addb) re[k] = at_byte(addb, o);
    }
  };
  auto scales = [&](int t0, int buf) {  // first wave: the stage's per-frame powers of two
    if (tid < TS) {
      const int t = t0 + tid;
      const double lb = q_mb + (double)q_eb;
      const float eg = (float)(q_ma + (double)q_ea + lb - z2);
      float ex = WFL_NEG_INF;
      if (t > 0) ex = (float)(q_mp + (double)q_ep + (double)q_m2 + lb - z2);
      sc[buf][tid][0] = __builtin_amdgcn_exp2f(0.5f * eg);
      sc[buf][tid][1] = __builtin_amdgcn_exp2f(0.5f * ex);
      sc[buf][tid][2] = q_m2;
    }
  };
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  issue(t_begin);
  scales(t_begin, 0);
  int buf = 0;
  for (int t0 = t_begin; t0 < t_end; t0 += TS, buf ^= 1) {
    const int nr = min(TS, t_end - t0);
    __syncthreads();  // the previous stage's FMAs are done with A/U; sc[buf] is visible
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (er[k] < 0) continue;
      const int r = er[k], i = ei[k], t = t0 + r;
      float uu = 0.f, aa = 0.f;
      if (t < t_end) {
        const float hg = sc[buf][r][0], hx = sc[buf][r][1];
        const float g = (ra[k] * hg) * (rb[k] * hg);
        if (dxb)
          *reinterpret_cast<float*>(reinterpret_cast<char*>(dxb) + 4u * (__umul24((unsigned)t, (unsigned)C) + (unsigned)i)) =
              (prev_dx ? rd[k] : 0.f) + (addb ? g0 * re[k] : 0.f) + cf * g;
        if (partial) {
          if (t > 0) {
            const float e = __builtin_amdgcn_exp2f(fmaf(nan_to_neg(rx[k]), kLog2e, wr2[i]) - sc[buf][r][2]);
            uu = e * rb[k] * hx * cw;
            aa = rp[k] * hx;
          } else {
            s0[i] = g * cw;
          }
        }
      }
      U[r][i] = uu, A[r][i] = aa;
    }
    __syncthreads();
    if (t0 + TS < t_end) issue(t0 + TS);  // in flight during the FMAs below
    if (active) {
      for (int r = 0; r < nr; ++r) {
        const float4 u0 = *reinterpret_cast<const float4*>(&U[r][8 * ti]);
        const float4 u1 = *reinterpret_cast<const float4*>(&U[r][8 * ti + 4]);
        const float4 a0 = *reinterpret_cast<const float4*>(&A[r][8 * tj]);
        const float4 a1 = *reinterpret_cast<const float4*>(&A[r][8 * tj + 4]);
        const float u[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(u[i], a[j], acc[i][j]);
      }
    }
    if (t0 + TS < t_end) scales(t0 + TS, buf ^ 1);
  }
  if (partial) {
    __syncthreads();
    float* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (int64_t)(C + 1) * C;
    for (int i = tid; i < C; i += NT) dst[i] = s0[i];
    if (active) {
      // all 64 transition scores first, from clamped (always valid) addresses: a load inside the bounds test below
      // would be waited for element by element -- 64 dependent HBM round trips, most of this kernel's time before
      // (in two halves of four rows: 64 more live registers would set the kernel's register count)
#pragma unroll
      for (int ih = 0; ih < 8; ih += 4) {
        float wv[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[i][j] = W[(1 + min(8 * ti + ih + i, C - 1)) * C + min(8 * tj + j, C - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int si = 8 * ti + ih + i;
          const float wr = wr2[min(si, CP - 1)];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int sj = 8 * tj + j;
            const float p = __builtin_amdgcn_exp2f(fmaf(wv[i][j], kLog2e, -wr));
            if (si < C && sj < C) dst[(1 + si) * C + sj] = acc[ih + i][j] * p;
          }
        }
      }
    }
  }
}

// The same gradient with the outer-product accumulation on the MATRIX cores.  The transition gradient
//     dW[1+i][j] = P[i][j] * sum_{(b,t)} U[(b,t)][i] A[(b,t)][j]
// is a matrix product over K = B (T - 1) rows -- the one GEMM-shaped piece of the dense path (the sweeps are
// matrix-VECTOR products per utterance: nothing for a matrix core).  v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate:
// exact fp32 FMAs, the vector pipe's peak rate) takes its operands straight from the staged LDS tiles: lane l feeds
// U[4g + l / 16][16 mi + l % 16] and A[4g + l / 16][16 nj + l % 16] for the k-group g -- rows of consecutive columns,
// no transpose.  Four waves, a 2 x 2 grid of blocks of up to 4 x 4 tiles each; a wave's accumulators are 4 VGPRs per
// tile.  Per stage of 16 frames a wave issues <= 64 MFMAs (2048 cycles of its SIMD's matrix pipe) where the 8 x 8
// register tiles of dense_fast_grad_kernel cost every wave 1024 FMAs (~4400 issue cycles) -- and the vector pipe is
// free for the staging arithmetic of the next stage meanwhile.  Staging, scales and the emission gradient are those
// of dense_fast_grad_kernel.
typedef float mfma_v4f __attribute__((ext_vector_type(4)));
// (stages of 8 frames with three workgroups per CU -- 168 VGPRs -- instead of 16 frames with two, re-measured in round 5:
// cfg3 0.439 -> 0.449 ms, 0.446 with 768 workgroups instead of 512: not kept)
template <int CP, int TS>
__global__ void __launch_bounds__(256, 2)
    dense_mfma_grad_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C, int B,
                           const float* __restrict__ alpha, const float* __restrict__ beta, const void* wsp,
                           const float* __restrict__ coef, const float* __restrict__ coef_w,
                           const float* __restrict__ gout, int accumulate, const float* __restrict__ addend,
                           float* __restrict__ dx, float* __restrict__ partial, int rows_per_block) {
  constexpr int NT = 256, NI = (TS * CP + NT - 1) / NT;
  constexpr int CPT = (CP + 15) / 16, CPP = CPT * 16, H = (CPT + 1) / 2;  // tiles per side, padded width, tiles per wave side
  static_assert(TS <= 64 && TS % 4 == 0, "stages are whole k-groups; the scale words come from the first wave");
  __shared__ __attribute__((aligned(16))) float A[TS][CPP];
  __shared__ __attribute__((aligned(16))) float U[TS][CPP];
  __shared__ float sc[2][TS][4];
  __shared__ float wr2[CPP], s0[CPP];
  const int b = blockIdx.y, tid = threadIdx.x;
  const DenseWs ws = dense_ws_carve(const_cast<void*>(wsp), B, T);
  if (ws.flag[2 * b] | ws.flag[2 * b + 1]) return;  // recomputed by the log-domain kernels
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = (coef ? coef[b] : 1.f) * g0;
  const float cw = (coef_w ? coef_w[b] : 1.f) * g0;
  const double z2 = ws.z2[b];
  const double *Ma = ws.M + (int64_t)b * 2 * T, *Mb = Ma + T;
  const int32_t *Ea = ws.E + (int64_t)b * 2 * T, *Eb = Ea + T;
  const float* mx2 = ws.mx2 + (int64_t)b * T;
  const int t_begin = blockIdx.x * rows_per_block, t_end = min(T, t_begin + rows_per_block);
  const int64_t base = (int64_t)b * T * C;
  for (int i = tid; i < CPP; i += NT) wr2[i] = i < C ? ws.wr2[i] : 0.f, s0[i] = 0.f;
  if constexpr (CPP > CP)
    for (int i = tid; i < TS * (CPP - CP); i += NT) {  // the tiles' columns beyond CP: zeros, written once
      const int r = i / (CPP - CP), c = CP + i % (CPP - CP);
      A[r][c] = 0.f, U[r][c] = 0.f;
    }
  int er[NI], ei[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int idx = tid + k * NT;
    er[k] = idx / CP, ei[k] = idx - er[k] * CP;
    if (idx >= TS * CP || ei[k] >= C) er[k] = -1, ei[k] = 0;
  }
  // TWO stages in flight (register sets X and Y, used alternately): a stage's loads are issued two stages before they
  // are consumed -- with one set they flew under ONE stage's products (~2000 cycles of the matrix pipe) and every
  // stage waited out the rest of an HBM round trip (measured: 5.9 us per stage of 16 frames, two workgroups per CU).
  struct Stage {
    float ra[NI], rb[NI], rx[NI], rp[NI], re[NI];  // (no accumulate-into-dx here: those calls take dense_fast_grad_kernel)
    double q_ma, q_mb, q_mp;  // lane r < TS: scale words of frame t0 + r (and of the frame before it)
    int32_t q_ea, q_eb, q_ep;
    float q_m2;
  };
  const float* const ab = alpha + base;
  const float* const bb = beta + base;
  const float* const xb = x + base;
  float* const dxb = dx ? dx + base : nullptr;
  const float* const addb = addend ? addend + base : nullptr;
  auto issue = [&](Stage& S, int t0) {
    if (tid < TS) {
      const int t = min(max(t0, 0) + tid, T - 1), tp = max(t - 1, 0);
      S.q_ma = Ma[t], S.q_ea = Ea[t], S.q_mb = Mb[t], S.q_eb = Eb[t], S.q_m2 = mx2[t], S.q_mp = Ma[tp], S.q_ep = Ea[tp];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int t = min(t0 + max(er[k], 0), t_end - 1);  // (a stage past the end: the last frame again, never used)
      const unsigned o = 4u * (__umul24((unsigned)t, (unsigned)C) + (unsigned)ei[k]);
      S.ra[k] = at_byte(ab, o), S.rb[k] = at_byte(bb, o), S.rx[k] = at_byte(xb, o);
      S.rp[k] = at_byte(ab, t > 0 ? o - 4u * (unsigned)C : o);
      if (addb) S.re[k] = at_byte(addb, o);
    }
  };
  auto scales = [&](const Stage& S, int t0, int buf) {  // first wave: the stage's per-frame powers of two
    if (tid < TS) {
      const int t = t0 + tid;
      const double lb = S.q_mb + (double)S.q_eb;
      const float eg = (float)(S.q_ma + (double)S.q_ea + lb - z2);
      float ex = WFL_NEG_INF;
      if (t > 0) ex = (float)(S.q_mp + (double)S.q_ep + (double)S.q_m2 + lb - z2);
      sc[buf][tid][0] = __builtin_amdgcn_exp2f(0.5f * eg);
      sc[buf][tid][1] = __builtin_amdgcn_exp2f(0.5f * ex);
      sc[buf][tid][2] = S.q_m2;
    }
  };
  // this wave's block of tiles
  const int wave = tid >> 6, lane = tid & 63;
  const int mi0 = (wave >> 1) * H, nj0 = (wave & 1) * H;
  const int nmi = max(0, min(H, CPT - mi0)), nnj = max(0, min(H, CPT - nj0));  // (wave-uniform)
  const int lrow = lane >> 4, lcol = lane & 15;
  mfma_v4f acc[H][H];
#pragma unroll
  for (int i = 0; i < H; ++i)
#pragma unroll
    for (int j = 0; j < H; ++j) acc[i][j] = mfma_v4f{0.f, 0.f, 0.f, 0.f};
  // one stage: the staged tiles of frames [t0, t0 + TS) from S's registers, then the products; S is refilled with the
  // stage two further on (and its scale words computed) while the other set's stage runs
  auto stage = [&](Stage& S, int t0, int buf) {
    __syncthreads();  // the previous stage's products are done with A / U; sc[buf] is visible
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (er[k] < 0) continue;
      const int r = er[k], i = ei[k], t = t0 + r;
      float uu = 0.f, aa = 0.f;
      if (t < t_end) {
        const float hg = sc[buf][r][0], hx = sc[buf][r][1];
        const float g = (S.ra[k] * hg) * (S.rb[k] * hg);
        if (dxb)
          *reinterpret_cast<float*>(reinterpret_cast<char*>(dxb) + 4u * (__umul24((unsigned)t, (unsigned)C) + (unsigned)i)) =
              (addb ? g0 * S.re[k] : 0.f) + cf * g;
        if (partial) {
          if (t > 0) {
            const float e = __builtin_amdgcn_exp2f(fmaf(nan_to_neg(S.rx[k]), kLog2e, wr2[i]) - sc[buf][r][2]);
            uu = e * S.rb[k] * hx * cw;
            aa = S.rp[k] * hx;
          } else {
            s0[i] = g * cw;
          }
        }
      }
      U[r][i] = uu, A[r][i] = aa;  // (rows beyond the stage's last frame: zeros -- they take part in the products)
    }
    __syncthreads();
    issue(S, t0 + 2 * TS);  // in flight during this stage's products AND the whole next stage
    if (partial) {
#pragma unroll
      for (int g = 0; g < TS / 4; ++g) {
        float fu[H], fa[H];
#pragma unroll
        for (int i = 0; i < H; ++i) fu[i] = i < nmi ? U[4 * g + lrow][16 * (mi0 + i) + lcol] : 0.f;
#pragma unroll
        for (int j = 0; j < H; ++j) fa[j] = j < nnj ? A[4 * g + lrow][16 * (nj0 + j) + lcol] : 0.f;
#pragma unroll
        for (int i = 0; i < H; ++i)
#pragma unroll
          for (int j = 0; j < H; ++j)
            if (i < nmi && j < nnj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[i], fa[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  // sc[] has one slot per register set: X's scale words are rewritten (for X's NEXT stage) while Y's stage runs
  Stage X, Y;
  issue(X, t_begin);
  issue(Y, t_begin + TS);
  scales(X, t_begin, 0);
  for (int t0 = t_begin; t0 < t_end; t0 += 2 * TS) {
    scales(Y, t0 + TS, 1);      // (Y's words arrived long ago; visible behind stage X's first barrier... its second)
    stage(X, t0, 0);
    if (t0 + TS < t_end) {
      scales(X, t0 + 2 * TS, 0);  // X was refilled inside stage(X): its words are waited for here, a stage later
      stage(Y, t0 + TS, 1);
    }
  }
  if (partial) {
    __syncthreads();
    float* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (int64_t)(C + 1) * C;
    for (int i = tid; i < C; i += NT) dst[i] = s0[i];
    // accumulator v of tile (mi, nj) at lane l: state i = 16 mi + 4 (l / 16) + v, previous state j = 16 nj + l % 16
#pragma unroll
    for (int i = 0; i < H; ++i)
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (!(i < nmi && j < nnj)) continue;
        const int sj = 16 * (nj0 + j) + lcol;
        float wv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) wv[v] = W[(1 + min(16 * (mi0 + i) + 4 * lrow + v, C - 1)) * C + min(sj, C - 1)];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int si = 16 * (mi0 + i) + 4 * lrow + v;
          const float p = __builtin_amdgcn_exp2f(fmaf(wv[v], kLog2e, -wr2[min(si, CPP - 1)]));
          if (si < C && sj < C) dst[(1 + si) * C + sj] = acc[i][j][v] * p;
        }
      }
  }
}

// frames per LDS stage of dense_fast_grad_kernel: 16 where a thread stages few elements per frame (wide matrices:
// 192 / 256 threads), 8 for the one-wave workgroups of the small ones (16 frames would double their operand registers).
// cfg3 (C = 100): 128 -> 122 us with 16; more workgroups per utterance than 512 / B lose (768: 160 us, 1024: 155 us).
template <int CP>
constexpr int grad_stage() { return CP > 128 ? 8 : CP >= 104 ? 16 : 8; }  // (beyond 128 classes: up to nine waves per workgroup, fewer registers each)
static int dense_chunks(int B, int T) {
  constexpr int target = 512;  // workgroups the gradient launch aims at (768: 160 us, 1024: 155 us against 122 at cfg3)
  return std::max(1, std::min(T, (target + B - 1) / B));
}

#include "dense_wide.h"

}  // namespace wfl

using namespace wfl;

extern "C" {

constexpr int kDenseMaxClasses = 16384;  // (wide path: 2 x 4 C^2 bytes of workspace -- 2 GB here; an int32 index limit, not a design one)

// does the (C+1) x C matrix of the log-domain / Viterbi kernels fit the LDS?  (otherwise: dense_wide.h)
static bool dense_on_chip(int C) { return dense_chain_lds(C, C | 1) <= (size_t)kLdsBytes; }

static int dense_check(const float* x, const float* W, int B, int T, int C, const char* who) {
  if (!x || !W || B <= 0 || T <= 0 || C <= 0) {
    set_error("%s: bad arguments (B=%d T=%d C=%d)", who, B, T, C);
    return WFL_ERR_INVALID;
  }
  if (C > kDenseMaxClasses) {
    set_error("%s: C=%d classes (limit %d)", who, C, kDenseMaxClasses);
    return WFL_ERR_UNSUPPORTED;
  }
  return WFL_OK;
}

// Largest number of classes the dense-transition entry points take.  Up to wfl_dense_on_chip_classes() the
// transition matrix is private to a workgroup (LDS / registers); beyond, the frame update of the whole batch is a
// tiled matrix product with the matrix streamed from L2 (dense_wide.h) -- asg.py:198-199 has no limit.
extern "C" int wfl_dense_max_classes(void) { return kDenseMaxClasses; }
// smallest instantiated padded class count >= C (0: no fast path)
static int dense_fast_cp(int C) { return C <= 32 ? 32 : C <= 64 ? 64 : C <= 104 ? 104 : C <= 128 ? 128 : C <= 160 ? 160 : C <= 192 ? 192 : 0; }
// Log semiring: a workgroup keeps the matrix to itself where the register-resident probability-domain sweeps exist -- up to
// 192 classes since round 5 (five / six chain waves of 32 states beyond 128: a lane holds up to 96 factors of P; the
// LDS-resident log-domain kernels serve what those flag, they fit up to 195).  Beyond, the frame of the whole batch is
// dense_wide.h's product on the matrix cores, one launch per frame.  N = 150, B = 128, T = 1000: 0.88 ms per step with
// the register-resident sweeps against 16.7 ms per-frame launches and 33.6 ms with one LDS-resident log-domain workgroup
// per utterance (profiles/r05_asg_129_to_200_classes.txt; Viterbi: dense_viterbi_sweep_kernel, registers up to 256 classes).
static bool dense_log_on_chip(int C) { return dense_fast_cp(C) != 0; }
extern "C" int wfl_dense_on_chip_classes(void) {  // (the log semiring's limit: what sizes wfl_dense_workspace)
  int c = 1;
  while (dense_log_on_chip(c + 1)) ++c;
  return c;
}

// Where a field of the opaque workspace lies (diagnostics / tests: engine.dense_flagged reads the per-utterance flags
// through this instead of restating dense_ws_carve).
extern "C" int wfl_dense_workspace_field(int B, int T, int field, int64_t* offset_bytes, int64_t* length_bytes) {
  if (B <= 0 || T <= 0 || !offset_bytes || !length_bytes) {
    set_error("dense_workspace_field: bad arguments");
    return WFL_ERR_INVALID;
  }
  const DenseWs w = dense_ws_carve(nullptr, B, T);
  switch (field) {
    case WFL_DENSE_WS_FLAGS:  // int32 [B][2]: non-zero = handed to the log-domain kernels (forward, backward sweep)
      *offset_bytes = (int64_t)((char*)w.flag - (char*)nullptr), *length_bytes = (int64_t)8 * B;
      return WFL_OK;
    default:
      set_error("dense_workspace_field: unknown field %d", field);
      return WFL_ERR_INVALID;
  }
}

int wfl_dense_forward(const float* x, const float* W, int B, int T, int C, int semiring, float* alpha, float* beta,
                      int32_t* bptr, float* logz, void* ws, void* stream) {
  return wfl_dense_forward_parts(x, W, B, T, C, semiring, alpha, beta, bptr, logz, ws, WFL_DENSE_ALL, stream);
}

int wfl_dense_forward_parts(const float* x, const float* W, int B, int T, int C, int semiring, float* alpha, float* beta,
                            int32_t* bptr, float* logz, void* ws, int parts, void* stream) {
  if (int rc = dense_check(x, W, B, T, C, "dense_forward")) return rc;
  const bool main_part = parts & WFL_DENSE_MAIN, repair_part = parts & WFL_DENSE_REPAIR;
  if (!alpha) {
    set_error("dense_forward: alpha is required");
    return WFL_ERR_INVALID;
  }
  const bool wide = semiring == WFL_SEMIRING_LOG ? !dense_log_on_chip(C) : !dense_on_chip(C);
  if (wide || semiring != WFL_SEMIRING_LOG) {
    if (!main_part) return WFL_OK;  // (one piece: it goes with the main part)
  }
  if (wide) {
    if (semiring == WFL_SEMIRING_LOG ? (!ws || !logz) : !bptr) {
      set_error("dense_forward: missing buffers (log: logz + workspace, tropical: back-pointers)");
      return WFL_ERR_INVALID;
    }
    const int rc = wide_forward(x, W, B, T, C, semiring, alpha, beta, bptr, logz, ws, (hipStream_t)stream);
    WFL_LAUNCH_CHECK();
    return rc;
  }
  const int ldw = C | 1;
  const size_t lds = dense_chain_lds(C, ldw);
  hipStream_t st = (hipStream_t)stream;
  if (semiring == WFL_SEMIRING_LOG) {
    if (!ws || !logz) {
      set_error("dense_forward: the log semiring needs logz and a workspace (wfl_dense_workspace)");
      return WFL_ERR_INVALID;
    }
    const int cp = dense_fast_cp(C);
    const DenseWs w = dense_ws_carve(ws, B, T);
    const dim3 grid((unsigned)B, beta ? 2u : 1u);
    // (one 8-byte load per emission row where the rows are 8-byte aligned: see the helper wave)
    const bool pair = (C & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0;
#define WFL_FAST_CHAIN(CP)                                                                                              \
  do {                                                                                                                  \
    if (pair)                                                                                                           \
      hipLaunchKernelGGL((dense_fast_chain_kernel<CP, true>), grid, dim3(dense_threads<CP>()), 0, st, x, W, T, C, ws, B, alpha, \
                         beta, logz);                                                                                   \
    else                                                                                                                \
      hipLaunchKernelGGL((dense_fast_chain_kernel<CP, false>), grid, dim3(dense_threads<CP>()), 0, st, x, W, T, C, ws, B, \
                         alpha, beta, logz);                                                                            \
  } while (0)
    if (main_part) {
      if (cp == 32)
        WFL_FAST_CHAIN(32);
      else if (cp == 64)
        WFL_FAST_CHAIN(64);
      else if (cp == 104)
        WFL_FAST_CHAIN(104);
      else if (cp == 128)
        WFL_FAST_CHAIN(128);
      else if (cp == 160)
        WFL_FAST_CHAIN(160);
      else if (cp == 192)
        WFL_FAST_CHAIN(192);
      WFL_LAUNCH_CHECK();
      if (!cp)  // no fast path for this C: every utterance is served by the log-domain kernels
        WFL_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)w.flag, 1, (size_t)2 * B, st));
    }
#undef WFL_FAST_CHAIN
    if (!repair_part) return WFL_OK;
    // log-domain sweep: everything when there is no fast path, otherwise only flagged utterances
    auto k = dense_chain_kernel<WFL_SEMIRING_LOG>;
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)k, (int)lds));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, W, T, C, ldw, alpha, beta, (int32_t*)nullptr, logz,
                       cp ? (const int32_t*)w.flag : (const int32_t*)nullptr, w.M, w.z2);
  } else {
    if (!bptr) {
      set_error("dense_forward: tropical semiring needs a back-pointer buffer");
      return WFL_ERR_INVALID;
    }
    auto k = dense_chain_kernel<WFL_SEMIRING_TROPICAL>;
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)k, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)B, 1u), dim3(256), lds, st, x, W, T, C, ldw, alpha, (float*)nullptr, bptr,
                       logz, (const int32_t*)nullptr, (double*)nullptr, (double*)nullptr);
  }
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_dense_workspace(int B, int T, int C, int64_t* partial_elems, int64_t* ws_bytes) {
  if (B <= 0 || T <= 0 || C <= 0) {
    set_error("dense_workspace: bad arguments");
    return WFL_ERR_INVALID;
  }
  if (!dense_log_on_chip(C)) {
    if (partial_elems) *partial_elems = (int64_t)wide_split(C) * C * C;
    if (ws_bytes) *ws_bytes = (int64_t)wide_ws_bytes(B, T, C);
    return WFL_OK;
  }
  if (partial_elems) *partial_elems = (int64_t)B * dense_chunks(B, T) * (int64_t)(C + 1) * C;
  if (ws_bytes) *ws_bytes = (int64_t)dense_ws_bytes(B, T);
  return WFL_OK;
}

int wfl_dense_grad(const float* x, const float* W, int B, int T, int C, const float* alpha, const float* beta,
                   const float* logz, const float* coef, const float* coef_w, const float* gout, int accumulate,
                   const float* addend, const float* dW_addend, float* dx, float* dW, float* dW_partial, const void* ws,
                   void* stream) {
  return wfl_dense_grad_parts(x, W, B, T, C, alpha, beta, logz, coef, coef_w, gout, accumulate, addend, dW_addend, dx, dW,
                              dW_partial, ws, WFL_DENSE_ALL, stream);
}

int wfl_dense_grad_parts(const float* x, const float* W, int B, int T, int C, const float* alpha, const float* beta,
                         const float* logz, const float* coef, const float* coef_w, const float* gout, int accumulate,
                         const float* addend, const float* dW_addend, float* dx, float* dW, float* dW_partial,
                         const void* ws, int parts, void* stream) {
  if (int rc = dense_check(x, W, B, T, C, "dense_grad")) return rc;
  const bool main_part = parts & WFL_DENSE_MAIN, repair_part = parts & WFL_DENSE_REPAIR, reduce_part = parts & WFL_DENSE_REDUCE;
  if (!alpha || !beta || !logz || !ws || (!dx && !dW) || (dW && !dW_partial)) {
    set_error("dense_grad: missing buffers");
    return WFL_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!dense_log_on_chip(C)) {
    if (!main_part) return WFL_OK;  // (one piece: it goes with the main part)
    const int rc = wide_grad(x, B, T, C, alpha, beta, logz, coef, coef_w, gout, accumulate, addend, dW_addend, dx, dW,
                             dW_partial, ws, st);
    WFL_LAUNCH_CHECK();
    return rc;
  }
  const int chunks = dense_chunks(B, T);
  const int rows = (T + chunks - 1) / chunks;
  const int np = (C * C + 255) / 256;
  const size_t lds = 8 * (size_t)C + 64;
  dim3 grid((unsigned)chunks, (unsigned)B);
  float* part = dW ? dW_partial : nullptr;
  const int cp = dense_fast_cp(C);
  const DenseWs gws = dense_ws_carve(const_cast<void*>(ws), B, T);
  const int32_t* flags = cp ? gws.flag : nullptr;
#define WFL_FAST_GRAD(CP)                                                                                       \
  hipLaunchKernelGGL((dense_fast_grad_kernel<CP, grad_stage<CP>()>), grid, dim3(((CP / 8) * (CP / 8) + 63) / 64 * 64), 0, st, \
                     x, W, T, C, B, alpha, beta, ws, coef, coef_w, gout, accumulate, addend, dx, part, rows)
  constexpr bool use_mfma = true;  // (the vector-pipe kernel serves the calls without a transition gradient and 32 / 160 / 192 classes)
#define WFL_MFMA_GRAD(CP)                                                                                       \
  hipLaunchKernelGGL((dense_mfma_grad_kernel<CP, 16>), grid, dim3(256), 0, st, x, W, T, C, B, alpha, beta, ws, coef, coef_w, \
                     gout, accumulate, addend, dx, part, rows)
  if (!main_part) {
  } else if (cp == 32)
    WFL_FAST_GRAD(32);
  else if (cp == 64) {
    if (use_mfma && part && !accumulate) WFL_MFMA_GRAD(64); else WFL_FAST_GRAD(64);
  } else if (cp == 104) {
    if (use_mfma && part && !accumulate) WFL_MFMA_GRAD(104); else WFL_FAST_GRAD(104);
  } else if (cp == 128) {
    if (use_mfma && part && !accumulate) WFL_MFMA_GRAD(128); else WFL_FAST_GRAD(128);
  } else if (cp == 160) {  // (beyond 128 classes: the 8 x 8 register tiles of the vector pipe -- (CP / 8)^2 threads)
    WFL_FAST_GRAD(160);
  } else if (cp == 192) {
    WFL_FAST_GRAD(192);
  }
#undef WFL_MFMA_GRAD
#undef WFL_FAST_GRAD
  WFL_LAUNCH_CHECK();
  // log-domain gradient for what the fast sweeps did not serve
#define WFL_DENSE_GRAD(NP)                                                                                 \
  hipLaunchKernelGGL(dense_grad_kernel<NP>, grid, dim3(256), lds, st, x, W, T, C, alpha, beta, logz, coef, \
                     coef_w, gout, accumulate, addend, dx, part, rows, flags, gws.M, gws.z2)
  if (!repair_part) {
  } else if (np <= 4)
    WFL_DENSE_GRAD(4);
  else if (np <= 16)
    WFL_DENSE_GRAD(16);
  else if (np <= 40)
    WFL_DENSE_GRAD(40);
  else if (!dW)
    WFL_DENSE_GRAD(4);  // emission gradient only: the pair accumulators are unused
  else
    WFL_DENSE_GRAD(64);  // (C > 128: several passes over the frames, 16384 transition pairs each)
#undef WFL_DENSE_GRAD
  WFL_LAUNCH_CHECK();
  if (dW && reduce_part) {
    const int n = (C + 1) * C;
    hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, dW_partial, B * chunks,
                       n, dW, accumulate, dW_addend, gout);
    WFL_LAUNCH_CHECK();
  }
  return WFL_OK;
}

int wfl_dense_viterbi(const float* x, const float* W, int B, int T, int C, float* alpha, int32_t* bptr, int32_t* path,
                      void* stream) {
  if (!path) {
    set_error("dense_viterbi: path is required");
    return WFL_ERR_INVALID;
  }
  if (C <= 256) {
    // transition rows in registers, no back-pointers (`bptr` is not written): dense_viterbi_sweep_kernel
    if (int rc = dense_check(x, W, B, T, C, "dense_viterbi")) return rc;
    if (!alpha) {
      set_error("dense_viterbi: alpha is required");
      return WFL_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    // four lanes per state up to 128 classes (a quarter of the row each: half the per-frame work of a lane), two beyond
    const int lanes = C <= 128 ? 4 : 2;
    const unsigned nt = (unsigned)std::min(512, ((lanes * C + 63) / 64) * 64);
#define WFL_VIT_SWEEP(SPAN, R) \
  hipLaunchKernelGGL((dense_viterbi_sweep_kernel<SPAN, R>), dim3((unsigned)B), dim3(nt), 0, st, x, W, T, C, alpha)
    if (C <= 32)
      WFL_VIT_SWEEP(8, 4);
    else if (C <= 64)
      WFL_VIT_SWEEP(16, 4);
    else if (C <= 112)
      WFL_VIT_SWEEP(28, 4);
    else if (C <= 128)
      WFL_VIT_SWEEP(32, 4);
    else if (C <= 192)
      WFL_VIT_SWEEP(96, 2);
    else
      WFL_VIT_SWEEP(128, 2);  // (256 VGPRs, 12 B of scratch; four lanes per state on 1024 threads -- 128 VGPRs each -- spill more and
                              // take 2.08 instead of 1.38 ms at C = 200)
#undef WFL_VIT_SWEEP
    WFL_LAUNCH_CHECK();
    const size_t fixed = 2 * ((size_t)kVitChunkFloats + 256) * 4 + 1024 * 4 + 16;
    const bool wlds = fixed + (size_t)C * C * 4 <= (size_t)kLdsBytes;
    const size_t lds = fixed + (wlds ? (size_t)C * C * 4 : 0);
    auto launch_bt = [&](auto k) -> int {
      if (lds > 48 * 1024) WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)k, (int)lds));
      hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(256), lds, st, alpha, W, B, T, C, path);
      return WFL_OK;
    };
    const int nj = (C + 63) / 64;
    int rc = WFL_OK;
    if (wlds)
      rc = nj == 1 ? launch_bt(dense_viterbi_backtrace_kernel<true, 1>) : nj == 2 ? launch_bt(dense_viterbi_backtrace_kernel<true, 2>)
           : nj == 3 ? launch_bt(dense_viterbi_backtrace_kernel<true, 3>) : launch_bt(dense_viterbi_backtrace_kernel<true, 4>);
    else
      rc = nj == 3 ? launch_bt(dense_viterbi_backtrace_kernel<false, 3>) : launch_bt(dense_viterbi_backtrace_kernel<false, 4>);
    if (rc) return rc;
    WFL_LAUNCH_CHECK();
    return WFL_OK;
  }
  {
    // beyond 256 classes: one tiled max-plus launch per frame for the whole batch (csrc/dense_wide.h), no back-pointers
    // either (`bptr` is not written), then the walk that re-derives the ones it follows
    if (int rc = dense_check(x, W, B, T, C, "dense_viterbi")) return rc;
    if (!alpha) {
      set_error("dense_viterbi: alpha is required");
      return WFL_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wide_viterbi_first_max_kernel, dim3((unsigned)(((int64_t)B * C + 255) / 256)), dim3(256), 0, st, x, W, B, T,
                       C, alpha);
    const dim3 grid((unsigned)((C + 15) / 16), (unsigned)((B + 15) / 16));
    const bool vec4 = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(alpha)) & 15) == 0;
    static const bool resident_off = [] {
      const char* e = getenv("WFL_DENSE_WIDE_RESIDENT");
      return e && atoi(e) == 0;
    }();
    if (wide_resident_rows(C) == 5 && T >= 2 && !resident_off) {
      // up to 320 classes: the frames of an utterance inside one workgroup, the matrix in its registers (same vectors)
      hipLaunchKernelGGL(wide_resident_viterbi_kernel<5>, dim3((unsigned)B), dim3(1024), 0, st, x, W, B, T, C, alpha);
    } else {
      for (int t = 1; t < T; ++t) {
        if (vec4)
          hipLaunchKernelGGL(wide_viterbi_max_kernel<true>, grid, dim3(64 * kVitWideWaves), 0, st, x, W, B, T, C, t, alpha);
        else
          hipLaunchKernelGGL(wide_viterbi_max_kernel<false>, grid, dim3(64 * kVitWideWaves), 0, st, x, W, B, T, C, t, alpha);
      }
    }
    hipLaunchKernelGGL(dense_viterbi_walk_kernel, dim3((unsigned)B), dim3(64), 0, st, alpha, W, B, T, C, path);
    WFL_LAUNCH_CHECK();
    return WFL_OK;
  }
}

}  // extern "C"
