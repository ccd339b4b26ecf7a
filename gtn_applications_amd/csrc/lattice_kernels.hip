// Generic lattice engine for gfx950 (MI355X): time-synchronous log-/max-plus forward-backward over
// an arbitrary per-utterance acceptor A_b composed with the implicit emissions chain.
//
//   stage 1  wfl_lattice_gather   all CUs, one wave per (b,t) row: xg[b,t,k] = x[b,t,labels_b[k]]
//                                 (optionally minus the row's log-sum-exp: fused log_softmax)
//   stage 2  wfl_lattice_forward  one workgroup per (utterance, direction); arcs of A_b staged in
//                                 LDS as CSR lists; emission rows prefetched a chunk of frames
//                                 ahead; per frame one of: banded (DPP) / single-wave (ds_bpermute) /
//                                 multi-wave lean / general path with epsilon levels, cooperative
//                                 16-lane relaxation of high in-degree states
//   stage 3  wfl_lattice_grad     all CUs, tiles of frames: posteriors accumulated per (frame,
//                                 distinct label) by threads owning <= 16 arcs of one label, dense
//                                 rows streamed out through a column -> slot map (fused log_softmax
//                                 backward optional); learnable-weight grads reduced per workgroup
//
// This replaces gtn.intersect(emissions, A) + gtn.forward_score / viterbi_path + gtn.backward
// (criterions/ctc.py:49-51,78-81; asg.py:111-113; stc.py:85-87; transducer.py:283,321-325) without
// ever materialising the T*|A| composed lattice.  Memory/latency-bound DP: no MFMA by design.
#include <atomic>
#include <map>
#include <mutex>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "device_common.h"

namespace wfl {

#ifdef WFL_LIVE_STATS
// wall clock (100 MHz) at the first and last workgroup of the launches around the sweeps: [k][0] first begin, [k][1] last end
// k: 0 certificate, 1 log-domain launch, 2 gradient kernel of the rest, 3 gather
__device__ unsigned long long g_marks[8][2];
__device__ __forceinline__ unsigned long long wall_clock64_early() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}
#define WFL_MARK_BEGIN(k) if (threadIdx.x == 0) atomicMin(&g_marks[k][0], wall_clock64_early())
#define WFL_MARK_END(k) if (threadIdx.x == 0) atomicMax(&g_marks[k][1], wall_clock64_early())
// a log of launches: workgroup (0, 0) of a kernel appends {id, wall clock} (ids: scripts/live_timeline.py)
__device__ unsigned long long g_log[512][2];
__device__ unsigned int g_log_n;
#define WFL_LOG(k) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { const unsigned int i_ = atomicAdd(&g_log_n, 1u) & 511u; g_log[i_][0] = (k); g_log[i_][1] = wall_clock64_early(); }
#else
#define WFL_MARK_BEGIN(k)
#define WFL_MARK_END(k)
#define WFL_LOG(k)
#endif

struct UttView {
  int Q, A, E, K, nlev;
  int a0, e0;
  const int32_t *in_ptr, *out_ptr, *out_arc, *ein_ptr, *eout_ptr, *eout_arc;
  const int32_t *arc_src, *arc_dst, *arc_slot, *arc_lab, *arc_wid, *arc_orig;
  const int32_t *eps_src, *eps_dst, *eps_wid, *eps_orig, *labels, *lvl_ptr, *slot_ptr, *slot_arc;
  const float *arc_w, *eps_w, *start_w, *accept_w;
  int64_t ab_base, xg_base;
};

__device__ __forceinline__ UttView make_view(const wfl_lattice_desc& d, const int32_t* ints, const float* floats,
                                             int b, int T) {
  UttView v;
  const int bb = d.shared ? 0 : b;
  const int s0 = ints[d.state_off + bb];
  v.Q = ints[d.state_off + bb + 1] - s0;
  v.a0 = ints[d.arc_off + bb];
  v.A = ints[d.arc_off + bb + 1] - v.a0;
  v.e0 = ints[d.eps_off + bb];
  v.E = ints[d.eps_off + bb + 1] - v.e0;
  const int l0 = ints[d.lab_off + bb];
  v.K = ints[d.lab_off + bb + 1] - l0;
  const int lv0 = ints[d.lvl_off + bb];
  v.nlev = ints[d.lvl_off + bb + 1] - lv0 - 1;
  v.in_ptr = ints + d.in_ptr + s0 + bb;
  v.out_ptr = ints + d.out_ptr + s0 + bb;
  v.ein_ptr = ints + d.ein_ptr + s0 + bb;
  v.eout_ptr = ints + d.eout_ptr + s0 + bb;
  v.out_arc = ints + d.out_arc + v.a0;
  v.eout_arc = ints + d.eout_arc + v.e0;
  v.arc_src = ints + d.arc_src + v.a0, v.arc_dst = ints + d.arc_dst + v.a0;
  v.arc_slot = ints + d.arc_slot + v.a0, v.arc_lab = ints + d.arc_lab + v.a0;
  v.arc_wid = ints + d.arc_wid + v.a0, v.arc_orig = ints + d.arc_orig + v.a0;
  v.eps_src = ints + d.eps_src + v.e0, v.eps_dst = ints + d.eps_dst + v.e0;
  v.eps_wid = ints + d.eps_wid + v.e0, v.eps_orig = ints + d.eps_orig + v.e0;
  v.labels = ints + d.labels + l0;
  v.slot_ptr = ints + d.slot_ptr + l0 + bb, v.slot_arc = ints + d.slot_arc + v.a0;
  v.lvl_ptr = ints + d.lvl_ptr + lv0;
  v.arc_w = floats + d.arc_w + v.a0, v.eps_w = floats + d.eps_w + v.e0;
  v.start_w = floats + d.start_w + s0, v.accept_w = floats + d.accept_w + s0;
  v.ab_base = d.shared ? (int64_t)b * (T + 1) * v.Q : (int64_t)(T + 1) * s0;
  v.xg_base = (int64_t)b * T * d.max_labels;
  return v;
}

// ------------------------------------------------------------------------------------------------
// stage 1: gather
// ------------------------------------------------------------------------------------------------
// Probability-domain copy of a gathered row for the fp64 chains (run_chain_prob): factors
// fg[k] = 2^((xg[k] - r) * log2 e) <= 1 relative to the row's reference r = max_k xg[k] (0 if the row is all -inf).
// One wave per row; `vmx` is the lane's running maximum of the values it wrote to `dst` (re-read from L1 here).
__device__ __forceinline__ void emit_factors(const float* dst, float* fdst, float* rdst, int K, int lane, float vmx) {
  if (!fdst) return;
  float r = wave_all_max(vmx);
  if (!(r > WFL_NEG_INF)) r = 0.f;
  for (int k = lane; k < K; k += 64) fdst[k] = __builtin_amdgcn_exp2f((dst[k] - r) * 1.4426950408889634f);
  if (lane == 0) *rdst = r;
}

__global__ void __launch_bounds__(256) gather_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints,
                                                      const float* __restrict__ x, int T, int C,
                                                      float* __restrict__ xg, float* __restrict__ row_lse,
                                                      float* __restrict__ fg, float* __restrict__ rmax) {
#ifndef WFL_GATHER_PRIO
#define WFL_GATHER_PRIO 0
#endif
  if (WFL_GATHER_PRIO) __builtin_amdgcn_s_setprio(WFL_GATHER_PRIO);
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bb = d.shared ? 0 : b;
  const int l0 = ints[d.lab_off + bb];
  const int K = ints[d.lab_off + bb + 1] - l0;
  const int32_t* labels = ints + d.labels + l0;
  const int Kmax = d.max_labels;
  if (!row_lse && K <= 64) {
    // At most one label per lane and no row reduction (the ASG numerator): the lane's column is looked up once, four
    // rows are in flight per wave, and a row's values go from the register to xg, to the factors and to the row
    // maximum -- the general loop below reads labels[k], then row[labels[k]], stores, and reads the stored value
    // back for the factors: three dependent round trips per row and wave.
    constexpr int RU = 4;
    const int col = lane < K ? labels[lane] : -1;
    const float* xb = x + (int64_t)b * T * C;
    float* gb = xg + (int64_t)b * T * Kmax;
    float* fb = fg ? fg + (int64_t)b * T * Kmax : nullptr;
    for (int t0 = (blockIdx.x * 4 + wave) * RU; t0 < T; t0 += gridDim.x * 4 * RU) {
      float v[RU];
#pragma unroll
      for (int q = 0; q < RU; ++q) v[q] = xb[(int64_t)min(t0 + q, T - 1) * C + max(col, 0)];
#pragma unroll
      for (int q = 0; q < RU; ++q) {
        const int t = t0 + q;
        if (t < T) {  // (uniform)
          const float val = col >= 0 ? nan_to_neg(v[q]) : WFL_NEG_INF;
          if (col >= 0) gb[(int64_t)t * Kmax + lane] = val;
          if (fb) {  // emit_factors, from the register
            float r = wave_all_max(val);
            if (!(r > WFL_NEG_INF)) r = 0.f;
            if (col >= 0) fb[(int64_t)t * Kmax + lane] = __builtin_amdgcn_exp2f((val - r) * 1.4426950408889634f);
            if (lane == 0) rmax[(int64_t)b * T + t] = r;
          }
        }
      }
    }
    return;
  }
  for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
    const float* row = x + ((int64_t)b * T + t) * C;
    float lse = 0.f;
    if (row_lse) {  // fused log_softmax (ctc.py:107, transducer.py:186-187)
      float m = WFL_NEG_INF;
      for (int c = lane; c < C; c += 64) m = fmaxf(m, nan_to_neg(row[c]));
      m = wave_max(m);
      float s = 0.f;
      if (m > WFL_NEG_INF)
        for (int c = lane; c < C; c += 64) s += fast_exp(nan_to_neg(row[c]) - m);
      s = wave_sum(s);
      lse = (m > WFL_NEG_INF) ? m + fast_log(s) : WFL_NEG_INF;
      if (lane == 0) row_lse[(int64_t)b * T + t] = lse;
    }
    float* dst = xg + ((int64_t)b * T + t) * Kmax;
    float vmx = WFL_NEG_INF;
    for (int k = lane; k < K; k += 64) {
      const float v = nan_to_neg(row[labels[k]]) - lse;
      dst[k] = v;
      vmx = fmaxf(vmx, v);
    }
    emit_factors(dst, fg ? fg + ((int64_t)b * T + t) * Kmax : nullptr, rmax ? rmax + (int64_t)b * T + t : nullptr, K, lane,
                 vmx);
  }
}

// The fused-log_softmax gather for C <= 64 * NV classes: the row is read ONCE, into registers (NV loads in flight per
// lane; the generic kernel above walks it three times with one load in flight) and reduced with DPP.  One wave per row,
// RU rows per iteration for the narrow cases.
// Round 6: the label columns are picked out of an LDS copy of the row, through an LDS copy of the utterance's label
// list, and the factors are formed from the values still in registers.  Until then a row cost FOUR dependent round
// trips -- the row, labels[k], row[labels[k]], and the stored value read back for the factors -- and the second read of
// the row missed: 32 waves per CU x 4 KB rows are four times the L1, 32 CUs of them the XCD's whole L2, so at the
// Transducer benchmark the launch fetched 272 MB for 205 MB of emissions (profiles/r05_pmc_traffic.json).  Now one round
// trip per row and every byte of x once.
template <int NV, int RU>
__global__ void __launch_bounds__(256) gather_lse_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints,
                                                          const float* __restrict__ x, int T, int C,
                                                          float* __restrict__ xg, float* __restrict__ row_lse,
                                                          float* __restrict__ fg, float* __restrict__ rmax) {
  WFL_LOG(1);
  __shared__ float srow[4][64 * NV];   // a wave's current row
  __shared__ int16_t slab[1024];       // the utterance's label columns (wfl_lattice_forward: at most 1024 per utterance)
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bb = d.shared ? 0 : b;
  const int l0 = ints[d.lab_off + bb];
  const int K = ints[d.lab_off + bb + 1] - l0;
  const int32_t* labels = ints + d.labels + l0;
  const int Kmax = d.max_labels;
  const bool lds_labels = K <= 1024 && C <= 32767;
  if (lds_labels)
    for (int k = threadIdx.x; k < K; k += 256) slab[k] = (int16_t)labels[k];
  __syncthreads();
  float* mine = srow[wave];
  constexpr int KR = 2;  // label slots per lane kept in registers for the factors (K <= 128: every recipe's case)
  for (int t0 = (blockIdx.x * 4 + wave) * RU; t0 < T; t0 += gridDim.x * 4 * RU) {
    float v[RU][NV];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float* row = x + ((int64_t)b * T + min(t0 + u, T - 1)) * C;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[u][i] = row[min(c, C - 1)];  // (clamped, no test: the NV loads of a row go out back to back)
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (t0 + u >= T) break;
      float m = WFL_NEG_INF;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[u][i] = lane + 64 * i < C ? nan_to_neg(v[u][i]) : WFL_NEG_INF, m = fmaxf(m, v[u][i]);
      m = wave_all_max(m);
      float sum = 0.f;
      if (m > WFL_NEG_INF) {
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += fast_exp(v[u][i] - m);
      }
      sum = wave_all_sum(sum);
      const float lse = (m > WFL_NEG_INF) ? m + fast_log(sum) : WFL_NEG_INF;
      const int64_t r = (int64_t)b * T + t0 + u;
      if (lane == 0) row_lse[r] = lse;
      float* dst = xg + r * Kmax;
      if (lds_labels && K <= 64 * KR) {
        // the row through LDS (the wave's own slot: its DS operations execute in order, no barrier), the labels from LDS
#pragma unroll
        for (int i = 0; i < NV; ++i) mine[lane + 64 * i] = v[u][i];
        float g[KR];
        float vmx = WFL_NEG_INF;
#pragma unroll
        for (int j = 0; j < KR; ++j) {
          const int k = lane + 64 * j;
          g[j] = k < K ? mine[slab[k]] - lse : WFL_NEG_INF;   // (the LDS copy is NaN-cleaned: what the old `nan_to_neg(row[label]) - lse` gave)
          if (k < K) dst[k] = g[j];
          vmx = fmaxf(vmx, k < K ? g[j] : WFL_NEG_INF);
        }
        if (fg) {  // emit_factors, from the registers
          float rr = wave_all_max(vmx);
          if (!(rr > WFL_NEG_INF)) rr = 0.f;
          float* fdst = fg + r * Kmax;
#pragma unroll
          for (int j = 0; j < KR; ++j) {
            const int k = lane + 64 * j;
            if (k < K) fdst[k] = __builtin_amdgcn_exp2f((g[j] - rr) * 1.4426950408889634f);
          }
          if (lane == 0) rmax[r] = rr;
        }
        continue;
      }
      const float* row = x + r * C;
      float vmx = WFL_NEG_INF;
      for (int k = lane; k < K; k += 64) {
        const float vv = nan_to_neg(row[labels[k]]) - lse;
        dst[k] = vv;
        vmx = fmaxf(vmx, vv);
      }
      emit_factors(dst, fg ? fg + r * Kmax : nullptr, rmax ? rmax + r : nullptr, K, lane, vmx);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stage 2: chains
// ------------------------------------------------------------------------------------------------
struct ChainLds {
  int2* arcs;   // [A] {other_state | slot << 16, weight bits}
  int2* eps;    // [E] {other_state, weight bits}
  int* ptr;     // [Q+1]
  int* eptr;    // [Q+1]
  float* buf0;  // [Q]
  float* buf1;  // [Q]
  float* rows;  // [2][R][Kmax] emission rows of the current / next chunk of R frames
  float* red;   // [64]
  int* lvl;     // [nlev+1]
  int* heavy;   // [Q] states with more than kHeavyDeg in-arcs (cooperative relaxation), heavy[Q] = their count
};

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}

// ---- state values of the general sweep (run_chain) -------------------------------------------------------------
// Tropical semiring: float (sums and comparisons of the caller's floats: ties stay ties).  Log semiring: DOUBLE -- in LDS
// and in HBM.  The states that carry the posteriors lie 20-50 nats below the frame's largest state (measured on the
// reference's n-gram benchmark graphs), where a float32 resolves 2-4e-6: float state vectors relative to the frame's
// maximum put 3e-5 .. 5e-5 on the posteriors at T = 250 and 1.0 .. 2.5e-4 at T = 1000 whatever the renormalisation
// interval or the accuracy of exp / log.  With doubles a frame's error is that of its
// exponentials (float, of differences to the state's own largest term) and of one double log per state.
template <int SR>
struct ChainVal {
  using type = float;
};
template <>
struct ChainVal<WFL_SEMIRING_LOG> {
  using type = double;
};
__device__ __forceinline__ float lse_exp(float d) { return fast_exp(d); }   // d <= 0: a term relative to the largest
// (lse_log: device_common.h)

// One relaxation of a state over its labelled in-arcs (values read from `from`).  The first four
// arcs are independent LDS chains whose terms stay in registers (most states of the criteria's
// acceptors have in-degree <= 4); longer lists continue with a streaming max / rescale loop over
// LDS arcs [kt0, kt1).  `k0` is the CSR index of the first arc (for Viterbi back-pointers).
struct Arc4 {
  int other[4], slot[4];
  float w[4];
};

__device__ __forceinline__ Arc4 load_arc4(const int2* arcs, int k0, int k1) {
  Arc4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool ok = k0 + i < k1;
    const int2 a = ok ? arcs[k0 + i] : make_int2(0, __float_as_int(WFL_NEG_INF));
    r.other[i] = a.x & 0xffff, r.slot[i] = (unsigned)a.x >> 16, r.w[i] = __int_as_float(a.y);
  }
  return r;
}

template <int SR, typename VT>
__device__ __forceinline__ void relax_labelled(const ChainLds& L, const Arc4& a4, const VT* from, const float* row,
                                               int k0, int kt0, int kt1, VT& val, int& arg) {
  const VT v0 = from[a4.other[0]] + (VT)row[a4.slot[0]] + (VT)a4.w[0];
  const VT v1 = from[a4.other[1]] + (VT)row[a4.slot[1]] + (VT)a4.w[1];
  const VT v2 = from[a4.other[2]] + (VT)row[a4.slot[2]] + (VT)a4.w[2];
  const VT v3 = from[a4.other[3]] + (VT)row[a4.slot[3]] + (VT)a4.w[3];
  VT m = v0;
  int am = v0 > WFL_NEG_INF ? k0 : -1;
  if (v1 > m) m = v1, am = k0 + 1;
  if (v2 > m) m = v2, am = k0 + 2;
  if (v3 > m) m = v3, am = k0 + 3;
  auto term = [&](int k) {
    const int2 a = L.arcs[k];
    return from[a.x & 0xffff] + (VT)row[(unsigned)a.x >> 16] + (VT)__int_as_float(a.y);
  };
  if (SR == WFL_SEMIRING_LOG) {
    VT s = 0;
    if (m > WFL_NEG_INF)
      s = (VT)lse_exp((float)(v0 - m)) + (VT)lse_exp((float)(v1 - m)) + (VT)lse_exp((float)(v2 - m)) + (VT)lse_exp((float)(v3 - m));
    for (int k = kt0; k < kt1; ++k) {
      const VT v = term(k);
      if (v > m) {
        s = s * (VT)lse_exp((float)(m - v)) + 1;  // m == -inf: s is 0 and exp(-inf) = 0
        m = v;
      } else if (v > WFL_NEG_INF) {
        s += (VT)lse_exp((float)(v - m));
      }
    }
    if (m > WFL_NEG_INF) m += (VT)lse_log((double)s);
  } else {
    for (int k = kt0; k < kt1; ++k) {
      const VT v = term(k);
      if (v > m) m = v, am = k;
    }
  }
  val = m, arg = am;
}

// High in-degree states (dense n-gram transition graphs: 81 in-arcs per state in the reference's
// transducer benchmark) are relaxed cooperatively: a row of 16 lanes strides over the state's
// labelled in-arcs and merges with DPP row operations, four states per wavefront at a time.
// Tropical ties keep the lowest arc index, like the serial walk.
constexpr int kHeavyDeg = 24;

__device__ __forceinline__ float row16_max(float v) {  // every lane of the 16-lane row receives the row's maximum
  v = fmaxf(v, dpp_f32<0x111, 0xf>(WFL_NEG_INF, v));
  v = fmaxf(v, dpp_f32<0x112, 0xf>(WFL_NEG_INF, v));
  v = fmaxf(v, dpp_f32<0x114, 0xf>(WFL_NEG_INF, v));
  v = fmaxf(v, dpp_f32<0x118, 0xf>(WFL_NEG_INF, v));
  return __shfl(v, 15, 16);
}
__device__ __forceinline__ double row16_max(double v) {  // (doubles: through the cross-lane network, 8 bytes a step)
  v = fmax(v, __shfl_xor(v, 1, 16));
  v = fmax(v, __shfl_xor(v, 2, 16));
  v = fmax(v, __shfl_xor(v, 4, 16));
  v = fmax(v, __shfl_xor(v, 8, 16));
  return v;
}
__device__ __forceinline__ double row16_sum(double v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}
// the same through DPP (no LDS crossbar round trips: the shuffles above are eight dependent ds_bpermute): every lane of
// the row receives the row's sum.  quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror -- after each step the
// partial sums are uniform over twice as many lanes, whichever lane of the other group a lane reads.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double row16_sum_dpp(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ int row16_min(int v) {
  v = min(v, __shfl_xor(v, 1, 16));
  v = min(v, __shfl_xor(v, 2, 16));
  v = min(v, __shfl_xor(v, 4, 16));
  v = min(v, __shfl_xor(v, 8, 16));
  return v;
}

// epsilon closure of a state with many epsilon in-arcs (the back-off state of an n-gram graph collects
// one from every history): same 16-lane cooperation; `val` / `arg` enter with the labelled result
template <int SR, typename VT>
__device__ __forceinline__ void relax_eps_row16(const ChainLds& L, const VT* vals, int k0, int k1, int A, VT& val,
                                                int& arg) {
  const int r = threadIdx.x & 15;
  VT m = r == 0 ? val : (VT)WFL_NEG_INF, s = (r == 0 && val > WFL_NEG_INF) ? 1 : 0;
  int am = 0x7fffffff;  // (the labelled result wins ties: it is "earlier" than every epsilon arc)
  for (int k = k0 + r; k < k1; k += 16) {
    const int2 a = L.eps[k];
    const VT v = vals[a.x] + (VT)__int_as_float(a.y);
    if (SR == WFL_SEMIRING_LOG) {
      if (v > m) {
        s = s * (VT)lse_exp((float)(m - v)) + 1;
        m = v;
      } else if (v > WFL_NEG_INF) {
        s += (VT)lse_exp((float)(v - m));
      }
    } else if (v > m) {
      m = v, am = A + k;
    }
  }
  const VT mt = row16_max(m);
  if (SR == WFL_SEMIRING_LOG) {
    const double st = row16_sum(m > WFL_NEG_INF ? (double)s * (double)lse_exp((float)(m - mt)) : 0.0);
    val = mt > WFL_NEG_INF ? mt + (VT)lse_log(st) : (VT)WFL_NEG_INF;
  } else {
    // an epsilon arc replaces the labelled back-pointer only if it is strictly better than it
    const int cand = row16_min((m == mt && mt > val) ? am : 0x7fffffff);
    if (cand != 0x7fffffff) arg = cand;
    val = mt;
  }
}

// all 64 lanes of the wave must call this (rows without a state pass k0 == k1)
template <int SR, typename VT>
__device__ __forceinline__ void relax_labelled_row16(const ChainLds& L, const VT* from, const float* row, int k0,
                                                     int k1, VT& val, int& arg) {
  const int r = threadIdx.x & 15;
  VT m = WFL_NEG_INF, s = 0;
  int am = 0x7fffffff;
  auto term = [&](int k) -> VT {  // -inf past the end of the list
    if (k >= k1) return (VT)WFL_NEG_INF;
    const int2 a = L.arcs[k];
    return from[a.x & 0xffff] + (VT)row[(unsigned)a.x >> 16] + (VT)__int_as_float(a.y);
  };
  // four independent arcs per step: their LDS round trips overlap instead of queueing behind each other
  for (int k = k0 + r; k < k1; k += 64) {
    const VT v0 = term(k), v1 = term(k + 16), v2 = term(k + 32), v3 = term(k + 48);
    const VT m4 = v0 > v1 ? (v0 > v2 ? (v0 > v3 ? v0 : v3) : (v2 > v3 ? v2 : v3)) : (v1 > v2 ? (v1 > v3 ? v1 : v3) : (v2 > v3 ? v2 : v3));
    if (SR == WFL_SEMIRING_LOG) {
      if (m4 > WFL_NEG_INF) {
        const VT mn = m > m4 ? m : m4;
        s = s * (VT)lse_exp((float)(m - mn)) + (VT)lse_exp((float)(v0 - mn)) + (VT)lse_exp((float)(v1 - mn)) +
            (VT)lse_exp((float)(v2 - mn)) + (VT)lse_exp((float)(v3 - mn));
        m = mn;
      }
    } else {
      // ascending arc index: strict '>' keeps the first maximum
      if (v0 > m) m = v0, am = k;
      if (v1 > m) m = v1, am = k + 16;
      if (v2 > m) m = v2, am = k + 32;
      if (v3 > m) m = v3, am = k + 48;
    }
  }
  const VT mt = row16_max(m);
  if (SR == WFL_SEMIRING_LOG) {
    const double st = row16_sum(m > WFL_NEG_INF ? (double)s * (double)lse_exp((float)(m - mt)) : 0.0);
    val = mt > WFL_NEG_INF ? mt + (VT)lse_log(st) : (VT)WFL_NEG_INF;
    arg = -1;
  } else {
    const int cand = row16_min((m == mt && mt > WFL_NEG_INF) ? am : 0x7fffffff);
    val = mt;
    arg = cand == 0x7fffffff ? -1 : cand;
  }
}

template <int SR, typename VT>
__device__ __forceinline__ void relax_eps(const ChainLds& L, VT* vals, int q, int k0, int k1, int A, VT& val, int& arg) {
  // combine the current value of q with its epsilon arcs (other endpoints are already final)
  VT m = val;
  int am = arg;
  for (int k = k0; k < k1; ++k) {
    const int2 a = L.eps[k];
    const VT v = vals[a.x] + (VT)__int_as_float(a.y);
    if (v > m) m = v, am = A + k;
  }
  if (SR == WFL_SEMIRING_LOG) {
    if (m > WFL_NEG_INF) {
      VT s = (VT)lse_exp((float)(val - m));
      for (int k = k0; k < k1; ++k) {
        const int2 a = L.eps[k];
        s += (VT)lse_exp((float)(vals[a.x] + (VT)__int_as_float(a.y) - m));
      }
      m += (VT)lse_log((double)s);
    }
  }
  val = m, arg = am;
}

constexpr int kLeanDeg = 8;  // most arcs into (and out of) a state of an acceptor the probability-domain sweeps take
constexpr int kPre = 8;  // prefetch registers per thread: rows_per_chunk * max_labels <= kPre * blockDim

// The general sweep: any acceptor that fits LDS -- epsilon arcs in topological levels, states of any in-degree (more
// than kHeavyDeg arcs: a row of 16 lanes per state), both semirings.  What the probability-domain sweeps
// (run_chain_prob) do not take comes here: acceptors with epsilon arcs or high degrees, utterances whose certificate
// failed, the tropical semiring.  (Until round 4 this function also held four "lean" frame loops for chains and low-degree
// acceptors in the fp32 log domain; the probability-domain sweeps serve those since round 2, and what reaches this
// function on their behalf -- a certificate failure -- is rare enough for the general loop.)
template <int SR, int DIR>
__device__ void run_chain(const wfl_lattice_desc& d, const UttView& u, const ChainLds& L, int T, int rows_per_chunk,
                          const float* __restrict__ xg, const float* __restrict__ weights, float* __restrict__ out_f,
                          int32_t* __restrict__ bptr, float* __restrict__ logz, int b, double* __restrict__ offs,
                          double* __restrict__ z64) {
  using VT = typename ChainVal<SR>::type;
  const int tid = threadIdx.x, NT = blockDim.x;
  VT* const out = reinterpret_cast<VT*>(out_f);  // (indexed like the float array: u.ab_base + slot * Q + q)
  VT* const buf0 = reinterpret_cast<VT*>(L.buf0);
  VT* const buf1 = reinterpret_cast<VT*>(L.buf1);
  // Renormalisation (log semiring): at the start of every chunk of R frames the maximum of the state vector moves into
  // a double offset: offs[1 + c] is what the slots produced in chunk c are relative to (offs[0] = 0: the boundary
  // slot), and the gradient kernel adds offs_alpha + offs_beta - log Z before it exponentiates.  With double states
  // this only keeps magnitudes small for the float conversions of differences; it carries no accuracy any more.
  double cum = 0.0;
  if (SR == WFL_SEMIRING_LOG && tid == 0) offs[0] = 0.0;
  const int Q = u.Q, A = u.A, E = u.E, nlev = u.nlev, Kmax = d.max_labels;
  // ---- stage the acceptor into LDS in this direction's CSR order
  for (int k = tid; k < A; k += NT) {
    const int a = DIR == 0 ? k : u.out_arc[k];
    const int other = DIR == 0 ? u.arc_src[a] : u.arc_dst[a];
    float w = u.arc_w[a];
    const int wid = u.arc_wid[a];
    if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
    L.arcs[k] = make_int2(other | (u.arc_slot[a] << 16), __float_as_int(w));
  }
  for (int k = tid; k < E; k += NT) {
    const int e = DIR == 0 ? k : u.eout_arc[k];
    const int other = DIR == 0 ? u.eps_src[e] : u.eps_dst[e];
    float w = u.eps_w[e];
    const int wid = u.eps_wid[e];
    if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
    L.eps[k] = make_int2(other, __float_as_int(w));
  }
  for (int q = tid; q <= Q; q += NT) {
    L.ptr[q] = DIR == 0 ? u.in_ptr[q] : u.out_ptr[q];
    L.eptr[q] = DIR == 0 ? u.ein_ptr[q] : u.eout_ptr[q];
  }
  for (int l = tid; l <= nlev; l += NT) L.lvl[l] = u.lvl_ptr[l];

  // epsilon closure of `vals` for this direction; `tslot` is the time slot for back-pointers
  bool eps_heavy = false;  // set once the acceptor is staged (below)
  auto closure = [&](VT* vals, int tslot) {
    if (nlev <= 1) return;
    for (int step = 1; step < nlev; ++step) {
      const int lev = DIR == 0 ? step : nlev - 1 - step;
      lds_barrier();  // (the levels meet in LDS: not __syncthreads, which also waits for the frame's global stores)
      for (int q = L.lvl[lev] + tid; q < L.lvl[lev + 1]; q += NT) {
        if (L.eptr[q + 1] - L.eptr[q] > kHeavyDeg) continue;  // cooperative pass below
        VT v = vals[q];
        int arg = -2;
        relax_eps<SR, VT>(L, vals, q, L.eptr[q], L.eptr[q + 1], A, v, arg);
        vals[q] = v;
        if (SR == WFL_SEMIRING_TROPICAL && DIR == 0 && arg != -2) bptr[u.ab_base + (int64_t)tslot * Q + q] = arg;
      }
      if (eps_heavy) {  // states of this level with many epsilon in-arcs: one 16-lane row each
        for (int q = L.lvl[lev] + (tid >> 4); q < L.lvl[lev + 1]; q += NT >> 4) {
          const int k0 = L.eptr[q], k1 = L.eptr[q + 1];
          if (k1 - k0 <= kHeavyDeg) continue;  // uniform within the row
          VT v = vals[q];
          int arg = -2;
          relax_eps_row16<SR, VT>(L, vals, k0, k1, A, v, arg);
          if ((tid & 15) == 0) {
            vals[q] = v;
            if (SR == WFL_SEMIRING_TROPICAL && DIR == 0 && arg != -2) bptr[u.ab_base + (int64_t)tslot * Q + q] = arg;
          }
        }
      }
    }
  };

  const int t_first = DIR == 0 ? 0 : T;  // time slot of the boundary vector
  VT* cur = (t_first & 1) ? buf1 : buf0;
  __syncthreads();
  {
    int eh = 0;
    for (int q = tid; q < Q; q += NT) eh |= (L.eptr[q + 1] - L.eptr[q]) > kHeavyDeg;
    eps_heavy = __syncthreads_or(eh) != 0;
  }
  for (int q = tid; q < Q; q += NT) {
    cur[q] = (VT)(DIR == 0 ? u.start_w[q] : u.accept_w[q]);
    if (SR == WFL_SEMIRING_TROPICAL && DIR == 0) bptr[u.ab_base + q] = -1;
  }
  closure(cur, t_first);
  __syncthreads();
  for (int q = tid; q < Q; q += NT) out[u.ab_base + (int64_t)t_first * Q + q] = cur[q];

  // Emission rows travel HBM -> registers -> LDS one chunk of R frames ahead of the chain: the loads
  // of chunk c+1 are issued before the first frame of chunk c and land in LDS after its last frame,
  // so their latency is paid once per R frames instead of once per frame.  A chunk is a contiguous
  // slab of xg (R rows of pitch Kmax) in both directions; the backward sweep walks it downwards.
  const int R = rows_per_chunk;
  const int nchunks = (T + R - 1) / R;
  auto chunk_frames = [&](int c, int& f0, int& n) {  // frames [f0, f0+n) in ascending order
    const int s0 = c * R;
    n = min(R, T - s0);
    f0 = DIR == 0 ? s0 : T - s0 - n;
  };
  if (T > 0) {
    int f0, n;
    chunk_frames(0, f0, n);
    const float* src = xg + u.xg_base + (int64_t)f0 * Kmax;
    for (int e = tid; e < n * Kmax; e += NT) L.rows[e] = src[e];
  }
  __syncthreads();
  // one state per thread in the common case: its first four in-arcs live in registers
  const int kq0 = tid < Q ? L.ptr[tid] : 0, kq1 = tid < Q ? L.ptr[tid + 1] : 0;
  const Arc4 mine = load_arc4(L.arcs, kq0, kq1);
  const bool direct = nlev <= 1;  // no epsilon closure: the relaxed value is final
  if (tid == 0) L.heavy[Q] = 0;
  __syncthreads();
  for (int q = tid; q < Q; q += NT)
    if (L.ptr[q + 1] - L.ptr[q] > kHeavyDeg) L.heavy[atomicAdd(&L.heavy[Q], 1)] = q;
  __syncthreads();
  const int n_heavy = L.heavy[Q];
  for (int c = 0; c < nchunks; ++c) {
    int f0, n;
    chunk_frames(c, f0, n);
    const float* tile = L.rows + (size_t)(c & 1) * R * Kmax;
    float pre[kPre];
    int pf0 = 0, pn = 0;
    if (c + 1 < nchunks) {
      chunk_frames(c + 1, pf0, pn);
      const float* src = xg + u.xg_base + (int64_t)pf0 * Kmax;
#pragma unroll
      for (int j = 0; j < kPre; ++j) {
        const int e = tid + j * NT;
        if (e < pn * Kmax) pre[j] = src[e];
      }
    }
    if (SR == WFL_SEMIRING_LOG) {
      if (c > 0) {
        const int tf = DIR == 0 ? f0 : f0 + n;  // slot the first frame of the chunk reads
        VT* fromb = (tf & 1) ? buf1 : buf0;
        float v = WFL_NEG_INF;
        for (int q = tid; q < Q; q += NT) v = fmaxf(v, (float)fromb[q]);
        float m = block_reduce_max(v, L.red);
        if (m > WFL_NEG_INF && m < __builtin_inff()) {
          for (int q = tid; q < Q; q += NT) fromb[q] -= (VT)m;
        } else {
          m = 0.f;
        }
        __syncthreads();
        cum += (double)m;
      }
      if (tid == 0) offs[1 + c] = cum;
    }
    for (int i = 0; i < n; ++i) {
      // forward: consume frame t, produce slot t+1.  backward: consume frame t, produce slot t.
      const int t = DIR == 0 ? f0 + i : f0 + n - 1 - i;
      const int slot_from = DIR == 0 ? t : t + 1, slot_to = DIR == 0 ? t + 1 : t;
      const VT* from = (slot_from & 1) ? buf1 : buf0;
      VT* to = (slot_to & 1) ? buf1 : buf0;
      const float* row = tile + (size_t)(t - f0) * Kmax;
      VT* orow = out + u.ab_base + (int64_t)slot_to * Q;
      if (tid < Q && kq1 - kq0 <= kHeavyDeg) {
        VT v;
        int arg;
        relax_labelled<SR, VT>(L, mine, from, row, kq0, kq0 + 4, kq1, v, arg);
        to[tid] = v;
        if (direct) orow[tid] = v;
        if (SR == WFL_SEMIRING_TROPICAL && DIR == 0) bptr[u.ab_base + (int64_t)slot_to * Q + tid] = arg;
      }
      for (int q = tid + NT; q < Q; q += NT) {
        VT v;
        int arg;
        const int k0 = L.ptr[q], k1 = L.ptr[q + 1];
        if (k1 - k0 > kHeavyDeg) continue;
        relax_labelled<SR, VT>(L, load_arc4(L.arcs, k0, k1), from, row, k0, k0 + 4, k1, v, arg);
        to[q] = v;
        if (direct) orow[q] = v;
        if (SR == WFL_SEMIRING_TROPICAL && DIR == 0) bptr[u.ab_base + (int64_t)slot_to * Q + q] = arg;
      }
      if (n_heavy) {  // one 16-lane row per high in-degree state, NT / 16 states at a time
        const int grow = tid >> 4, nrows = NT >> 4;
        for (int h0 = 0; h0 < n_heavy; h0 += nrows) {  // (uniform trip count: DPP rows need the whole wave)
          const int hq = h0 + grow;
          const int q = hq < n_heavy ? L.heavy[hq] : -1;
          const int k0 = q >= 0 ? L.ptr[q] : 0, k1 = q >= 0 ? L.ptr[q + 1] : 0;
          VT v;
          int arg;
          relax_labelled_row16<SR, VT>(L, from, row, k0, k1, v, arg);
          if (q >= 0 && (tid & 15) == 0) {
            to[q] = v;
            if (direct) orow[q] = v;
            if (SR == WFL_SEMIRING_TROPICAL && DIR == 0) bptr[u.ab_base + (int64_t)slot_to * Q + q] = arg;
          }
        }
      }
      closure(to, slot_to);
      // (LDS traffic only; __syncthreads() would also wait for the frame's own global stores.  Measured on the bigram
      // Transducer's epsilon numerator -- 92 states, 250 frames, scripts/lattice_eps_probe.py -- it is NOT what this
      // sweep waits for: 375 -> 372 us, 1.5 us per frame either way)
      lds_barrier();
      if (!direct)
        for (int q = tid; q < Q; q += NT) orow[q] = to[q];
    }
    if (c + 1 < nchunks) {
      float* dst = L.rows + (size_t)((c + 1) & 1) * R * Kmax;
#pragma unroll
      for (int j = 0; j < kPre; ++j) {
        const int e = tid + j * NT;
        if (e < pn * Kmax) dst[e] = pre[j];
      }
      __syncthreads();
    }
  }
  if (DIR == 0 && logz) {
    const VT* fin = (T & 1) ? buf1 : buf0;
    float m = WFL_NEG_INF;
    for (int q = tid; q < Q; q += NT) m = fmaxf(m, (float)(fin[q] + (VT)u.accept_w[q]));
    m = block_reduce_max(m, L.red);
    double z = m;
    if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF) {
      float s = 0.f;
      for (int q = tid; q < Q; q += NT) s += fast_exp((float)(fin[q] + (VT)u.accept_w[q] - (VT)m));
      s = block_reduce_sum(s, L.red);
      z = (double)m + lse_log((double)s);
    }
    if (tid == 0) {
      const double zd = z + cum;  // (-inf + cum = -inf: no accepting path)
      logz[b] = (float)zd;
      if (SR == WFL_SEMIRING_LOG) z64[b] = zd;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// stage 2, probability domain (the default for the log semiring on "lean" acceptors: one state per thread, no
// epsilon arcs, at most kLeanDeg arcs into AND out of every state -- CTC-like chains, ASG force alignment, STC, the
// Transducer's alignment graphs).
//
// fp32 log-domain sweeps pay a v_exp_f32 per arc and a v_log_f32 per state and frame, and their results carry the
// transcendentals' (biased) rounding: measured 1.5e-4 .. 2.5e-4 on posteriors / log Z after T = 800..1000 frames
// against the float64 oracle, whatever the renormalisation.  Here a state is a DOUBLE probability relative to a
// workgroup-uniform power-of-two scale (renormalised every chunk: exact) and the per-frame references of the gathered
// emission factors; a frame is one multiply-add per arc:
//     p'[q] = sum_{arcs s->q} p[s] * wf[arc] * f_t[slot(arc)]        wf = e^(w - wref), f_t = e^(x_t - r_t) <= 1
// The 11-bit exponent of a double holds what a float cannot: alpha mass piles up at the last states and beta mass
// at the first ones (2^328 apart at T = 1000), far from the diagonal that carries the posteriors.  What even a
// double cannot hold shows up as disagreement between the two sweeps' log Z (alpha from the accept states at T,
// beta from the start states at 0) or as a non-finite / vanished state vector: such an utterance is re-run in the
// log domain by the repair launch that follows (a no-op otherwise) and marked in fmt[b].
//   out[slot][q] (double) relative to offs[slot] in LOG2 units:  log2 value = log2 out + offs[slot]
// ------------------------------------------------------------------------------------------------
constexpr int kFmtLog = 0, kFmtProb = 1;
__device__ __forceinline__ int64_t xg_main_dev(const wfl_lattice_desc& d, int T) {
  return (((int64_t)d.B * T * d.max_labels) + 3) & ~(int64_t)3;
}
constexpr double kLog2e_d = 1.4426950408889634074;
#define WFL_SWEEP_STORE(ptr, v) (*(ptr) = (v))
// ---- progress words of the sweeps (the in-launch gradient, prob_chain_occ_kernel) ---------------------------------
// One 64-bit word per (utterance, direction) behind the dump area of the alpha / beta tails:
//     [63:32] launch token   [31:28] XCC id of the sweep's workgroup   [27:0] chunks whose stores have reached L2
// written by thread 0 of the sweep with one write-through store, polled by the gradient workgroups of the same launch.
// The token (a per-launch counter from the host) makes words left by earlier launches read as "not yet".
constexpr uint32_t kProgSkip = 0x0fffffffu;  // the utterance is not swept in the probability domain
// bit 27 of the chunk field, set from a sweep's crossing on when it stores OCCUPANCIES instead of its own vector beyond
// the middle (run_chain_prob, "meet in the middle"); the count is the 27 bits below it
constexpr uint32_t kProgMitm = 1u << 27, kProgCount = kProgMitm - 1u;
constexpr int kFmtOcc = 2;  // fmt[b]: probability-domain sweeps that met in the middle (see run_chain_prob)
// the middle slot of sweeps that met in the middle (run_chain_prob: T a multiple of 16)
__host__ __device__ inline int mitm_middle(int T) { return 16 * ((T / 16) / 2); }
// chunks each sweep stores as plain vectors before it switches to occupancies: the forward sweep up to the middle slot,
// the backward sweep down to the first chunk boundary of ITS chunking (T - 16 c) at or below the middle
__host__ __device__ inline int mitm_plain_chunks_a(int T) { return (T / 16) / 2; }
__host__ __device__ inline int mitm_plain_chunks_b(int T) { return T / 16 - (T / 16) / 2 + (T % 16 != 0 ? 1 : 0); }
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 15u;
}
__device__ __forceinline__ double ld_l2(const double* p) {  // L1-bypassing: what another CU of this XCD stored during the launch
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void prog_publish(uint64_t* w, uint32_t token, uint32_t chunks) {
  __hip_atomic_store(w, ((uint64_t)token << 32) | ((uint64_t)xcc_id() << 28) | chunks, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__host__ __device__ inline int64_t prog_offset_doubles(const wfl_lattice_desc& d, int nch1) {  // from the tail's start
  return (int64_t)d.B * nch1 + 2 * (int64_t)d.B + 1024 + 1;
}
constexpr int kLiveTile = 32;  // frames per job of the gradient beside the sweeps (at least: T / tiles, rounded up)
                               // (40 / 48 / 56 / 64 with the gradient in two phases, same box, three runs each: 0.324-0.326 / 0.307-0.317 /
                               //  0.307-0.310 / 0.335-0.337 against 0.309-0.335 -- inside the runs' own spread)
// header of the in-flight gradient, behind the progress words and `bad` of the alpha tail (int32 units)
struct OccHeader {
  uint32_t* bad;    // [B]
  int32_t* nx;      // [8]  utterances whose sweeps run on XCD x
  uint32_t* next;   // [8]  next job of XCD x
  int32_t* list;    // [8][B]
  uint32_t* busy;   // [8][256]  == token while a sweep runs on CU (xcc, HW_ID[15:8])
  uint32_t* meta;   // [2]       the launch's token and its tiles per utterance (for served_beside_the_sweeps)
  uint32_t* next_base;  // [8]   next base-row job of XCD x (occ_live_kernel, first phase)
  uint32_t* based;  // [B][tiles] == token once the tile's base rows are written
};
// Did the gradient workgroups that ran beside the sweeps write utterance b's rows?  Not if one of them gave up on it
// (`bad`), nor if jobs of its XCD were left undrawn (an XCD that hosted sweeps but no gradient workgroup).  Read by the
// general gradient kernel of wfl_lattice_grad_rest, launches later.
__device__ __forceinline__ bool served_beside_the_sweeps(const wfl_lattice_desc& d, const float* alpha, int64_t tail, int nch1,
                                                         int b);
__device__ __forceinline__ uint32_t cu_key() {  // this wave's CU among the 8 x 256 the ids can name
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return (xcc_id() & 7u) * 256u + ((v >> 8) & 255u);
}
__device__ __forceinline__ OccHeader occ_header(const wfl_lattice_desc& d, float* alpha, int64_t tail, int nch1) {
  OccHeader h;
  double* ta = reinterpret_cast<double*>(alpha + tail) + prog_offset_doubles(d, nch1);
  h.bad = reinterpret_cast<uint32_t*>(ta + d.B);
  h.nx = reinterpret_cast<int32_t*>(h.bad + ((d.B + 1) & ~1));
  h.next = reinterpret_cast<uint32_t*>(h.nx + 8);
  h.list = h.nx + 16;
  h.busy = reinterpret_cast<uint32_t*>(h.list + 8 * (int64_t)d.B);
  h.meta = h.busy + 8 * 256;
  h.next_base = h.meta + 2;
  h.based = h.next_base + 8;
  return h;
}

__device__ __forceinline__ bool served_beside_the_sweeps(const wfl_lattice_desc& d, const float* alpha, int64_t tail, int nch1,
                                                         int b) {
  const OccHeader h = occ_header(d, const_cast<float*>(alpha), tail, nch1);
  const uint32_t token = h.meta[0], ntiles = h.meta[1];
  if (h.bad[b] == token) return false;
  const uint64_t va = reinterpret_cast<const uint64_t*>(reinterpret_cast<const double*>(alpha + tail) + prog_offset_doubles(d, nch1))[b];
  if ((uint32_t)(va >> 32) != token || ((uint32_t)va & 0x0fffffffu) == kProgSkip) return false;
  const uint32_t x = ((uint32_t)va >> 28) & 7u;
  return h.next[x] >= (uint32_t)h.nx[x] * ntiles;
}
constexpr int kDumpDoubles = 1024;  // scratch behind the alpha / beta tails (see run_chain_prob)
constexpr int kBandDepth = 4;
// floats of the probability-domain sweeps' row tile: two chunks of the tile path, or the banded sweep's two tiles of
// 16 rows (+ their references)
// (a tile slot has room for every thread's kPre prefetch registers: the hand-over stores them all, no test)
__host__ __device__ inline size_t prob_tile_floats(const wfl_lattice_desc& d, int rows_per_chunk, int nt) {
  const size_t rows = (size_t)rows_per_chunk * d.max_labels, regs = (size_t)kPre * nt;
  return rows > regs ? rows : regs;
}
__host__ __device__ inline size_t prob_rows_floats(const wfl_lattice_desc& d, int rows_per_chunk, int nt) {
  const size_t tile = 2 * prob_tile_floats(d, rows_per_chunk, nt);
  const size_t band = d.max_states <= 64 ? (size_t)2 * 1024 + 2 * 64 : 0;
  return tile > band ? tile : band;
}  // chunks of 16 frames the banded sweep loads ahead of the one it works on

struct ProbLds {
  double* buf0;  // [Q]
  double* buf1;  // [Q]
  float* rows;   // [2][prob_tile_floats] emission factors of the current / next chunk, rows of Kmax
  float* refs;   // [2][threads] their references (the first R of each)
  float* red;    // [64] (reductions; reused as int)
};

__device__ __forceinline__ int block_reduce_max_int(int v, int* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_all_max_int(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  int r = red[0];
  for (int i = 1; i < nw; ++i) r = max(r, red[i]);
  return r;
}
__device__ __forceinline__ double block_reduce_sum_f64(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}

// Is utterance `u` one for the probability-domain chains?  (block-uniform; both directions must agree, so both
// degrees are checked by both workgroups)
// Epsilon arcs (back-off transition models): at most kEpsDeg into and out of a state -- they live in the owning thread's
// registers -- in at most kProbMaxLev topological levels (a barrier per level and frame: run_chain_prob, eps_closure).
constexpr int kEpsDeg = 4;
constexpr int kProbMaxLev = 8;
__device__ __forceinline__ bool prob_eligible(const UttView& u, int NT) {  // NT: threads of the CHAIN workgroups
  int bad = (u.Q > NT) | (u.nlev > kProbMaxLev);
  if (!bad)
    for (int q = threadIdx.x; q < u.Q; q += blockDim.x) {
      bad |= (u.in_ptr[q + 1] - u.in_ptr[q] > kLeanDeg) | (u.out_ptr[q + 1] - u.out_ptr[q] > kLeanDeg);
      if (u.E > 0) bad |= (u.ein_ptr[q + 1] - u.ein_ptr[q] > kEpsDeg) | (u.eout_ptr[q + 1] - u.eout_ptr[q] > kEpsDeg);
    }
  return !__syncthreads_or(bad);
}
// the factor of an epsilon arc in the probability domain (sweeps, certificate and gradient form the same float)
__device__ __forceinline__ double eps_factor(const UttView& u, const float* __restrict__ weights, int e) {
  float w = u.eps_w[e];
  const int wid = u.eps_wid[e];
  if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
  return (double)fast_exp(nan_to_neg(w));
}

// Slots s_lo .. s_lo + cnt - 1 (cnt <= 16), of which BOTH sweeps of an utterance have stored their vectors (doubles, rows
// of Q), as occupancies (floats) into the first half of `dst`'s rows -- this sweep's own rows or the partner's:
// everything is read first (the floats of a row overwrite other threads' doubles of the same row), eight slots at a
// time.  Workgroup-wide: barriers inside.  zlog2 = +inf: no accepting path, zeros.
__device__ __noinline__ void mitm_gamma_rows(const double* own, const double* oth, const double* offs, const double* oth_offs,
                                             double* dst, int s_lo, int cnt, int Q, double zlog2) {
  const int tid = threadIdx.x, tidc = min(tid, Q - 1);
  for (int b0 = 0; b0 < cnt; b0 += 8) {
    double av[8], bv[8], ex[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sl = s_lo + min(b0 + j, cnt - 1);
      av[j] = ld_l2(own + (int64_t)sl * Q + tidc);
      bv[j] = ld_l2(oth + (int64_t)sl * Q + tidc);
      ex[j] = ld_l2(offs + sl) + ld_l2(oth_offs + sl);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (b0 + j < cnt && tid < Q)
        reinterpret_cast<float*>(dst + (int64_t)(s_lo + b0 + j) * Q)[tid] =
            zlog2 == __builtin_inf() ? 0.f : (float)(av[j] * bv[j] * exp2(ex[j] - zlog2));
    }
  }
}
// What a sweep needs of its partner (the other direction of the same utterance, same launch) to meet it in the middle.
struct MitmArgs {
  int req = 0;                       // the launch asks for it (wfl_lattice_forward_grad, T a multiple of 16, ...)
  double* oth = nullptr;             // the partner's score rows of this utterance ([T+1][Q] doubles)
  const double* oth_offs = nullptr;  // its per-slot offsets
  const uint64_t* oth_prog = nullptr;  // its progress word
  int32_t* fmt = nullptr;            // forward sweep: fmt[b] <- kFmtOcc
  uint32_t* mitm_b = nullptr;        // backward sweep: its own flag (0 / 1), where the alpha buffer keeps `bad`
  int spins = 1 << 18;
};

// BAND: the register-resident banded sweep is compiled in (chain wave + loader wave workgroups); UNR: full chunks as
// straight-line code (not for the 1024-thread instantiation: its 128-register budget would spill)
// PUB: the sweep publishes its progress (prog_publish) for the gradient workgroups of the same launch
template <int DIR, bool BAND, bool UNR = true, bool PUB = false>
__device__ __forceinline__ void run_chain_prob(const wfl_lattice_desc& d, const UttView& u, const ProbLds& L, int T, int R,
                               const float* __restrict__ fg, const float* __restrict__ rmax,
                               const float* __restrict__ weights, double* __restrict__ out, float* __restrict__ logz,
                               int b, double* __restrict__ offs, double* __restrict__ z64, float* __restrict__ wref_out,
                               double* __restrict__ dump,  // dump: kDumpDoubles doubles nobody reads
                               uint64_t* prog = nullptr, uint32_t token = 0, const MitmArgs& mm = MitmArgs()) {
  const int tid = threadIdx.x, NT = blockDim.x;
  // (the LDS pointers as locals: read through the struct they stayed a 48-byte object in scratch, loaded at every chunk)
  double *const lbuf0 = L.buf0, *const lbuf1 = L.buf1;
  float *const lrows = L.rows, *const lrefs = L.refs, *const lred = L.red;
  const int Q = u.Q, Kmax = d.max_labels;
  // ---- this thread's state: its in-arcs (forward) / out-arcs (backward) in registers
  const int32_t* ptr = DIR == 0 ? u.in_ptr : u.out_ptr;
  const int k0 = tid < Q ? ptr[tid] : 0, k1 = tid < Q ? ptr[tid + 1] : 0;
  int asrc[kLeanDeg], aslot[kLeanDeg];
  float aw[kLeanDeg];
  float wmx = WFL_NEG_INF;
#pragma unroll
  for (int i = 0; i < kLeanDeg; ++i) {
    asrc[i] = 0, aslot[i] = 0, aw[i] = WFL_NEG_INF;
    if (k0 + i < k1) {
      const int a = DIR == 0 ? k0 + i : u.out_arc[k0 + i];
      asrc[i] = DIR == 0 ? u.arc_src[a] : u.arc_dst[a];
      aslot[i] = u.arc_slot[a];
      float w = u.arc_w[a];
      const int wid = u.arc_wid[a];
      if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
      aw[i] = nan_to_neg(w);
      wmx = fmaxf(wmx, aw[i]);
    }
  }
  // (readfirstlane: these are wave-uniform, and the compiler must KNOW it -- a branch it takes for divergent becomes two
  // masked regions with a static path around both, and on that path the loads of the chunk loop's prefetch stay
  // unconsumed: see hand_over)
  const int deg_class = __builtin_amdgcn_readfirstlane(__syncthreads_or(k1 - k0 > 4) ? 2 : (__syncthreads_or(k1 - k0 > 2) ? 1 : 0));
  const int wave_deg = __builtin_amdgcn_readfirstlane(wave_all_max_int(k1 - k0));  // the most arcs into (out of) a state of this wave
  float wref = block_reduce_max(wmx, lred);  // every frame multiplies by e^wref once more: part of the offset
  if (!(wref > WFL_NEG_INF)) wref = 0.f;
  if (wref_out && tid == 0) wref_out[b] = wref;
  double wf[kLeanDeg];  // (float exponential: the gradient kernel recomputes the same factor)
#pragma unroll
  for (int i = 0; i < kLeanDeg; ++i) wf[i] = (double)fast_exp(aw[i] - wref);

  // single-wave banded acceptors (ASG force alignment, CTC-like chains without skips): every arc comes from the state
  // itself or from its neighbour -- the state vector stays in registers, the neighbour is one DPP wave shift away
  const int adj = DIR == 0 ? tid - 1 : tid + 1;
  double wf_self = 0.0, wf_adj = 0.0;
  int slot_self = 0, slot_adj = 0, band_ok = 1;
#pragma unroll
  for (int i = 0; i < kLeanDeg; ++i) {
    if (!(aw[i] > WFL_NEG_INF)) continue;
    if (asrc[i] == tid && wf_self == 0.0)
      wf_self = wf[i], slot_self = aslot[i];
    else if (asrc[i] == adj && wf_adj == 0.0)
      wf_adj = wf[i], slot_adj = aslot[i];
    else
      band_ok = 0;
  }
  const bool banded = BAND && NT == 128 && Q <= 64 && (Kmax & 3) == 0 && Kmax <= 64 && __syncthreads_and(band_ok && u.E == 0);
  // "uniform-label" acceptors: every arc INTO a state carries the same emission column (CTC-like chains, force
  // alignment, token-level alignment graphs: the label belongs to the destination state).  Then the frame's factor is
  // applied once per state by its owner -- after the sum (alpha), or before publishing (beta: the owner publishes
  // f_t[label(d)] * beta_{t+1}[d]) -- instead of once per arc: one LDS read, one conversion and one multiply per
  // thread and frame where the general form needs kLeanDeg of each.
  int my_slot = 0, uni_ok = 1;
  if (tid < Q) {
    const int i0 = u.in_ptr[tid], i1 = u.in_ptr[tid + 1];
    if (i0 < i1) my_slot = u.arc_slot[i0];
    for (int k = i0 + 1; k < i1; ++k) uni_ok &= u.arc_slot[k] == my_slot;
  }
  const bool uniform = __builtin_amdgcn_readfirstlane(__syncthreads_and(uni_ok && u.E == 0)) != 0;
  // ---- epsilon arcs into (forward) / out of (backward) this thread's state, in registers; states are numbered level by
  // level (u.lvl_ptr), an epsilon arc leads from a lower level to a higher one.  After the labelled arcs of a frame the
  // levels are closed one after the other, a barrier each (eps_closure): the same order as run_chain's closure.
  const int nlev = u.nlev;
  const bool eps_on = u.E > 0 && nlev > 1;  // (block-uniform)
  int esrc[kEpsDeg], my_lev = 0, ne = 0;
  double ewf[kEpsDeg];
#pragma unroll
  for (int i = 0; i < kEpsDeg; ++i) esrc[i] = 0, ewf[i] = 0.0;
  if (eps_on && tid < Q) {
    const int32_t* eptr = DIR == 0 ? u.ein_ptr : u.eout_ptr;
    const int e0 = eptr[tid], e1 = eptr[tid + 1];
    ne = e1 - e0;
#pragma unroll
    for (int i = 0; i < kEpsDeg; ++i)
      if (e0 + i < e1) {
        const int e = DIR == 0 ? e0 + i : u.eout_arc[e0 + i];
        esrc[i] = DIR == 0 ? u.eps_src[e] : u.eps_dst[e];
        ewf[i] = eps_factor(u, weights, e);
      }
    for (int l = 1; l < nlev; ++l)
      if (tid >= u.lvl_ptr[l]) my_lev = l;
  }
  const bool wave_live = __builtin_amdgcn_readfirstlane((int)((tid & ~63) < Q)) != 0;  // (waves without a state only take part in the barriers)

  const int t_first = DIR == 0 ? 0 : T;
  double p = 0.0;
  if (tid < Q) p = (DIR == 0 ? u.start_w[tid] : u.accept_w[tid]) > WFL_NEG_INF ? 1.0 : 0.0;  // (boundary weights are 0 / -inf)
  double cum = 0.0;  // log2 of everything factored out of the stored probabilities so far
  double* cur = (t_first & 1) ? lbuf1 : lbuf0;
  // `vec` holds the frame's vector after its labelled arcs, this thread's entry = p, a barrier since: level by level,
  // p += sum over the epsilon arcs of vec[other end] x factor.  Ends behind a barrier.  (workgroup-wide)
  auto eps_closure = [&](double* vec) {
    for (int step = 1; step < nlev; ++step) {
      const int lev = DIR == 0 ? step : nlev - 1 - step;
      if (my_lev == lev && ne > 0) {
        double sum = p;
#pragma unroll
        for (int k = 0; k < kEpsDeg; ++k) sum = fma(vec[esrc[k]], ewf[k], sum);
        p = sum;
        vec[tid] = p;
      }
      lds_barrier();
    }
  };
  if (tid < Q) cur[tid] = p;
  if (eps_on) {
    lds_barrier();
    eps_closure(cur);
  }
  if (tid < Q) out[u.ab_base + (int64_t)t_first * Q + tid] = p;
  if (tid == 0) offs[t_first] = 0.0;

  if constexpr (BAND) if (banded) {
    // The whole sweep in ONE wave's registers (wave 0), fed by a loader wave (wave 1); the two meet at one LDS-only
    // barrier per chunk of 16 frames.  A frame is ~100 cycles of dependent arithmetic and an HBM round trip is
    // 2000-5000, so the rows must be requested several chunks ahead -- and on gfx9 a wave's loads and stores share
    // one in-order counter (vmcnt), so a wave that both streams its scores out every frame and waits for prefetched
    // rows ends up waiting for its own stores.  Hence two waves: the chain wave only stores, the loader only loads
    // (kBandDepth chunks in flight in its registers: the compact rows of a chunk are contiguous, up to four float4
    // per lane) and hands each chunk over through a double-buffered LDS tile.  The tile path below has one chunk of
    // lookahead and a barrier per FRAME: 235 us at T = 1000 for the ASG force-alignment lattice; this one 60.
    constexpr int RB = 16, D = kBandDepth, NV = 4;
    const int K4 = Kmax >> 2, nch = (T + RB - 1) / RB;
    constexpr int kSlot = 64 * 4 * NV;     // floats per tile (every loader lane stores its NV float4: no divergence)
    float* ring = lrows;                  // [2][kSlot], rows of RB * Kmax <= kSlot floats
    float* rref = ring + 2 * kSlot;        // [2][64]: per-frame references of the chunk (first RB entries)
    auto chunk_lo = [&](int c, int n) { return DIR == 0 ? c * RB : T - c * RB - n; };  // lowest frame of chunk c
    if (tid >= 64) {
      // ---- loader
      const int l = tid - 64;
      const float4* fgu = reinterpret_cast<const float4*>(fg + u.xg_base);  // (Kmax % 4 == 0: checked by `banded`)
      const float* rmu = rmax + (int64_t)b * T;
      float gv[D][4 * NV];  // (plain floats: an array of float4 is not promoted to registers)
      float gr[D];
      auto load = [&](int c, float (&v_)[4 * NV], float& r_) {
        const int cc = min(c, nch - 1);  // (past the end: the last chunk again, never handed over)
        const int n = min(RB, T - cc * RB);
        const int64_t base4 = (int64_t)chunk_lo(cc, n) * K4;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const float4 q = fgu[base4 + min(l + 64 * k, n * K4 - 1)];
          v_[4 * k] = q.x, v_[4 * k + 1] = q.y, v_[4 * k + 2] = q.z, v_[4 * k + 3] = q.w;
        }
        r_ = rmu[min(chunk_lo(cc, n) + (l & 15), T - 1)];
      };
      auto hand_over = [&](int c, float (&v_)[4 * NV], float& r_) {  // chunk c -> LDS slot c & 1, then refill
        // (branch-free on purpose: around a divergent branch the compiler waits for ALL outstanding loads)
        float4* dst = reinterpret_cast<float4*>(ring + (size_t)(c & 1) * kSlot);
#pragma unroll
        for (int k = 0; k < NV; ++k) dst[l + 64 * k] = make_float4(v_[4 * k], v_[4 * k + 1], v_[4 * k + 2], v_[4 * k + 3]);
        rref[(c & 1) * 64 + l] = r_;
        load(c + D, v_, r_);
      };
#pragma unroll
      for (int j = 0; j < D; ++j) load(j, gv[j], gr[j]);
      hand_over(0, gv[0], gr[0]);
      lds_barrier();
      // while the chain wave works on chunk c, prepare chunk c + 1.  The steady state is straight-line code (D
      // hand-overs per trip, no test in between): the compiler counts the loads a wait may leave outstanding along
      // the SHORTEST path, so a skipped hand-over anywhere in the loop would shrink every wait to "all but the last".
      int c = 0;
      for (; c + D < nch; c += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          hand_over(c + j + 1, gv[(j + 1) % D], gr[(j + 1) % D]);
          lds_barrier();
        }
      }
      for (; c < nch; ++c) {  // the last D chunks or fewer
#pragma unroll
        for (int j = 0; j < D; ++j)
          if (c + 1 < nch && (c + 1) % D == j) hand_over(c + 1, gv[j], gr[j]);
        lds_barrier();
      }
    } else {
      // ---- chain
      const int ss = tid < Q ? slot_self : 0, sa = tid < Q ? slot_adj : 0;
      bool two_l = ss != sa;
      const bool two = __any(two_l);
      lds_barrier();
      for (int c = 0; c < nch; ++c) {
        const int n = min(RB, T - c * RB);
        const float* tile = ring + (size_t)(c & 1) * kSlot;
        if (c > 0) {  // power-of-two renormalisation of the vector the chunk starts from (exact)
          int ex = (tid < Q && p > 0.0) ? __builtin_amdgcn_frexp_exp(p) - 1 : -(1 << 30);
          ex = wave_all_max_int(ex);  // (DPP: six VALU instructions; the __shfl_xor form is six ds_bpermute round trips)
          if (ex > -(1 << 30) && ex < 2000) p = ldexp(p, -ex), cum += (double)ex;
        }
        // this lane's factors of the chunk: all LDS reads issued back to back, THEN converted (a test or a use
        // between two reads makes the compiler wait for each read in turn: 16 LDS round trips per chunk)
        float fs[RB], fa[RB];
        const int r0 = DIR == 0 ? 0 : n - 1, dr = DIR == 0 ? Kmax : -Kmax;
        const float* ts = tile + r0 * Kmax + ss;
        const float* ta = tile + r0 * Kmax + sa;
#pragma unroll
        for (int i = 0; i < RB; ++i) fs[i] = ts[(i < n ? i : 0) * dr];  // (i >= n: row r0 again, not consumed)
        if (two) {
#pragma unroll
          for (int i = 0; i < RB; ++i) fa[i] = ta[(i < n ? i : 0) * dr];
        } else {
#pragma unroll
          for (int i = 0; i < RB; ++i) fa[i] = fs[i];
        }
        double cs[RB], ca[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) cs[i] = wf_self * (double)fs[i], ca[i] = wf_adj * (double)fa[i];
        // offsets of the chunk's frames: inclusive prefix sum of the per-frame log2 factors over lanes 0..15
        // (lane i: the chunk's i-th frame IN SWEEP ORDER)
        const float rsel = rref[(c & 1) * 64 + ((DIR == 0 ? (tid & 15) : n - 1 - (tid & 15)) & 15)];
        double pre = (tid & 15) < n ? ((double)rsel + (double)wref) * kLog2e_d : 0.0;
#pragma unroll
        for (int o = 1; o < RB; o <<= 1) {
          const int lo = __double2loint(pre), hi = __double2hiint(pre);  // row_shr:o, lanes without a source read 0
          const int slo = o == 1   ? __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true)
                          : o == 2 ? __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true)
                          : o == 4 ? __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true)
                                   : __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true);
          const int shi = o == 1   ? __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true)
                          : o == 2 ? __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true)
                          : o == 4 ? __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true)
                                   : __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true);
          pre += __hiloint2double(shi, slo);
        }
        // lane i: the offset after the chunk's i-th frame -- the chunk's offsets in one store
        if (tid < n) offs[DIR == 0 ? c * RB + tid + 1 : T - 1 - (c * RB + tid)] = cum + pre;
        cum += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pre), 15),
                                __builtin_amdgcn_readlane(__double2loint(pre), 15));  // (lanes >= n added 0)
        // The frames: two DPP moves, a multiply, an fma and a store each -- a single wave issues one instruction every
        // ~5 cycles, so the instruction count IS the frame time.  Lanes without a state sit the loop out (a DPP read
        // from a disabled lane returns 0, which is what their probability is), the row pointer is wave-uniform
        // (scalar adds), and full chunks run without the per-frame bound test.
        double* orow = out + u.ab_base + (int64_t)(DIR == 0 ? c * RB + 1 : T - 1 - c * RB) * Q;
        auto frames16 = [&](auto full) {
          constexpr bool FULL = decltype(full)::value;
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            if (FULL || i < n) {
              // the neighbour's value: both halves of the double through a DPP wave shift (lanes without one: 0)
              const int lo = __double2loint(p), hi = __double2hiint(p);
              const int nlo = DIR == 0 ? __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false)
                                       : __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
              const int nhi = DIR == 0 ? __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false)
                                       : __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
              const double pn = __hiloint2double(nhi, nlo);
              p = fma(p, cs[i], pn * ca[i]);
              orow[tid] = p;
              orow = DIR == 0 ? orow + Q : orow - Q;
            }
          }
        };
        if (tid < Q) {
          if (n == RB)
            frames16(std::true_type{});
          else
            frames16(std::false_type{});
        }
        lds_barrier();
      }
    }
    __syncthreads();
  }
  const int nchunks = banded ? 0 : (T + R - 1) / R;
  const size_t tstride = prob_tile_floats(d, R, NT);
  auto chunk_frames = [&](int c, int& f0, int& n) {
    const int s0 = c * R;
    n = min(R, T - s0);
    f0 = DIR == 0 ? s0 : T - s0 - n;
  };
  if (T > 0 && !banded) {
    int f0, n;
    chunk_frames(0, f0, n);
    const float* src = fg + u.xg_base + (int64_t)f0 * Kmax;
    for (int e = tid; e < n * Kmax; e += NT) lrows[e] = src[e];
    if (tid < n) lrefs[tid] = rmax[(int64_t)b * T + f0 + tid];
  }
  // (vmcnt(0), once, as the BUILTIN -- which the compiler's wait-count pass sees, unlike an asm statement: no load of
  // the set-up above is pending, on any path, when the chunk loop starts)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
#ifdef WFL_SWEEP_PHASE_TIMERS
  long long e_pre = 0, e_ren = 0, e_offs = 0, e_fr = 0, e_hand = 0;
  const long long exp_begin = clock64();
#endif
  // The chunk loop.  A chunk asks for the next chunk's rows at its start (registers), sweeps its sixteen frames -- a
  // score store each -- and moves the rows to LDS at its end (hand_over).  Loads and stores share ONE in-order counter
  // (vmcnt) and the compiler's wait-count pass is static: in front of the first use (or overwrite) of a register with a
  // load pending it waits until as many operations may be outstanding as the path with the FEWEST operations since that
  // load has issued.  With the straight-line frames on every path that is "all but the sixteen youngest" -- the loads,
  // issued 5000 cycles earlier, and nothing else.  With ANY static path that issues fewer stores, or none, it is "all":
  // the wave waits for its own last stores, a store round trip (~800 cycles) at the chunk's end and another at the next
  // chunk's top -- 1800 of a chunk's 8100 cycles (WFL_SWEEP_PHASE_TIMERS; rocprof: prob_chain_pub_kernel<512> 192 ->
  // 160 us at the Transducer benchmark).  Such paths were, and what removed each:
  //  - the waves without a state (no stores), the chunks taken by a frame LOOP (a count the pass cannot know) and the
  //    seven frame-loop variants, all chosen at run time inside the chunk: the chain of uniform tests reaches the pass as
  //    flag-driven regions with a static path around ALL of them, on which the rows are never consumed -- so the first
  //    reuse of their registers, at the top of the next chunk, waited for everything.  Now the variant (SEL), the waves
  //    without a state (LIVE) and the straight-line chunks (FULL) are compile-time choices of the body, made outside
  //    the loop: every copy of the body is one path from the prefetch to the hand-over;
  //  - the hand-over behind the join of the frame paths: it sits at the end of each path now, and each ends in its own
  //    (empty) asm statement -- identical tails are merged behind the join again otherwise;
  //  - a test around a load or around the use of a loaded register (the path that skips it leaves the load pending):
  //    clamped addresses, every thread stores every register (the tile slots have room: prob_tile_floats), the last
  //    chunk asks for itself again; and no initial value for those registers (`rpre = 0` is a write to a register with
  //    a load pending on such a path).
  // 16-byte loads and LDS stores on the way (rows are padded to four labels: pack.cpp, pad_labels).
  // ---- meeting the partner in the middle (PUB launches that ask for it: MitmArgs).  The emission gradient of a
  // uniform-label acceptor only needs the state occupancies gamma_s[q] = alpha_s[q] beta_s[q] / Z, and past the middle
  // slot m a sweep can form them itself: the partner passed those slots in ITS first half and its vectors lie in the
  // L2 of the XCD both run on.  So a sweep stores its own vector (doubles) up to m, then occupancies (floats, Z taken at
  // m: every path passes through exactly one state per slot) into the first half of its own rows:
  //     alpha buffer: slots 0 .. m alpha (doubles), m+1 .. T gamma (floats);   beta buffer: slots m+1 .. T beta, 0 .. m gamma
  // (gamma_m by the forward sweep, into the beta buffer).  16 bytes per (slot, state) written and read back by the
  // gradient become 8 + 4 written, 8 read by the partner and 4 by the gradient, and the gradient needs neither offsets
  // nor a Z of its own.  Both sweeps must agree; each decides for itself at its crossing (the partner's progress word:
  // same launch, same XCD, its half stored) and says so in its progress word (kProgMitm), fmt[b] / mitm_b[b].
  double pbA[8], pbB[8];  // the partner's values of this thread's state: frames 0-7 / 8-15 of the chunk
  double oo_cur = 0.0;    // lane li: the partner's offset at the slot of the chunk's li-th frame
  double zlog2 = 0.0;     // log2 Z (+inf: no accepting path)
  bool mitm_on = false;
  const int tidc = min(tid, Q - 1);
  auto slot_of = [&](int c, int i) { return DIR == 0 ? c * 16 + i + 1 : T - c * 16 - 1 - i; };
  // chunks whose sixteen frames run as straight-line code (nchunks is 0 when the banded sweep above has done the work)
  const int nfull = (UNR && R == 16) ? min(T / 16, nchunks) : 0;
  // (MITM) slots of which BOTH sweeps have stored their vectors, as occupancies: mitm_gamma_rows, a real call -- inlined
  // at its two sites in each of the eight copies of the sweep it cost the kernel 9 VGPRs and 200 more spilled SGPRs
  auto gamma_rows = [&](int s_lo, int cnt, double* dst) {
    mitm_gamma_rows(out + u.ab_base, mm.oth, offs, mm.oth_offs, dst, s_lo, cnt, Q, zlog2);
  };
  auto sweep = [&](auto sel_, auto live_) {
    constexpr int SEL = decltype(sel_)::value;
    constexpr bool LIVE = decltype(live_)::value;
    auto chunk = [&](int c, auto full_, auto mitm_) {
      constexpr bool FULL = decltype(full_)::value;
      constexpr bool MITM = PUB && FULL && LIVE && SEL < 4 && decltype(mitm_)::value;  // (uniform-label frame loops)
#ifdef WFL_SWEEP_PHASE_TIMERS
      const long long eA = clock64();
#endif
      int f0, n;
      chunk_frames(c, f0, n);
      const float* tile = lrows + (size_t)(c & 1) * tstride;
      const float* rtile = lrefs + (size_t)(c & 1) * NT;
      // the next chunk's rows (see above: no test, no initial value)
      float pre[kPre], rpre;
      int pf0, pn;
      chunk_frames(min(c + 1, nchunks - 1), pf0, pn);
      {
        const float4* src4 = reinterpret_cast<const float4*>(fg + u.xg_base + (int64_t)pf0 * Kmax);
        const int last4 = ((pn * Kmax) >> 2) - 1;
        rpre = rmax[(int64_t)b * T + pf0 + min(tid, pn - 1)];
#pragma unroll
        for (int j = 0; j < kPre / 4; ++j) {
          const float4 q = src4[min(tid + j * NT, last4)];
          pre[4 * j] = q.x, pre[4 * j + 1] = q.y, pre[4 * j + 2] = q.z, pre[4 * j + 3] = q.w;
        }
      }
      double oo_next = 0.0;
      double corrv = 0.0;  // (MITM) lane li: 2^(own offset + the partner's - log2 Z) at the slot of the chunk's li-th frame
      if constexpr (MITM) {
        // the partner's values for the second half of this chunk (its first half came with the chunk before, at frame 7),
        // and its offset for the next chunk's frames: L1-bypassing loads (another CU wrote them during this launch)
#pragma unroll
        for (int j = 0; j < 8; ++j) pbB[j] = ld_l2(mm.oth + (int64_t)slot_of(c, 8 + j) * Q + tidc);
        oo_next = ld_l2(mm.oth_offs + slot_of(min(c + 1, nfull - 1), tid & 15));
      }
      // (`path`: 0 behind the sixteen straight-line frames, 1 a wave without a state, 2 behind a frame loop)
      auto hand_over = [&](auto path) {
        lrefs[(size_t)((c + 1) & 1) * NT + tid] = rpre;
        float4* dst4 = reinterpret_cast<float4*>(lrows + (size_t)((c + 1) & 1) * tstride);
#pragma unroll
        for (int j = 0; j < kPre / 4; ++j)
          dst4[tid + j * NT] = make_float4(pre[4 * j], pre[4 * j + 1], pre[4 * j + 2], pre[4 * j + 3]);
        if constexpr (decltype(path)::value == 0)
          asm volatile("; rows handed over behind sixteen straight-line frames" ::: "memory");
        else if constexpr (decltype(path)::value == 1)
          asm volatile("; rows handed over by a wave without a state" ::: "memory");
        else
          asm volatile("; rows handed over behind a frame loop" ::: "memory");
      };
      using PathUnrolled = std::integral_constant<int, 0>;
      using PathIdle = std::integral_constant<int, 1>;
      using PathLoop = std::integral_constant<int, 2>;
#ifdef WFL_SWEEP_PHASE_TIMERS
      const long long eB = clock64();
      e_pre += eB - eA;
#endif
      // (every 4th chunk: a renormalisation is three barriers, ~1100 cycles of a chunk's 9300, and a double has room for
      // far more than 64 frames of factors <= 1 -- what it has no room for, the certificate catches)
      if (c > 0 && (c & 3) == 0) {  // power-of-two renormalisation of the vector the chunk starts from (exact)
        // (LDS-only barriers: __syncthreads() also waits for the wave's global operations -- the loads issued just above,
        // i.e. an HBM round trip in every chunk: 1300-1400 cycles in this section)
        const int ex = (tid < Q && p > 0.0) ? __builtin_amdgcn_frexp_exp(p) - 1 : -(1 << 30);
        int* red = (int*)lred;
        const int wmax = wave_all_max_int(ex);
        lds_barrier();
        if ((tid & 63) == 0) red[tid >> 6] = wmax;
        lds_barrier();
        int emax = red[0];
        for (int i = 1; i < (NT + 63) >> 6; ++i) emax = max(emax, red[i]);
        if (emax > -(1 << 30) && emax < 2000) {
          p = ldexp(p, -emax);
          cum += (double)emax;
          double* fromb = ((DIR == 0 ? f0 : f0 + n) & 1) ? lbuf1 : lbuf0;
          if (tid < Q) fromb[tid] = p;
        }
        lds_barrier();
      }
#ifdef WFL_SWEEP_PHASE_TIMERS
      const long long eC = clock64();
      e_ren += eC - eB;
#endif
      // Software pipeline: the arc coefficients c[k] = wf[k] * f_t[slot_k] of a frame do not depend on the chain, so
      // they are formed while the previous frame's sources are still on their way from LDS; after the barrier only the
      // DEG source reads (issued back to back) and DEG multiply-adds (two independent accumulators) remain.
      auto coeffs = [&](int i, double (&c)[kLeanDeg], auto deg) {
        constexpr int DEG = decltype(deg)::value;
        const int t = DIR == 0 ? f0 + i : f0 + n - 1 - i;
        const float* row = tile + (size_t)(t - f0) * Kmax;
        float f[DEG];
#pragma unroll
        for (int k = 0; k < DEG; ++k) f[k] = row[aslot[k]];
#pragma unroll
        for (int k = 0; k < DEG; ++k) c[k] = wf[k] * (double)f[k];
      };
      // Frame loops.  The workgroup's waves meet at ONE LDS-only barrier per frame and a wave issues an instruction every
      // ~5 cycles, so the instruction count of a frame is its time (SQ counters at cfg4: 27 VALU + 19 SALU + 5 LDS
      // instructions per live wave and frame, waves waiting 63 % of their cycles): waves without a state only run the
      // barriers, the row pointer and the tile pointer advance by scalar adds, the ping-pong buffers swap by parity.
      auto frames = [&](auto deg) {
        constexpr int DEG = decltype(deg)::value;
        double c[kLeanDeg], cn[kLeanDeg];
        coeffs(0, c, deg);
        double* orow = out + u.ab_base + (int64_t)(DIR == 0 ? f0 + 1 : f0 + n - 1) * Q;
        int par = (DIR == 0 ? f0 : f0 + n) & 1;  // parity of the slot the frame reads from
        if constexpr (FULL) {  // (a full chunk as straight-line code with unconditional stores: see frames_uniform)
          const bool mine = tid < Q;
          double* po = mine ? orow + tid : dump + (tid & (kDumpDoubles - 1));
          const int64_t pstep = mine ? (DIR == 0 ? (int64_t)Q : -(int64_t)Q) : 0;
          const double* bA = par ? lbuf1 : lbuf0;
          const double* bB = par ? lbuf0 : lbuf1;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const double* from = (i & 1) ? bB : bA;
            double* to = const_cast<double*>((i & 1) ? bA : bB);
            double ps[DEG];
#pragma unroll
            for (int k = 0; k < DEG; ++k) ps[k] = from[asrc[k]];
            if (i + 1 < 16) coeffs(i + 1, cn, deg);
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int k = 0; k < DEG; k += 2) {
              acc0 = fma(ps[k], c[k], acc0);
              acc1 = fma(ps[k + 1], c[k + 1], acc1);
            }
            p = acc0 + acc1;
            to[tid] = p;
            if (eps_on) {  // (block-uniform; LDS and barriers only -- no memory operation on either side of the test)
              lds_barrier();
              eps_closure(to);
            }
            WFL_SWEEP_STORE(po, p);
            po += pstep;
#pragma unroll
            for (int k = 0; k < DEG; ++k) c[k] = cn[k];
            if (!eps_on) lds_barrier();
          }
          hand_over(PathUnrolled{});
          return;
        }
        for (int i = 0; i < n; ++i) {
          const double* from = par ? lbuf1 : lbuf0;
          double* to = par ? lbuf0 : lbuf1;
          double ps[DEG];
#pragma unroll
          for (int k = 0; k < DEG; ++k) ps[k] = from[asrc[k]];
          if (i + 1 < n) coeffs(i + 1, cn, deg);
          double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
          for (int k = 0; k < DEG; k += 2) {
            acc0 = fma(ps[k], c[k], acc0);
            acc1 = fma(ps[k + 1], c[k + 1], acc1);
          }
          p = acc0 + acc1;
          if (tid < Q) to[tid] = p;
          if (eps_on) {
            lds_barrier();
            eps_closure(to);
          }
          if (tid < Q) WFL_SWEEP_STORE(orow + tid, p);
#pragma unroll
          for (int k = 0; k < DEG; ++k) c[k] = cn[k];
          orow = DIR == 0 ? orow + Q : orow - Q;
          par ^= 1;
          if (!eps_on) lds_barrier();  // (not __syncthreads: the stores of this frame's scores need not have landed)
        }
        hand_over(PathLoop{});
      };
      auto frames_uniform = [&](auto deg) {
        constexpr int DEG = decltype(deg)::value;
        if (DIR == 1) {
          // beta: the LDS vector holds G = f[label(d)] * beta[d] for the frame about to be consumed; at the chunk's
          // first frame it still holds plain beta (the tile of this chunk was not there when it was written)
          const int t0 = f0 + n - 1;
          const double* fromb = ((t0 + 1) & 1) ? lbuf1 : lbuf0;
          if (tid < Q) {  // (own entry only: no hazard before the write)
            double* fb = const_cast<double*>(fromb);
            fb[tid] = fb[tid] * (double)tile[(size_t)(t0 - f0) * Kmax + my_slot];
          }
          lds_barrier();
        }
        double* orow = out + u.ab_base + (int64_t)(DIR == 0 ? f0 + 1 : f0 + n - 1) * Q;
        int par = (DIR == 0 ? f0 : f0 + n) & 1;
        // alpha: this frame's factor of the state; beta: the factor of the NEXT frame to be consumed (t - 1), with
        // which the owner publishes; past the chunk (the tile is not there yet) plain beta is published
        const float* fptr = tile + (size_t)(DIR == 0 ? 0 : max(n - 2, 0)) * Kmax + my_slot;
        if constexpr (FULL) {
          // A full chunk as straight-line code whose stores EVERY lane executes (lanes without a state store to a dump
          // and to their own, never read, LDS entry).  The threads of this workgroup also prefetch the next chunk's rows:
          // loads and stores share the in-order vmcnt counter and the compiler counts, for the wait in front of the
          // prefetched rows, only the operations that are issued on EVERY path -- with the stores inside `if (tid < Q)`
          // or inside a loop of unknown trip count that wait became "everything", i.e. the last frame's store round trip
          // (~1.5 us) at the end of every chunk.
          const bool mine = tid < Q;
          double* po = mine ? orow + tid : dump + (tid & (kDumpDoubles - 1));
          const int64_t pstep = mine ? (DIR == 0 ? (int64_t)Q : -(int64_t)Q) : 0;
          // (past the middle: occupancies, floats, into the first half of the same rows)
          float* pof = mine ? reinterpret_cast<float*>(orow) + tid : reinterpret_cast<float*>(dump) + (tid & (kDumpDoubles - 1));
          const int64_t pfstep = 2 * pstep;
          const int cn = min(c + 1, nfull - 1);  // (occupancy chunks are whole chunks)
          const double* bA = par ? lbuf1 : lbuf0;  // read by the even frames of the chunk, written by the odd ones
          const double* bB = par ? lbuf0 : lbuf1;
          const int fstep = DIR == 0 ? Kmax : -Kmax;
          // (MITM) the occupancy of frame i is formed one frame late, in the shadow of frame i + 1's LDS reads: on the
          // frame's own path -- sum, LDS write, barrier -- its three multiplies, two lane reads and conversion cost the
          // sweep 85 ns a frame (193 instead of 159 us at the Transducer benchmark)
          double p_late = 0.0;
          auto emit_gamma = [&](int i, double pv) {
            // gamma = own value x the partner's x 2^(both offsets - log2 Z) (the frame's power of two: lane i of corrv)
            const double pbv = i < 8 ? pbA[i & 7] : pbB[i & 7];
            const double cr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(corrv), i),
                                               __builtin_amdgcn_readlane(__double2loint(corrv), i));
            *pof = (float)(pv * pbv * cr);
            pof += pfstep;
            if (i == 7) {  // the first half of the NEXT chunk (the last chunk asks for itself again: not consumed)
#pragma unroll
              for (int j = 0; j < 8; ++j) pbA[j] = ld_l2(mm.oth + (int64_t)slot_of(cn, j) * Q + tidc);
            }
          };
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const double* from = (i & 1) ? bB : bA;
            double* to = const_cast<double*>((i & 1) ? bA : bB);
            double ps[DEG];
#pragma unroll
            for (int k = 0; k < DEG; ++k) ps[k] = from[asrc[k]];
            if constexpr (MITM) {
              if (i > 0) {
                __builtin_amdgcn_sched_barrier(0);
                emit_gamma(i - 1, p_late);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            const float fr = fptr[(DIR == 0 || i < 15) ? i * fstep : 14 * fstep];
            const float f = (DIR == 0 || i < 15) ? fr : 1.f;
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int k = 0; k < DEG; k += 2) {
              acc0 = fma(ps[k], wf[k], acc0);
              acc1 = fma(ps[k + 1], wf[k + 1], acc1);
            }
            const double sum = acc0 + acc1;
            p = DIR == 0 ? sum * (double)f : sum;
            to[tid] = DIR == 0 ? p : p * (double)f;
            if constexpr (MITM) {
              p_late = p;
            } else {
              WFL_SWEEP_STORE(po, p);
              po += pstep;
            }
            lds_barrier();
          }
          if constexpr (MITM) emit_gamma(15, p_late);
          hand_over(PathUnrolled{});
          return;
        }
        for (int i = 0; i < n; ++i) {
          const double* from = par ? lbuf1 : lbuf0;
          double* to = par ? lbuf0 : lbuf1;
          double ps[DEG];
#pragma unroll
          for (int k = 0; k < DEG; ++k) ps[k] = from[asrc[k]];
          const float fr = *fptr;
          const float f = (DIR == 0 || i + 1 < n) ? fr : 1.f;
          double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
          for (int k = 0; k < DEG; k += 2) {
            acc0 = fma(ps[k], wf[k], acc0);
            acc1 = fma(ps[k + 1], wf[k + 1], acc1);
          }
          const double sum = acc0 + acc1;
          p = DIR == 0 ? sum * (double)f : sum;
          if (tid < Q) {
            to[tid] = DIR == 0 ? p : p * (double)f;
            WFL_SWEEP_STORE(orow + tid, p);
          }
          orow = DIR == 0 ? orow + Q : orow - Q;
          if (DIR == 0 || i + 2 < n) fptr = DIR == 0 ? fptr + Kmax : fptr - Kmax;
          par ^= 1;
          lds_barrier();
        }
        // (beta: the chunk's last step published plain beta -- no factor past the chunk -- which is what the
        // renormalisation and the next chunk's first step expect)
        hand_over(PathLoop{});
      };
      // The chunk's per-slot offsets at once, outside the frame loop (n <= 16 frames): lane i of every row of 16 takes
      // the chunk's i-th frame in sweep order, an inclusive prefix sum over the row (DPP) gives the offset after each
      // frame, wave 0 stores them in one instruction.  Inside the loop the same bookkeeping was an LDS read whose wait
      // sat in front of every frame's barrier.
      double chunk_log2;
      {
        const int li = tid & 15;
        const float rsel = rtile[(DIR == 0 ? li : n - 1 - li) & 15];
        double pre = li < n ? ((double)rsel + (double)wref) * kLog2e_d : 0.0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const int lo = __double2loint(pre), hi = __double2hiint(pre);  // row_shr:o, lanes without a source read 0
          const int slo = o == 1   ? __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true)
                          : o == 2 ? __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true)
                          : o == 4 ? __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true)
                                   : __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true);
          const int shi = o == 1   ? __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true)
                          : o == 2 ? __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true)
                          : o == 4 ? __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true)
                                   : __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true);
          pre += __hiloint2double(shi, slo);
        }
        if (tid < n) offs[DIR == 0 ? f0 + tid + 1 : f0 + n - 1 - tid] = cum + pre;
        chunk_log2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pre), 15),
                                      __builtin_amdgcn_readlane(__double2loint(pre), 15));  // (lanes >= n added 0)
        if constexpr (MITM) corrv = exp2(cum + pre + oo_cur - zlog2);  // (zlog2 = +inf, no accepting path: 0)
      }
      // (block-uniform: absent arcs have wf = 0, so any class >= the true degree is exact)
      // (uniform-label acceptors: PER WAVE -- the frame loops differ only in how many of the eight arc slots they read,
      // one barrier per frame in all of them; the alignment graphs of the Transducer have 2-5 arcs into most states and
      // 8 into a few: waves that do not hold such a state read 4 or 6 vector entries per state and frame, not 8)
      // (even classes only: the frame loops take the arc slots in pairs)
#ifdef WFL_SWEEP_PHASE_TIMERS
      const long long eD = clock64();
      e_offs += eD - eC;
#endif
      if constexpr (!LIVE) {  // only the barriers (the frame loops below: one per frame, one more in front of beta's)
        if (DIR == 1 && uniform) lds_barrier();
        for (int i = 0; i < n; ++i) {
          lds_barrier();
          if (eps_on)
            for (int step = 1; step < nlev; ++step) lds_barrier();  // (eps_closure's)
        }
        hand_over(PathIdle{});
      } else if constexpr (SEL == 0)
        frames_uniform(std::integral_constant<int, 2>{});
      else if constexpr (SEL == 1)
        frames_uniform(std::integral_constant<int, 4>{});
      else if constexpr (SEL == 2)
        frames_uniform(std::integral_constant<int, 6>{});
      else if constexpr (SEL == 3)
        frames_uniform(std::integral_constant<int, kLeanDeg>{});
      else if constexpr (SEL == 4)
        frames(std::integral_constant<int, 2>{});
      else if constexpr (SEL == 5)
        frames(std::integral_constant<int, 4>{});
      else
        frames(std::integral_constant<int, kLeanDeg>{});
#ifdef WFL_SWEEP_PHASE_TIMERS
      const long long eE = clock64();
      e_fr += eE - eD;
#endif
      cum += chunk_log2;
      if (c + 1 < nchunks) {
        if (PUB) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // (see below)
        __syncthreads();
        // Chunk c - 1 is in L2: its stores are older than this chunk's prefetch loads and its (at most 16) frame stores,
        // vmcnt counts loads and stores in issue order, and every thread has just waited until at most 16 of its
        // operations were outstanding.  (One chunk of lag costs the gradient nothing; waiting for THIS chunk's stores
        // would put a store round trip, ~1.5 us, behind every 16 frames.)
        if (PUB && tid == 0 && c > 0) prog_publish(prog, token, (uint32_t)c | (MITM || mitm_on ? kProgMitm : 0u));
      }
      if constexpr (MITM) oo_cur = oo_next;
#ifdef WFL_SWEEP_PHASE_TIMERS
      e_hand += clock64() - eE;
#endif
    };
    // meeting in the middle: the launch asks, a uniform-label acceptor, at least two whole chunks a side.  c0: the chunks
    // this sweep stores as plain vectors.  The forward sweep's reach m_f = 16 c0 (the middle slot); the backward sweep's
    // chunks end at T - 16 c, so it goes on to m_b = the first such boundary <= m_f: with T a multiple of 16 the two
    // coincide, otherwise both sweeps hold plain vectors of slots m_b .. m_f and the forward sweep turns all of them
    // into occupancies at its crossing (gamma_rows).  The occupancy half is whole chunks; a last, partial chunk stores
    // plain vectors as ever and is turned into occupancies afterwards (below).
    // (every wave of the workgroup takes the same decision and the crossing's barriers: `uniform` is block-uniform, the
    // waves without a state -- SEL 7 -- included)
    const bool mitm_try = PUB && uniform && mm.req != 0 && nfull >= 4;
    const int c0 = DIR == 0 ? mitm_plain_chunks_a(T) : mitm_plain_chunks_b(T);
    int cfirst = nfull;  // (first chunk of the occupancy half, if it comes to that)
    for (int c = 0; c < nfull; ++c) {
      if constexpr (PUB) {
        if (mitm_try && c == c0) {
          // ---- the crossing.  Everything this sweep stored so far must be in L2 before the partner is told so.
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          int* red = (int*)lred;
          if (tid == 0) {
            prog_publish(prog, token, (uint32_t)c0);
            const uint32_t need = (uint32_t)(DIR == 0 ? mitm_plain_chunks_b(T) : mitm_plain_chunks_a(T)), me = xcc_id();
            int st = 0;
            for (int spin = 0; spin < mm.spins; ++spin) {
              const uint64_t v = __hip_atomic_load(mm.oth_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((uint32_t)(v >> 32) == token) {
                const uint32_t fld = (uint32_t)v & 0x0fffffffu;
                if (fld == kProgSkip || (((uint32_t)v >> 28) & 15u) != me) break;  // not swept / another XCD's L2: plain
                if ((fld & kProgCount) >= need) {
                  st = 1;
                  break;
                }
              }
              __builtin_amdgcn_s_sleep(16);
            }
            red[0] = st;
          }
          __syncthreads();
          const bool ok = red[0] != 0;
          __syncthreads();
          if (ok) {
            const int m = DIR == 0 ? 16 * c0 : T - 16 * c0;  // the middle slot: both sweeps hold their vector of it
            const double bm = ld_l2(mm.oth + (int64_t)m * Q + tidc);
            const double om = ld_l2(mm.oth_offs + m);
            const double tot = block_reduce_sum_f64(tid < Q ? p * bm : 0.0, (double*)lred);
            const bool deadz = !(tot > 0.0 && tot < 1.0e300);
            zlog2 = deadz ? __builtin_inf() : log2(tot) + cum + om;
            // gamma at the middle slot itself: the forward sweep's, into the beta buffer's row (nobody reads beta_m again:
            // the partner holds it in registers, and every thread of this workgroup has used its copy -- the reduction's barriers)
            if (DIR == 0 && tid < Q)
              reinterpret_cast<float*>(mm.oth + (int64_t)m * Q)[tid] = deadz ? 0.f : (float)(p * bm * exp2(cum + om - zlog2));
            // (T not a multiple of 16: slots m_b .. m - 1 as well, from the two sweeps' stored vectors)
            if (DIR == 0 && (T & 15) != 0) gamma_rows(m - 16 + (T & 15), 16 - (T & 15), mm.oth);
            if constexpr (LIVE) {
#pragma unroll
              for (int j = 0; j < 8; ++j) pbA[j] = ld_l2(mm.oth + (int64_t)slot_of(c0, j) * Q + tidc);
              oo_cur = ld_l2(mm.oth_offs + slot_of(c0, tid & 15));
            }
            if (tid == 0) {
              if (DIR == 0)
                *mm.fmt = kFmtOcc;
              else
                __hip_atomic_store(mm.mitm_b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (vmcnt(0) as the BUILTIN, which the compiler's wait-count pass sees: the values just requested are used by
            // the chunk loop below, whose steady state leaves them pending across its back edge with ~36 younger
            // operations behind them; entering the loop with them pending and only a handful behind, the static pass
            // sized every chunk's first wait for THAT path -- vmcnt(12): the previous chunk's stores, 1.4 us a chunk)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            mitm_on = true;
            cfirst = c0;
            break;
          }
        }
      }
      chunk(c, std::true_type{}, std::false_type{});
    }
    if constexpr (PUB) {
      for (int c = cfirst; c < nfull; ++c) chunk(c, std::true_type{}, std::true_type{});
    }
    for (int c = nfull; c < nchunks; ++c) chunk(c, std::false_type{}, std::false_type{});
    if constexpr (PUB) {
      if (mitm_on && nfull < nchunks) {
        // the last, partial chunk (alpha: slots 16 nfull + 1 .. T, beta: T % 16 - 1 .. 0) was stored as plain vectors:
        // occupancies in their place, once its stores -- vectors and offsets -- are in L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        gamma_rows(DIR == 0 ? 16 * nfull + 1 : 0, T - 16 * nfull, out + u.ab_base);
      }
    }
  };
  if (!wave_live)
    sweep(std::integral_constant<int, 7>{}, std::false_type{});
  else if (uniform && wave_deg <= 2)
    sweep(std::integral_constant<int, 0>{}, std::true_type{});
  else if (uniform && wave_deg <= 4)
    sweep(std::integral_constant<int, 1>{}, std::true_type{});
  else if (uniform && wave_deg <= 6)
    sweep(std::integral_constant<int, 2>{}, std::true_type{});
  else if (uniform)
    sweep(std::integral_constant<int, 3>{}, std::true_type{});
  else if (deg_class == 0)
    sweep(std::integral_constant<int, 4>{}, std::true_type{});
  else if (deg_class == 1)
    sweep(std::integral_constant<int, 5>{}, std::true_type{});
  else
    sweep(std::integral_constant<int, 6>{}, std::true_type{});
#ifdef WFL_SWEEP_PHASE_TIMERS
  if (tid == 0 && b == 0 && !banded) printf("dir %d: loop %lld = prefetch %lld + renorm %lld + offsets %lld + frames %lld + hand-over %lld (%d chunks)\n", DIR, clock64() - exp_begin, e_pre, e_ren, e_offs, e_fr, e_hand, nchunks);
#endif
  if (PUB) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) prog_publish(prog, token, (uint32_t)max(nchunks, 1) | (mitm_on ? kProgMitm : 0u));
  }
  // log2 of the total: alpha over the accept states at slot T, beta over the start states at slot 0
  {
    const float bw = tid < Q ? (DIR == 0 ? u.accept_w[tid] : u.start_w[tid]) : WFL_NEG_INF;
    const double tot = block_reduce_sum_f64(bw > WFL_NEG_INF ? p : 0.0, (double*)lred);
    if (tid == 0) {
      const bool ok = tot > 0.0 && tot < 1.0e300;
      const double z2 = ok ? log2(tot) + cum : (tot == 0.0 ? -__builtin_inf() : __builtin_nan(""));
      z64[b] = z2;  // log2 Z as this sweep sees it (the certificate compares the two)
      if (DIR == 0 && logz) logz[b] = (float)(z2 * 0.6931471805599453094);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stage 2, probability domain, ANY acceptor that fits LDS (round 6).  What run_chain_prob does not take -- states
// with more than kLeanDeg labelled or kEpsDeg epsilon arcs on a side, more than kProbMaxLev closure levels, more states than
// threads: the transition models themselves (the DENOMINATOR of a Transducer with `transitions=`: 8 states and 36 arcs
// for the reference's pruned back-off model, thousands of arcs for a word-piece n-gram) -- went to the log-domain
// run_chain: a log-add per arc and a closure whose levels are chains of double-precision exp / log, 2.8 us a frame
// for the 8-state model.  Here the same acceptor layout in LDS (arcs in this direction's CSR order, epsilon arcs,
// levels), double PROBABILITIES, one multiply-add per arc:
//   * a state with at most kProbRowDeg arcs is relaxed by one thread, the others by a row of 16 lanes each (a lane takes
//     every 16th arc, the row meets through row16_sum);
//   * epsilon closure level by level as in run_chain, an LDS-only barrier per level;
//   * offsets per slot in log2 units and the power-of-two renormalisation of run_chain_prob (at every chunk), the same
//     stored format (fmt[b] = kFmtProb), so the certificate and the gradient kernels read it like any other.
// Not tuned beyond that: no register-resident arcs, no straight-line chunks.
// ------------------------------------------------------------------------------------------------
constexpr int kProbRowDeg = 6;  // most labelled arcs a single thread takes
constexpr int kProbEpsDeg = 8;  // most epsilon arcs a single thread takes
// (inlined into its kernel: a function that is really called knows nothing of its caller's launch bounds and is compiled
// for 1024 threads, i.e. 128 VGPRs -- 127 spill instructions, some of them in the frame loop, where a scratch load
// waits for the frame's global stores)
template <int DIR>
__device__ __forceinline__ void run_chain_prob_general_body(const wfl_lattice_desc& d, const UttView& u, char* smem, int T, int R,
                                                    const float* __restrict__ fg, const float* __restrict__ rmax,
                                                    const float* __restrict__ weights, double* __restrict__ out,
                                                    float* __restrict__ logz, int b, double* __restrict__ offs,
                                                    double* __restrict__ z64, float* __restrict__ wref_out) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int Q = u.Q, A = u.A, E = u.E, nlev = u.nlev, Kmax = d.max_labels;
  // (carved like chain_kernel's ChainLds: chain_lds_bytes has room for exactly this)
  char* sp = smem;
  int2* arcs = (int2*)sp;
  sp += (size_t)d.max_arcs * 8;
  int2* eps = (int2*)sp;
  sp += (size_t)d.max_eps * 8;
  int* ptr = (int*)sp;
  sp += (size_t)(d.max_states + 1) * 4;
  int* eptr = (int*)sp;
  sp += (size_t)(d.max_states + 1) * 4;
  double* buf0 = (double*)sp;
  sp += (size_t)d.max_states * 8;
  double* buf1 = (double*)sp;
  sp += (size_t)d.max_states * 8;
  float* rows = (float*)sp;
  sp += (size_t)2 * R * Kmax * 4;
  float* red = (float*)sp;
  sp += 64 * 4;
  int* lvl = (int*)sp;
  sp += (size_t)(d.max_levels + 1) * 4;
  int* heavy = (int*)sp;  // [Q] states relaxed by a row, heavy[Q] = their count; then the same for the closure
  sp += (size_t)(d.max_states + 1) * 4;
  float* refs = (float*)sp;  // [2][R] per-frame references of the chunk's rows

  // ---- the reference of the labelled arcs' weights (every frame multiplies by e^wref once more: part of the offset)
  float wmx = WFL_NEG_INF;
  for (int k = tid; k < A; k += NT) {
    float w = u.arc_w[k];
    const int wid = u.arc_wid[k];
    if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
    wmx = fmaxf(wmx, nan_to_neg(w));
  }
  float wref = block_reduce_max(wmx, red);
  if (!(wref > WFL_NEG_INF)) wref = 0.f;
  if (wref_out && tid == 0) wref_out[b] = wref;
  // ---- stage the acceptor in this direction's CSR order; weights as factors (the gradient kernel forms the same floats)
  for (int k = tid; k < A; k += NT) {
    const int a = DIR == 0 ? k : u.out_arc[k];
    const int other = DIR == 0 ? u.arc_src[a] : u.arc_dst[a];
    float w = u.arc_w[a];
    const int wid = u.arc_wid[a];
    if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
    arcs[k] = make_int2(other | (u.arc_slot[a] << 16), __float_as_int(fast_exp(nan_to_neg(w) - wref)));
  }
  for (int k = tid; k < E; k += NT) {
    const int e = DIR == 0 ? k : u.eout_arc[k];
    eps[k] = make_int2(DIR == 0 ? u.eps_src[e] : u.eps_dst[e], __float_as_int((float)eps_factor(u, weights, e)));
  }
  for (int q = tid; q <= Q; q += NT) {
    ptr[q] = DIR == 0 ? u.in_ptr[q] : u.out_ptr[q];
    eptr[q] = DIR == 0 ? u.ein_ptr[q] : u.eout_ptr[q];
  }
  for (int l = tid; l <= nlev; l += NT) lvl[l] = u.lvl_ptr[l];
  if (tid == 0) heavy[Q] = 0;
  __syncthreads();
  for (int q = tid; q < Q; q += NT)
    if (ptr[q + 1] - ptr[q] > kProbRowDeg) heavy[atomicAdd(&heavy[Q], 1)] = q;
  int eh = 0;
  for (int q = tid; q < Q; q += NT) eh |= (eptr[q + 1] - eptr[q]) > kProbEpsDeg;
  const bool eps_heavy = __syncthreads_or(eh) != 0;
  const int n_heavy = heavy[Q];
  const int grow = tid >> 4, nrows = NT >> 4, r16 = tid & 15;
  // The common case in registers: this thread's own state (q = tid) with its few arcs, this row's first state of many
  // arcs with four arcs per lane, this thread's epsilon arcs and level -- in the frame loop they cost one round of
  // LDS reads (the vector and the row) instead of a chain of four (list -> pointers -> arcs -> vector).  What does
  // not fit (more states than threads, more rows than the workgroup has, arcs beyond 64 a row) takes the loops over
  // the LDS copy.  (absent arcs: factor 0, entry 0)
  // (decoded once: source entry, row slot and the factor as a double -- two waves that each run alone on their SIMD
  // issue an instruction every ~5 cycles, and the masks, shifts and conversions were a third of a frame's)
  struct RegArc {
    int src, slot;
    double fac;
  };
  auto decode = [](bool on, int2 a) {
    RegArc r;
    r.src = on ? (a.x & 0xffff) : 0, r.slot = on ? (int)((unsigned)a.x >> 16) : 0;
    r.fac = on ? (double)__int_as_float(a.y) : 0.0;
    return r;
  };
  const int lk0 = tid < Q ? ptr[tid] : 0, lk1 = tid < Q ? ptr[tid + 1] : 0;
  const bool light0 = tid < Q && lk1 - lk0 <= kProbRowDeg;
  RegArc la[kProbRowDeg];
#pragma unroll
  for (int i = 0; i < kProbRowDeg; ++i) la[i] = decode(light0 && lk0 + i < lk1, arcs[min(lk0 + i, max(A - 1, 0))]);
  const int hq0 = grow < n_heavy ? heavy[grow] : -1;
  const int hk0 = hq0 >= 0 ? ptr[hq0] : 0, hk1 = hq0 >= 0 ? ptr[hq0 + 1] : 0;
  RegArc ha[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ha[j] = decode(hk0 + r16 + 16 * j < hk1, arcs[min(hk0 + r16 + 16 * j, max(A - 1, 0))]);
  const int ek0 = tid < Q ? eptr[tid] : 0, ek1 = tid < Q ? eptr[tid + 1] : 0;
  const bool elight0 = ek1 > ek0 && ek1 - ek0 <= kProbEpsDeg;
  int ea_src[kProbEpsDeg];
  double ea_fac[kProbEpsDeg];
#pragma unroll
  for (int i = 0; i < kProbEpsDeg; ++i) {
    const bool on = elight0 && ek0 + i < ek1;
    const int2 a = eps[min(ek0 + i, max(E - 1, 0))];
    ea_src[i] = on ? a.x : 0, ea_fac[i] = on ? (double)__int_as_float(a.y) : 0.0;
  }
  // (no wave-uniform bounds on those slots: a test between two groups of LDS reads makes each group its own round trip --
  // measured: the relaxation 740 -> 1200-1800 cycles a frame)
  const int wh = __builtin_amdgcn_readfirstlane(wave_all_max_int(hq0 >= 0 ? 1 : 0));
  const bool eheavy0 = ek1 - ek0 > kProbEpsDeg;  // this thread's state is closed by a row: its value is read back
  int my_lev = 0;
  for (int l = 1; l < nlev; ++l)
    if (tid >= lvl[l]) my_lev = l;
  auto term = [](const double* vec, const float* row, const RegArc& a) { return vec[a.src] * (a.fac * (double)row[a.slot]); };
  auto term_lds = [](const double* vec, const float* row, int2 a) {
    return vec[a.x & 0xffff] * ((double)__int_as_float(a.y) * (double)row[(unsigned)a.x >> 16]);
  };

  // epsilon closure of `vals` (run_chain's order of levels); enters and ends behind a barrier
  // (`mine`: the value of this thread's own state q = tid, kept in a register across the levels)
  auto closure = [&](double* vals, double& mine) {
    for (int step = 1; step < nlev; ++step) {
      const int lev = DIR == 0 ? step : nlev - 1 - step;
      if (elight0 && my_lev == lev) {
        double v0 = mine, v1 = 0.0;
#pragma unroll
        for (int i = 0; i < kProbEpsDeg; i += 2) {
          v0 = fma(vals[ea_src[i]], ea_fac[i], v0);
          v1 = fma(vals[ea_src[i + 1]], ea_fac[i + 1], v1);
        }
        mine = v0 + v1;
        vals[tid] = mine;
      }
      for (int q = max(lvl[lev], NT) + tid; q < lvl[lev + 1]; q += NT) {  // (more states than threads)
        const int k0 = eptr[q], k1 = eptr[q + 1];
        if (k1 - k0 > kProbEpsDeg || k1 == k0) continue;
        double v = vals[q];
        for (int k = k0; k < k1; ++k) v = fma(vals[eps[k].x], (double)__int_as_float(eps[k].y), v);
        vals[q] = v;
      }
      if (eps_heavy) {
        for (int q0 = lvl[lev]; q0 < lvl[lev + 1]; q0 += nrows) {  // (uniform trip count: the rows of a wave meet)
          const int q = q0 + grow;
          const bool on = q < lvl[lev + 1];
          const int k0 = on ? eptr[q] : 0, k1 = on ? eptr[q + 1] : 0;
          double part = 0.0;
          if (k1 - k0 > kProbEpsDeg)
            for (int k = k0 + r16; k < k1; k += 16) part = fma(vals[eps[k].x], (double)__int_as_float(eps[k].y), part);
          part = row16_sum_dpp(part);
          if (k1 - k0 > kProbEpsDeg && r16 == 0) vals[q] += part;
        }
      }
      lds_barrier();
    }
    if (eheavy0 && nlev > 1) mine = vals[tid];
  };

  const int t_first = DIR == 0 ? 0 : T;
  double* cur = (t_first & 1) ? buf1 : buf0;
  for (int q = tid; q < Q; q += NT) cur[q] = (DIR == 0 ? u.start_w[q] : u.accept_w[q]) > WFL_NEG_INF ? 1.0 : 0.0;
  lds_barrier();
  {
    double mine = tid < Q ? cur[tid] : 0.0;
    closure(cur, mine);
  }
  for (int q = tid; q < Q; q += NT) out[u.ab_base + (int64_t)t_first * Q + q] = cur[q];
  if (tid == 0) offs[t_first] = 0.0;
  double cum = 0.0;  // log2 of everything factored out of the stored probabilities so far

  const int nchunks = (T + R - 1) / R;
  auto chunk_frames = [&](int c, int& f0, int& n) {  // frames [f0, f0 + n) in ascending order
    const int s0 = c * R;
    n = min(R, T - s0);
    f0 = DIR == 0 ? s0 : T - s0 - n;
  };
  if (T > 0) {
    int f0, n;
    chunk_frames(0, f0, n);
    const float* src = fg + u.xg_base + (int64_t)f0 * Kmax;
    for (int e = tid; e < n * Kmax; e += NT) rows[e] = src[e];
    if (tid < n) refs[tid] = rmax[(int64_t)b * T + f0 + tid];
  }
  __syncthreads();
#ifdef WFL_GENERAL_TIMERS
  long long gen_t[6] = {0, 0, 0, 0, 0, 0};
#define GEN_T(v) const long long v = clock64()
#define GEN_ADD(k, a, b_) gen_t[k] += (b_) - (a)
#else
#define GEN_T(v)
#define GEN_ADD(k, a, b_)
#endif
  for (int c = 0; c < nchunks; ++c) {
    GEN_T(gc0);
    int f0, n;
    chunk_frames(c, f0, n);
    const float* tile = rows + (size_t)(c & 1) * R * Kmax;
    const float* rtile = refs + (size_t)(c & 1) * R;
    float pre[kPre], rpre = 0.f;
    int pf0 = 0, pn = 0;
    if (c + 1 < nchunks) {  // the next chunk's rows: HBM -> registers now, -> LDS behind this chunk's frames
      chunk_frames(c + 1, pf0, pn);
      const float* src = fg + u.xg_base + (int64_t)pf0 * Kmax;
#pragma unroll
      for (int j = 0; j < kPre; ++j) {
        const int e = tid + j * NT;
        if (e < pn * Kmax) pre[j] = src[e];
      }
      if (tid < pn) rpre = rmax[(int64_t)b * T + pf0 + tid];
    }
    if (c > 0) {  // power-of-two renormalisation of the vector the chunk starts from (exact)
      double* fromb = ((DIR == 0 ? f0 : f0 + n) & 1) ? buf1 : buf0;
      int ex = -(1 << 30);
      for (int q = tid; q < Q; q += NT)
        if (fromb[q] > 0.0) ex = max(ex, __builtin_amdgcn_frexp_exp(fromb[q]) - 1);
      const float emf = block_reduce_max((float)ex, red);  // (|ex| < 2^24: exact as a float)
      if (emf > -1.0e9f && emf < 2000.f) {
        const int emax = (int)emf;
        for (int q = tid; q < Q; q += NT) fromb[q] = ldexp(fromb[q], -emax);
        cum += (double)emax;
      }
      __syncthreads();
    }
    for (int i = 0; i < n; ++i) {
      // forward: consume frame t, produce slot t + 1.  backward: consume frame t, produce slot t.
      const int t = DIR == 0 ? f0 + i : f0 + n - 1 - i;
      const int slot_from = DIR == 0 ? t : t + 1, slot_to = DIR == 0 ? t + 1 : t;
      const double* from = (slot_from & 1) ? buf1 : buf0;
      double* to = (slot_to & 1) ? buf1 : buf0;
      const float* row = tile + (size_t)(t - f0) * Kmax;
      const float ref_t = rtile[t - f0];  // (read here: its LDS round trip is off the frame's chain of them)
      GEN_T(g0);
      double mine = 0.0;
      if (light0) {
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int i = 0; i < kProbRowDeg; i += 2) v0 += term(from, row, la[i]), v1 += term(from, row, la[i + 1]);
        mine = v0 + v1;
        to[tid] = mine;
      }
      if (wh > 0) {  // (wave-uniform: some row of this wave has a state)
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j += 2) v0 += term(from, row, ha[j]), v1 += term(from, row, ha[j + 1]);
        for (int k = hk0 + r16 + 64; k < hk1; k += 16) v0 += term_lds(from, row, arcs[k]);
        const double v = row16_sum_dpp(v0 + v1);
        if (hq0 >= 0 && r16 == 0) to[hq0] = v;
      }
      for (int q = tid + NT; q < Q; q += NT) {  // (more states than threads)
        const int k0 = ptr[q], k1 = ptr[q + 1];
        if (k1 - k0 > kProbRowDeg) continue;
        double v0 = 0.0, v1 = 0.0;
        int k = k0;
        for (; k + 1 < k1; k += 2) {
          const int2 a0 = arcs[k], a1 = arcs[k + 1];
          v0 = fma(from[a0.x & 0xffff], (double)__int_as_float(a0.y) * (double)row[(unsigned)a0.x >> 16], v0);
          v1 = fma(from[a1.x & 0xffff], (double)__int_as_float(a1.y) * (double)row[(unsigned)a1.x >> 16], v1);
        }
        if (k < k1) {
          const int2 a0 = arcs[k];
          v0 = fma(from[a0.x & 0xffff], (double)__int_as_float(a0.y) * (double)row[(unsigned)a0.x >> 16], v0);
        }
        to[q] = v0 + v1;
      }
      for (int h0 = nrows; h0 < n_heavy; h0 += nrows) {  // (more states of many arcs than rows; uniform trip count)
        const int hq = h0 + grow;
        const int q = hq < n_heavy ? heavy[hq] : -1;
        const int k0 = q >= 0 ? ptr[q] : 0, k1 = q >= 0 ? ptr[q + 1] : 0;
        double v0 = 0.0, v1 = 0.0;
        int k = k0 + r16;
        for (; k + 16 < k1; k += 32) {
          const int2 a0 = arcs[k], a1 = arcs[k + 16];
          v0 = fma(from[a0.x & 0xffff], (double)__int_as_float(a0.y) * (double)row[(unsigned)a0.x >> 16], v0);
          v1 = fma(from[a1.x & 0xffff], (double)__int_as_float(a1.y) * (double)row[(unsigned)a1.x >> 16], v1);
        }
        if (k < k1) {
          const int2 a0 = arcs[k];
          v0 = fma(from[a0.x & 0xffff], (double)__int_as_float(a0.y) * (double)row[(unsigned)a0.x >> 16], v0);
        }
        const double v = row16_sum_dpp(v0 + v1);
        if (q >= 0 && r16 == 0) to[q] = v;
      }
      GEN_T(g1);
      lds_barrier();
      GEN_T(g2);
      if (!light0 && tid < Q) mine = to[tid];  // (a row's lane 0 wrote it)
      closure(to, mine);
      GEN_T(g3);
      cum += ((double)ref_t + (double)wref) * kLog2e_d;
      if (tid == 0) offs[slot_to] = cum;
      double* orow = out + u.ab_base + (int64_t)slot_to * Q;
      if (tid < Q) orow[tid] = mine;
      if (Q > NT)
        for (int q = tid + NT; q < Q; q += NT) orow[q] = to[q];
      GEN_T(g4);
      GEN_ADD(0, g0, g1);
      GEN_ADD(1, g1, g2);
      GEN_ADD(2, g2, g3);
      GEN_ADD(3, g3, g4);
    }
    GEN_T(g5);
    if (c + 1 < nchunks) {
      float* dst = rows + (size_t)((c + 1) & 1) * R * Kmax;
#pragma unroll
      for (int j = 0; j < kPre; ++j) {
        const int e = tid + j * NT;
        if (e < pn * Kmax) dst[e] = pre[j];
      }
      if (tid < pn) refs[(size_t)((c + 1) & 1) * R + tid] = rpre;
      __syncthreads();
    }
    GEN_T(g6);
    GEN_ADD(4, g5, g6);
    GEN_ADD(5, gc0, g5);
  }
#ifdef WFL_GENERAL_TIMERS
  if (tid == 0 && b == 0) printf("general dir %d: relax %lld barrier %lld closure %lld store %lld hand-over %lld chunk-total %lld\n", DIR, gen_t[0], gen_t[1], gen_t[2], gen_t[3], gen_t[4], gen_t[5]);
#endif
  // log2 of the total: alpha over the accept states at slot T, beta over the start states at slot 0
  {
    const double* fin = ((DIR == 0 ? T : 0) & 1) ? buf1 : buf0;
    double part = 0.0;
    for (int q = tid; q < Q; q += NT)
      if ((DIR == 0 ? u.accept_w[q] : u.start_w[q]) > WFL_NEG_INF) part += fin[q];
    const double tot = block_reduce_sum_f64(part, (double*)red);
    if (tid == 0) {
      const bool ok = tot > 0.0 && tot < 1.0e300;
      const double z2 = ok ? log2(tot) + cum : (tot == 0.0 ? -__builtin_inf() : __builtin_nan(""));
      z64[b] = z2;
      if (DIR == 0 && logz) logz[b] = (float)(z2 * 0.6931471805599453094);
    }
  }
}

// (the 1024-thread kernel has 128 VGPRs either way: there the general sweep stays a call, and its spills its own)
template <int DIR, bool INLINE>
__device__ __forceinline__ void run_chain_prob_general(const wfl_lattice_desc& d, const UttView& u, char* smem, int T, int R,
                                                       const float* __restrict__ fg, const float* __restrict__ rmax,
                                                       const float* __restrict__ weights, double* __restrict__ out,
                                                       float* __restrict__ logz, int b, double* __restrict__ offs,
                                                       double* __restrict__ z64, float* __restrict__ wref_out) {
  if constexpr (INLINE) {
    run_chain_prob_general_body<DIR>(d, u, smem, T, R, fg, rmax, weights, out, logz, b, offs, z64, wref_out);
  } else {
    auto call = [&]() __attribute__((noinline)) {
      run_chain_prob_general_body<DIR>(d, u, smem, T, R, fg, rmax, weights, out, logz, b, offs, z64, wref_out);
    };
    call();
  }
}

// The probability-domain sweeps as their own kernel (their register budget is not the general path's): utterances it
// does not take are left to the log-domain launch that follows (chain_kernel, mode 2).
template <int MAXT, bool PUB>
__device__ __forceinline__ void prob_chain_body(const wfl_lattice_desc& d, const int32_t* __restrict__ ints,
                                                const float* __restrict__ floats, const float* __restrict__ xg, int T,
                                                int rows_per_chunk, const float* __restrict__ weights,
                                                float* __restrict__ alpha, float* __restrict__ beta,
                                                float* __restrict__ logz, int64_t tail, int nch1, int b, int dir, char* smem,
                                                uint32_t token, int mitm_req = 0) {
  const UttView u = make_view(d, ints, floats, b, T);
  double* offs_a = reinterpret_cast<double*>(alpha + tail);  // (tail layout: see chain_kernel)
  double* offs_b = beta ? reinterpret_cast<double*>(beta + tail) : nullptr;
  double* za = offs_a + (int64_t)d.B * nch1;
  double* zb = offs_b ? offs_b + (int64_t)d.B * nch1 : nullptr;
  int32_t* fmt = reinterpret_cast<int32_t*>(za + d.B);
  float* wrefs = reinterpret_cast<float*>(fmt + d.B);
  uint64_t* prog = PUB ? reinterpret_cast<uint64_t*>((dir == 0 ? offs_a : offs_b) + prog_offset_doubles(d, nch1)) + b : nullptr;
  const float* fg = xg + xg_main_dev(d, T);
  const float* rmax = fg + xg_main_dev(d, T);
  if (!prob_eligible(u, blockDim.x)) {
    if (PUB) {  // (the launch with the gradient beside the sweeps: left to the log-domain launch that follows)
      if (threadIdx.x == 0) {
        prog_publish(prog, token, kProgSkip);
        if (dir == 0) fmt[b] = kFmtLog;
      }
      return;
    }
    // any other acceptor that fits LDS: the general probability-domain sweep (same stored format, same certificate)
    if (dir == 0) {
      if (threadIdx.x == 0) fmt[b] = kFmtProb;
      run_chain_prob_general<0, MAXT != 1024>(d, u, smem, T, rows_per_chunk, fg, rmax, weights, reinterpret_cast<double*>(alpha), logz, b,
                                offs_a + (int64_t)b * nch1, za, wrefs);
    } else {
      if (threadIdx.x == 0) zb[d.B + b] = 0.0;
      run_chain_prob_general<1, MAXT != 1024>(d, u, smem, T, rows_per_chunk, fg, rmax, weights, reinterpret_cast<double*>(beta), nullptr, b,
                                offs_b + (int64_t)b * nch1, zb, nullptr);
    }
    return;
  }
  if (PUB && threadIdx.x == 0) prog_publish(prog, token, 0u);  // "resident" (occ_gate_kernel waits for it)
  ProbLds P;
  char* p = smem;
  const size_t nvec = max((size_t)d.max_states, (size_t)blockDim.x);  // (one entry per THREAD: every lane may write)
  P.buf0 = (double*)p, p += nvec * 8;
  P.buf1 = (double*)p, p += nvec * 8;
  P.red = (float*)p, p += 64 * 4;
  P.rows = (float*)p, p += prob_rows_floats(d, rows_per_chunk, blockDim.x) * 4;
  P.refs = (float*)p;
  MitmArgs mm;
  if (PUB && beta) {
    // (the partner: the other direction's rows, offsets and progress word; the backward sweep's own flag sits in the
    // beta buffer where the alpha buffer keeps the gradient's `bad` words: occ_header)
    mm.req = mitm_req;
    mm.oth = reinterpret_cast<double*>(dir == 0 ? beta : alpha) + u.ab_base;
    mm.oth_offs = (dir == 0 ? offs_b : offs_a) + (int64_t)b * nch1;
    mm.oth_prog = reinterpret_cast<const uint64_t*>((dir == 0 ? offs_b : offs_a) + prog_offset_doubles(d, nch1)) + b;
    mm.fmt = fmt + b;
    mm.mitm_b = occ_header(d, beta, tail, nch1).bad + b;
  }
  if (dir == 0) {
    if (threadIdx.x == 0) fmt[b] = kFmtProb;
    run_chain_prob<0, MAXT == 128, MAXT <= 512, PUB>(d, u, P, T, rows_per_chunk, fg, rmax, weights, reinterpret_cast<double*>(alpha), logz, b,
                      offs_a + (int64_t)b * nch1, za, wrefs, offs_a + (int64_t)d.B * nch1 + 2 * (int64_t)d.B, prog, token, mm);
  } else {
    if (threadIdx.x == 0) {
      zb[d.B + b] = 0.0;  // the certificate's verdict: raised by prob_certify_kernel
      if (PUB) __hip_atomic_store(mm.mitm_b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    run_chain_prob<1, MAXT == 128, MAXT <= 512, PUB>(d, u, P, T, rows_per_chunk, fg, rmax, weights, reinterpret_cast<double*>(beta), nullptr, b,
                      offs_b + (int64_t)b * nch1, zb, nullptr, offs_b + (int64_t)d.B * nch1 + 2 * (int64_t)d.B, prog, token, mm);
  }
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT)
    prob_chain_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats,
                      const float* __restrict__ xg, int T, int rows_per_chunk, const float* __restrict__ weights,
                      float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ logz, int64_t tail,
                      int nch1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef WFL_BAND_PRIO
#define WFL_BAND_PRIO 0
#endif
  if (MAXT == 128 && WFL_BAND_PRIO) __builtin_amdgcn_s_setprio(WFL_BAND_PRIO);
  prob_chain_body<MAXT, false>(d, ints, floats, xg, T, rows_per_chunk, weights, alpha, beta, logz, tail, nch1, blockIdx.x,
                               blockIdx.y, smem, 0u);
}

// Certificate of the probability-domain sweeps.  A double holds a spread of 2^1000 between the largest state and
// the ones that carry the posteriors; beyond that the latter are flushed, in BOTH sweeps (comparing the two totals is
// not enough: each sweep can lose a different half of the paths and the halves can weigh the same).  What cannot
// fail silently is sum_s alpha_t[s] beta_t[s] = Z at every time slot: checked at every 8th slot (a flushed region
// spans many frames) plus both ends, against the forward sweep's log2 Z, to 1e-4 (the parity bar).  One workgroup per
// utterance; verdict[b] = 1 sends the utterance to the log-domain launch that follows.
__global__ void __launch_bounds__(256)
    prob_certify_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats, int T,
                        const float* __restrict__ alpha, float* __restrict__ beta, int64_t tail, int nch1, int chain_nt,
                        int pub,  // pub: the sweeps were prob_chain_pub_kernel's (they may have met in the middle)
                        const float* __restrict__ weights) {
  // grid (B, kCertSplit): every wave takes the checked slots s = wave index, + number of waves, ... on its own
  // (wave-level reductions only); a wave that finds a violation raises the utterance's verdict (cleared by the beta
  // sweep of prob_chain_kernel before it started)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  WFL_MARK_BEGIN(0);
  WFL_LOG(10);
  const UttView u = make_view(d, ints, floats, b, T);
  const double* offs_a = reinterpret_cast<const double*>(alpha + tail) + (int64_t)b * nch1;
  double* tail_b = reinterpret_cast<double*>(beta + tail);
  const double* offs_b = tail_b + (int64_t)b * nch1;
  const double za = (reinterpret_cast<const double*>(alpha + tail) + (int64_t)d.B * nch1)[b];
  const double zbv = (tail_b + (int64_t)d.B * nch1)[b];
  double* verdict = tail_b + (int64_t)d.B * (nch1 + 1) + b;
  const int32_t* fmt = reinterpret_cast<const int32_t*>(reinterpret_cast<const double*>(alpha + tail) + (int64_t)d.B * (nch1 + 1));
  if (fmt[b] != kFmtProb && fmt[b] != kFmtOcc) {
    if (tid == 0) *verdict = 1.0;  // (not swept in the probability domain at all: the log-domain launch takes it)
    return;
  }
  if (za == -__builtin_inf() && zbv == -__builtin_inf()) return;  // no accepting path: exact in any arithmetic
  const double* pa = reinterpret_cast<const double*>(alpha) + u.ab_base;
  const double* pb = reinterpret_cast<const double*>(beta) + u.ab_base;
  const int Q = u.Q;
  const int nchk = (T + 7) / 8 + 1;  // slots 0, 8, 16, ... and T
  const int w0 = blockIdx.y * 4 + (tid >> 6), nw = gridDim.y * 4;
  bool bad = false;
  // Sweeps that met in the middle (run_chain_prob) stored occupancies, normalised by the Z they formed at the middle slot:
  // the same identity reads sum_q gamma_s[q] = 1 (at slot T that also ties the middle's Z to the forward sweep's
  // total, at slot 0 to the backward sweep's).  One sweep that did and one that did not (a crossing that gave up
  // waiting) leave neither format: re-run in the log domain.
  const bool occ_a = fmt[b] == kFmtOcc;
  const bool occ_b = pub && occ_header(d, beta, tail, nch1).bad[b] == 1u;
  if (occ_a != occ_b) {
    if (tid == 0) *verdict = 1.0;
    return;
  }
  if (occ_a) {
    const int mid = mitm_middle(T);
    for (int c = w0; c < nchk; c += nw) {
      const int t = min(c * 8, T);
      const float* row = reinterpret_cast<const float*>((t <= mid ? pb : pa) + (int64_t)t * Q);
      float sacc = 0.f;
      for (int q = lane; q < Q; q += 64) sacc += row[q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
      const double dev = (sacc > 0.f && sacc < 3.0e38f) ? fabs(log2((double)sacc)) : 1.0e9;
      bad = bad || !(dev <= 1.0e-4);
    }
    if (bad && lane == 0) *verdict = 1.0;
    return;
  }
  // With epsilon arcs a path may pass through several states of a slot; it ARRIVES in exactly one -- by a labelled arc,
  // or at the start: alpha before the closure, i.e. minus what the epsilon arcs into the state brought (a positive
  // double minus part of itself; the back-off state, all of whose mass comes that way, leaves rounding dust).
  const bool eps = u.E > 0;
  for (int c = w0; c < nchk; c += nw) {
    const int t = min(c * 8, T);
    double s = 0.0;
    for (int q = lane; q < Q; q += 64) {
      double a = pa[(int64_t)t * Q + q];
      if (eps)
        for (int e = u.ein_ptr[q]; e < u.ein_ptr[q + 1]; ++e) a -= pa[(int64_t)t * Q + u.eps_src[e]] * eps_factor(u, weights, e);
      s = fma(a, pb[(int64_t)t * Q + q], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double dev = (s > 0.0 && s < 1.0e300) ? fabs(log2(s) + offs_a[t] + offs_b[t] - za) : 1.0e9;
    bad = bad || !(dev <= 1.0e-4);
  }
  if (bad && lane == 0) *verdict = 1.0;
}

template <int SR>
__global__ void chain_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats,
                             const float* __restrict__ xg, int T, int rows_per_chunk,
                             const float* __restrict__ weights, float* __restrict__ alpha, float* __restrict__ beta,
                             int32_t* __restrict__ bptr, float* __restrict__ logz, int64_t tail, int nch1, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, dir = blockIdx.y;
  WFL_MARK_BEGIN(1);
  WFL_LOG(11);
  const UttView u = make_view(d, ints, floats, b, T);
  // behind the score arrays (wfl_lattice_workspace reserves the room):
  //   alpha + tail: double offs[B][nch1], double Z[B] (ln Z; log2 Z for probability-domain utterances),
  //                 int32 fmt[B], float wref[B]
  //   beta + tail:  double offs[B][nch1], double Z[B] as the backward sweep sees it (probability domain), double
  //                 verdict[B] of prob_certify_kernel (0: certified, 1: re-run in the log domain)
  double* offs_a = reinterpret_cast<double*>(alpha + tail);
  double* offs_b = beta ? reinterpret_cast<double*>(beta + tail) : nullptr;
  double* za = offs_a + (int64_t)d.B * nch1;
  double* zb = offs_b ? offs_b + (int64_t)d.B * nch1 : nullptr;
  int32_t* fmt = reinterpret_cast<int32_t*>(za + d.B);
  float* wrefs = reinterpret_cast<float*>(fmt + d.B);
  // mode: 1 = every utterance in the log domain (WFL_LATTICE_DOMAIN=log, tropical semiring);
  //       2 = the launch after prob_chain_kernel: the log-domain sweeps of what that launch left -- utterances whose
  //           acceptor it does not take, and those whose two probability-domain sweeps disagree about Z
  if (SR == WFL_SEMIRING_LOG && mode == 2) {
    if (fmt[b] == kFmtProb || fmt[b] == kFmtOcc) {  // swept in the probability domain by the launch before
      if (!zb) return;                 // (forward only: nothing to compare)
      if (zb[d.B + b] == 0.0) return;  // certified by prob_certify_kernel
    }
  }
  if (SR == WFL_SEMIRING_LOG && dir == 0 && threadIdx.x == 0) fmt[b] = kFmtLog;
  ChainLds L;
  char* p = smem;
  L.arcs = (int2*)p, p += (size_t)d.max_arcs * 8;
  L.eps = (int2*)p, p += (size_t)d.max_eps * 8;
  L.ptr = (int*)p, p += (size_t)(d.max_states + 1) * 4;
  L.eptr = (int*)p, p += (size_t)(d.max_states + 1) * 4;
  L.buf0 = (float*)p, p += (size_t)d.max_states * 8;  // (doubles in the log semiring: ChainVal)
  L.buf1 = (float*)p, p += (size_t)d.max_states * 8;
  L.rows = (float*)p, p += (size_t)2 * rows_per_chunk * d.max_labels * 4;
  L.red = (float*)p, p += 64 * 4;
  L.lvl = (int*)p, p += (size_t)(d.max_levels + 1) * 4;
  L.heavy = (int*)p;
  if (dir == 0)
    run_chain<SR, 0>(d, u, L, T, rows_per_chunk, xg, weights, alpha, bptr, logz, b, offs_a + (int64_t)b * nch1, za);
  else
    run_chain<SR, 1>(d, u, L, T, rows_per_chunk, xg, weights, beta, nullptr, nullptr, b, offs_b + (int64_t)b * nch1,
                     nullptr);
}

// threads of the sweep workgroups: one state per thread up to 1024 states (the lean frame paths need it); beyond
// that threads loop
static int chain_threads(const wfl_lattice_desc& d) {
  int nt = d.max_states <= 128 ? 128 : d.max_states <= 256 ? 256 : d.max_states <= 512 ? 512 : 1024;
  while (nt < 256 && nt * kPre < 2 * d.max_labels) nt += 64;  // two rows per chunk must fit the prefetch registers
  return nt;
}
static size_t chain_lds_bytes(const wfl_lattice_desc& d, int rows_per_chunk) {
  const int nt = chain_threads(d);
  const size_t prob = (size_t)std::max(d.max_states, nt) * 16 + 64 * 4 + prob_rows_floats(d, rows_per_chunk, nt) * 4 +
                      (size_t)2 * nt * 4 + 64;
  return std::max(prob, (size_t)d.max_arcs * 8 + (size_t)d.max_eps * 8 + (size_t)(d.max_states + 1) * 8 + (size_t)d.max_states * 16 +
         (size_t)2 * rows_per_chunk * d.max_labels * 4 + 64 * 4 + (size_t)(d.max_levels + 1) * 4 +
         (size_t)(d.max_states + 1) * 4 + (size_t)2 * rows_per_chunk * 4 + 64);
}

// ------------------------------------------------------------------------------------------------
// stage 3: posteriors -> gradient rows
// ------------------------------------------------------------------------------------------------
// (banded acceptors: see band_grad_kernel)
struct BandArcs {
  int ok, has_self, has_adj, slot_self, slot_adj, wid_self, wid_adj;
  float w_self, w_adj;
};
// lane = state: its in-arcs if they fit the band (ok = 0 otherwise); -inf arcs do not exist (run_chain_prob)
__device__ __forceinline__ BandArcs band_in_arcs(const UttView& u, const float* __restrict__ weights, int lane) {
  BandArcs r{1, 0, 0, 0, 0, -1, -1, WFL_NEG_INF, WFL_NEG_INF};
  if (lane >= u.Q) return r;
  const int k0 = u.in_ptr[lane], k1 = u.in_ptr[lane + 1];
  for (int a = k0; a < k1; ++a) {
    const int wid = u.arc_wid[a];
    float w = u.arc_w[a];
    if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
    w = nan_to_neg(w);
    if (!(w > WFL_NEG_INF)) continue;
    const int src = u.arc_src[a];
    if (src == lane && !r.has_self)
      r.has_self = 1, r.slot_self = u.arc_slot[a], r.wid_self = wid, r.w_self = w;
    else if (src == lane - 1 && !r.has_adj)
      r.has_adj = 1, r.slot_adj = u.arc_slot[a], r.wid_adj = wid, r.w_adj = w;
    else
      r.ok = 0;
  }
  return r;
}
__device__ __forceinline__ bool band_shape(const wfl_lattice_desc& d, const UttView& u) {
  return u.Q >= 1 && u.Q <= 64 && u.E == 0 && d.max_labels <= 64;
}

// The dense gradient rows of a tile: base value (0, the existing gradient, or -cf * softmax(x) for the fused
// log-softmax backward) plus the accumulator of the column's label slot (`acc` [nr][Kmax], found through `colmap`);
// the rows never pass through LDS.  Shared by the general gradient kernel and the state-occupancy one.
// RU: rows of a trip (wide rows); NOACC: the caller never accumulates into an existing gradient (the workgroups beside the
// sweeps) -- no registers for it, so their trips take eight rows where the others take four: twice the bytes in flight
template <int RU_ = 4, bool NOACC = false>
__device__ __forceinline__ void stream_grad_rows(int b, int ts0, int nr, int T, int C, int Kmax, int tid, int NT,
                                                 float* __restrict__ dx, const float* __restrict__ x,
                                                 const float* __restrict__ row_lse, int accumulate, bool dead, float cf,
                                                 const float* acc, const int16_t* colmap, bool base_only = false) {
  // (base_only: the rows' base values alone -- the label columns' addends follow: occ_grad_tiles, "while it waits")
  // fused log_softmax backward (ctc.py:107, transducer.py:186-187): with g = cf * posteriors the
  // gradient w.r.t. the raw scores is g - softmax * sum_c g, and the posteriors of a frame sum to
  // one, so the base value of a row is -cf * softmax(x)
  float* gdst = dx + ((int64_t)b * T + ts0) * C;
  const float* xsrc = row_lse ? x + ((int64_t)b * T + ts0) * C : nullptr;
  const float* lse = row_lse ? row_lse + (int64_t)b * T + ts0 : nullptr;
  const bool soft = row_lse && !dead;
  auto value = [&](int r, int c, float have, float xv, float l) {
    float v = have;
    if (soft && l > WFL_NEG_INF) v -= cf * fast_exp(nan_to_neg(xv) - l);
    if (base_only) return v;
    const int k = colmap[c];
    if (k >= 0 && !dead) v += cf * acc[r * Kmax + k];
    return v;
  };
  if (C >= 512) {
    // wide rows: one float4 per thread and row (rows are only 4-byte aligned: scalar heads / tails, all rows' in one
    // pass at the end).  A trip is four rows with ALL their loads -- the rows' scores, their log-sum-exps, the gradient
    // they accumulate into -- issued together, and the NEXT trip's loads are issued before this one's values are
    // computed and stored: trip by trip, a tile of 32 rows was eight dependent round trips to HBM plus eight more for
    // the heads and tails (55 us of a workgroup's 77 per tile at the Transducer benchmark).
    const int64_t e0 = ((int64_t)b * T + ts0) * C;
    constexpr int RU = RU_;
    const int njb = ((C >> 2) + NT - 1) / NT, ntrips = ((nr + RU - 1) / RU) * njb;
    auto row_head = [&](int r) { return (int)((4 - ((e0 + (int64_t)r * C) & 3)) & 3); };
    struct Trip {
      float4 have[RU], xv[RU];
      float l[RU];
    };
    auto issue = [&](int trip, Trip& t) {  // (clamped everywhere: valid, aligned addresses; no branch around a load)
      const int r0 = (trip / njb) * RU, j = (trip % njb) * NT + tid;
#pragma unroll
      for (int q = 0; q < RU; ++q) {
        const int r = min(r0 + q, nr - 1);
        const int head = row_head(r), nvec = (C - head) >> 2;
        const int c = head + 4 * min(j, nvec - 1);
        t.have[q] = make_float4(0.f, 0.f, 0.f, 0.f), t.xv[q] = t.have[q];
        t.l[q] = soft ? lse[r] : 0.f;
        if (!NOACC && accumulate) t.have[q] = *reinterpret_cast<const float4*>(gdst + (int64_t)r * C + c);
        if (soft) t.xv[q] = *reinterpret_cast<const float4*>(xsrc + (int64_t)r * C + c);
      }
    };
    auto finish = [&](int trip, const Trip& t, bool live_trip) {
      const int r0 = (trip / njb) * RU, j = (trip % njb) * NT + tid;
#pragma unroll
      for (int q = 0; q < RU; ++q) {
        const int r = r0 + q;
        const int head = row_head(min(r, nr - 1)), nvec = (C - head) >> 2;
        if (live_trip && r < nr && j < nvec) {
          const int c = head + 4 * j;
          float4 o;
          o.x = value(r, c, t.have[q].x, t.xv[q].x, t.l[q]), o.y = value(r, c + 1, t.have[q].y, t.xv[q].y, t.l[q]);
          o.z = value(r, c + 2, t.have[q].z, t.xv[q].z, t.l[q]), o.w = value(r, c + 3, t.have[q].w, t.xv[q].w, t.l[q]);
          *reinterpret_cast<float4*>(gdst + (int64_t)r * C + c) = o;
        }
      }
    };
    Trip ta, tb;
    issue(0, ta);
    for (int trip = 0; trip < ntrips; trip += 2) {
      issue(min(trip + 1, ntrips - 1), tb);
      finish(trip, ta, true);
      issue(min(trip + 2, ntrips - 1), ta);
      finish(trip + 1, tb, trip + 1 < ntrips);
    }
    // heads and tails: up to 3 + 3 elements per row, eight slots per row, every row of the tile in one pass
    for (int e = tid; e < nr * 8; e += NT) {
      const int r = e >> 3, k = e & 7;
      const int head = row_head(r), nvec = (C - head) >> 2, ntail = C - head - 4 * nvec;  // ntail < 4
      if (k < head + ntail) {
        float* grow = gdst + (int64_t)r * C;
        const int c = k < head ? k : head + 4 * nvec + (k - head);
        grow[c] = value(r, c, accumulate ? grow[c] : 0.f, soft ? xsrc[(int64_t)r * C + c] : 0.f, soft ? lse[r] : 0.f);
      }
    }
  } else {
    // narrow rows: a lane owns a column (its label slot looked up once), a wave owns every fourth row
    const int lane = tid & 63, wv = tid >> 6, nw = NT >> 6;
    for (int c = lane; c < C; c += 64) {
      const int k = (dead || base_only) ? -1 : colmap[c];
#pragma unroll 4
      for (int r = wv; r < nr; r += nw) {
        const int i = r * C + c;
        float v = accumulate ? gdst[i] : 0.f;
        if (soft) {
          const float l = lse[r];
          if (l > WFL_NEG_INF) v -= cf * fast_exp(nan_to_neg(xsrc[i]) - l);
        }
        if (k >= 0) v += cf * acc[r * Kmax + k];
        gdst[i] = v;
      }
    }
  }
}

constexpr int kChunk = 16;
// Is utterance `u` one whose emission gradient is a sum of STATE occupancies?  Every arc INTO a state carries the same
// emission label ("uniform-label" acceptors: CTC-like chains, force alignment, the Transducer's alignment graphs -- the
// label belongs to the destination state), no epsilon arcs, swept in the probability domain.  Then
//     sum over the arcs a into q of  alpha_t[src a] w_a f_t[label q] beta_{t+1}[q]  =  alpha_{t+1}[q] beta_{t+1}[q]
// -- the left factor is what the forward sweep stored -- and the frame's gradient is one product per STATE (263 at the
// Transducer benchmark) instead of one per ARC (917), with no source / destination gathers and no alpha / beta tile in
// LDS.  (block-uniform; the general kernel applies the same test and skips what occ_grad_kernel served)
__device__ __forceinline__ bool occ_eligible(const UttView& u, bool prob) {
  int bad = !prob | (u.E > 0);
  if (!bad)
    for (int q = threadIdx.x; q < u.Q; q += blockDim.x) {
      const int i0 = u.in_ptr[q], i1 = u.in_ptr[q + 1];
      for (int k = i0 + 1; k < i1; ++k) bad |= u.arc_slot[k] != u.arc_slot[i0];
    }
  return !__syncthreads_or(bad);
}

// Emission gradient of uniform-label acceptors from state occupancies (see occ_eligible): tiles of up to 32 frames whose
// only LDS is the per-(frame, label) accumulator; a thread multiplies alpha and beta of one (frame, state) where they lie
// (rows of Q doubles: coalesced) and adds the product to its label's accumulator (ds_add_f32: a label's few states
// collide); the dense rows are streamed out as in the general kernel.  No learnable-weight gradient here (dW == NULL).
//
// LIVE: the workgroup runs INSIDE the launch of the sweeps (prob_chain_occ_kernel) and takes its tile as soon as both
// sweeps have passed it:
//   * it polls the two progress words of its utterance (relaxed agent-scope loads by one thread, s_sleep in between);
//   * alpha, beta and the per-slot offsets are read with L1-bypassing loads -- the sweeps' plain stores are in the L2
//     of THEIR XCD, which is this workgroup's XCD by construction of the grid (the XCC ids in the progress words are
//     compared with this workgroup's: a mismatch, or a poll that gives up, marks the utterance in `bad` and the general
//     gradient kernel of wfl_lattice_grad_rest writes its rows later, from the same alpha and beta);
//   * log2 Z is not known yet: sum_q alpha_s[q] beta_s[q] at the tile's first slot IS Z (every path passes through
//     exactly one state per slot; the identity the certificate checks at every 8th slot to 1e-4), so the tile
//     normalises by its own.
#ifdef WFL_LIVE_STATS
__device__ unsigned long long g_live[16];  // cycles (s_memtime) summed over workgroups: 0 busy-wait 1 job wait 2 setup 3 zloc 4 products 5 rows 6 jobs 7 polls
#define LIVE_T(v) const unsigned long long v = wall_clock64()
#define LIVE_ADD(k, a, b) if (threadIdx.x == 0) atomicAdd(&g_live[k], (b) - (a))
// a timeline of the last launch (100 MHz wall clock; scripts/live_timeline.py): per tile {utterance << 16 | first
// frame, wait began, wait over, occupancies accumulated, rows written}; per sweep {began, ended}
__device__ unsigned long long g_tl[4096][5];
__device__ unsigned int g_tl_n;
__device__ unsigned long long g_sweep[512][2];
__device__ unsigned long long g_wg[1024][4];  // gradient workgroup: began, its CU free, last job drawn, jobs done
#define LIVE_TILE(b, t, a0, a1, a2, a3) if (threadIdx.x == 0) { const unsigned int i_ = atomicAdd(&g_tl_n, 1u); if (i_ < 4096) { g_tl[i_][0] = ((unsigned long long)(b) << 16) | (unsigned)(t); g_tl[i_][1] = a0; g_tl[i_][2] = a1; g_tl[i_][3] = a2; g_tl[i_][4] = a3; } }
#else
#define LIVE_TILE(b, t, a0, a1, a2, a3)
#define LIVE_T(v)
#define LIVE_ADD(k, a, b)
#endif
struct OccLive {
  const uint64_t* prog_a;
  const uint64_t* prog_b;
  uint32_t* bad;
  uint32_t token;
  int R;          // frames per chunk of the sweeps
  int force_bad;  // (tests: behave as if the XCC ids differed)
  int mitm;       // the launch asked its sweeps to meet in the middle (they say in their progress words whether they did)
  const uint32_t* based;  // the tile's word in OccHeader::based (null: the tile writes whole rows itself)
};
__device__ __forceinline__ float ld_l2(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Have the sweeps of the launch reached frames ts0 .. ts0 + nr - 1 of the tile's utterance?  1 / 2: yes (full vectors /
// occupancies), 0: not this launch's (the general kernel's utterance), -1: cannot be served, -2: not within max_spins polls.
// One thread.
__device__ __forceinline__ int live_tile_state(const OccLive& live, int T, int ts0, int nr, int max_spins) {
  const uint32_t need_a = (uint32_t)((ts0 + nr + live.R - 1) / live.R);
  const uint32_t need_b = (uint32_t)((T - ts0 - 1 + live.R - 1) / live.R);
  // sweeps that meet in the middle (live.mitm): a sweep's choice is final once its word carries kProgMitm or counts
  // a chunk beyond its half (the crossing itself publishes the half's count, still without the flag).  Then a slot's
  // occupancy only needs the sweep that wrote it: slots > m the forward sweep, slots < m the backward one, slot m
  // the forward sweep's crossing (in L2 with its first chunk beyond).
  // (T not a multiple of 16: the backward sweep's occupancies end below m_b = T - 16 half_b <= m, slots m_b .. m
  // are the forward sweep's crossing too; a sweep's last, partial chunk comes with its final count)
  const int half_a = mitm_plain_chunks_a(T), half_b = mitm_plain_chunks_b(T), mid_b = T - 16 * half_b;
  const int s_lo = ts0 + 1, s_hi = ts0 + nr;
  const uint32_t gneed_a = s_hi >= mid_b ? (uint32_t)max((s_hi + 15) / 16, half_a + 1) : 0u;
  const uint32_t gneed_b = s_lo < mid_b ? (uint32_t)((T - 1 - s_lo) / 16 + 1) : 0u;
  const uint32_t me = xcc_id();
  int st = -2;
  for (int spin = 0; spin < max_spins; ++spin) {
    const uint64_t va = __hip_atomic_load(live.prog_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t vb = __hip_atomic_load(live.prog_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(va >> 32) == live.token && (uint32_t)(vb >> 32) == live.token) {
      const uint32_t fa = (uint32_t)va & 0x0fffffffu, fb = (uint32_t)vb & 0x0fffffffu;
      if (fa == kProgSkip || fb == kProgSkip) {
        st = 0;  // not swept in the probability domain: the general kernel's utterance
        break;
      }
      if ((((uint32_t)va >> 28) & 15u) != me || (((uint32_t)vb >> 28) & 15u) != me || live.force_bad) {
        st = -1;
        break;
      }
      const uint32_t ca = fa & kProgCount, cb = fb & kProgCount;
      if (!live.mitm) {
        if (ca >= need_a && cb >= need_b) {
          st = 1;
          break;
        }
      } else {
        const bool ma = (fa & kProgMitm) != 0, mb = (fb & kProgMitm) != 0;
        if ((ma || ca >= (uint32_t)half_a + 1) && (mb || cb >= (uint32_t)half_b + 1)) {  // both have chosen
          if (ma != mb) {
            st = -1;  // (one of them gave up waiting at its crossing: the certificate re-sweeps the utterance)
            break;
          }
          if (ma ? (ca >= gneed_a && cb >= gneed_b) : (ca >= need_a && cb >= need_b)) {
            st = ma ? 2 : 1;
            break;
          }
        }
      }
    }
    if (spin + 1 < max_spins) __builtin_amdgcn_s_sleep(64);
  }
  return st;
}

template <bool LIVE>
__device__ __forceinline__ void occ_grad_tiles(const wfl_lattice_desc& d, const UttView& u, int b, int T, int C,
                                               const float* __restrict__ alpha, const float* __restrict__ beta,
                                               const float* __restrict__ logz, const float* __restrict__ coef,
                                               const float* __restrict__ gout, int accumulate, const float* __restrict__ x,
                                               const float* __restrict__ row_lse, float* __restrict__ dx, int t_begin,
                                               int t_end, int TS, int64_t tail, int nch1, char* smem, const OccLive& live) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int Q = u.Q, K = u.K, Kmax = d.max_labels;
  double* red = (double*)smem;                                // [16] + the workgroup's verdict on its tile (LIVE)
  int* state = (int*)(red + 16);
  float* acc = (float*)(red + 18);                            // [TS][Kmax]
  double* corr = (double*)(acc + (((size_t)TS * Kmax + 1) & ~(size_t)1));  // [TS]
  int16_t* lab = (int16_t*)(corr + TS);                       // [Qmax]: label slot of the state (-1: no in-arc)
  int16_t* colmap = lab + ((d.max_states + 3) & ~3);          // [C]
  int32_t* colk = reinterpret_cast<int32_t*>(colmap + ((C + 3) & ~3));  // [Kmax]: the column of a label slot
  const double* offs_a = reinterpret_cast<const double*>(alpha + tail) + (int64_t)b * nch1;
  const double* offs_b = reinterpret_cast<const double*>(beta + tail) + (int64_t)b * nch1;
  double zd = 0.0;
  bool dead = false;
  // gamma: the sweeps met in the middle and stored occupancies (floats) -- slots <= m in the beta buffer's rows, slots > m in
  // the alpha buffer's (run_chain_prob); nothing to normalise.  LIVE: decided per tile from the progress words.
  bool gamma = false;
  const int mid = mitm_middle(T);
  if (!LIVE) {
    zd = reinterpret_cast<const double*>(alpha + tail)[(int64_t)d.B * nch1 + b];  // log2 Z
    const int32_t* fmt = reinterpret_cast<const int32_t*>(reinterpret_cast<const double*>(alpha + tail) + (int64_t)d.B * (nch1 + 1));
    gamma = fmt[b] == kFmtOcc;
    if (!occ_eligible(u, fmt[b] == kFmtProb || gamma)) return;
    const float z = logz[b];
    dead = !(z > WFL_NEG_INF) || !(z < __builtin_inff());  // no accepting path: zero gradient
  } else if (!occ_eligible(u, true)) {
    return;
  }
  LIVE_T(t_s0);
  const double* alpha_d = reinterpret_cast<const double*>(alpha) + u.ab_base;
  const double* beta_d = reinterpret_cast<const double*>(beta) + u.ab_base;
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = coef ? coef[b] * g0 : g0;
  for (int c = tid; c < C; c += NT) colmap[c] = -1;
  for (int q = tid; q < Q; q += NT) lab[q] = u.in_ptr[q] < u.in_ptr[q + 1] ? (int16_t)u.arc_slot[u.in_ptr[q]] : (int16_t)-1;
  __syncthreads();
  for (int k = tid; k < K; k += NT) colmap[u.labels[k]] = (int16_t)k, colk[k] = u.labels[k];
  const float inv_q = 1.f / (float)max(Q, 1);
  for (int ts0 = t_begin; ts0 < t_end; ts0 += TS) {
    const int nr = min(TS, t_end - ts0);
    bool prewritten = false;
    __syncthreads();
    LIVE_T(t_w0);
    if (LIVE) LIVE_ADD(2, t_s0, t_w0);
    if (LIVE) {
      // frames ts0 .. ts0 + nr - 1 need slots ts0 + 1 .. ts0 + nr of both sweeps: the forward sweep has stored slots
      // <= chunks * R, the backward sweep slots >= T - chunks * R
      // st: 1 / 2 ready (full vectors / occupancies), 0 not this launch's, -1 cannot be served, -2 not yet
      auto poll = [&](int max_spins) { return live_tile_state(live, T, ts0, nr, max_spins); };
      // The rows' base values, -cf softmax(x), are written by the first phase of occ_live_kernel -- any workgroup of this
      // XCD, before the sweeps have reached the tile; here only the label columns' occupancies are added to them.  So the
      // tile also waits for its word in `based` (it is drawn only after every base job has been: no deadlock).
      if (tid == 0) {
        int st = poll(1 << 20);  // (gives up after ~2 s)
        if (st >= 1 && live.based) {
          bool seen = false;
          for (int spin = 0; spin < (1 << 20) && !seen; ++spin) {
            seen = __hip_atomic_load(live.based, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == live.token;
            if (!seen) __builtin_amdgcn_s_sleep(32);
          }
          if (!seen) st = -2;
        }
        if (st < 0) __hip_atomic_store(live.bad, live.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *state = st;
      }
      __syncthreads();
      if (*state < 1) return;
      prewritten = live.based != nullptr;
      gamma = *state == 2;
      LIVE_T(t_w1);
      LIVE_ADD(1, t_w0, t_w1);
      if (!gamma) {
        // the tile's own log2 Z, from its first slot
        double part = 0.0;
        for (int q = tid; q < Q; q += NT)
          part = fma(ld_l2(alpha_d + (int64_t)(ts0 + 1) * Q + q), ld_l2(beta_d + (int64_t)(ts0 + 1) * Q + q), part);
        const double tot = block_reduce_sum_f64(part, red);
        dead = !(tot > 0.0 && tot < 1.0e300);
        zd = dead ? 0.0 : log2(tot) + ld_l2(offs_a + ts0 + 1) + ld_l2(offs_b + ts0 + 1);
      } else {
        dead = false;  // (an utterance without a path: its sweeps stored zeros)
      }
      LIVE_T(t_w2);
      LIVE_ADD(3, t_w1, t_w2);
    }
    LIVE_T(t_p0);
    for (int i = tid; i < nr * Kmax; i += NT) acc[i] = 0.f;
    // frame t's arcs end in slot t + 1 of both sweeps: gamma = p_alpha p_beta 2^(offs_a + offs_b - log2 Z) there
    if (tid < nr && !gamma)
      corr[tid] = LIVE ? exp2(ld_l2(offs_a + ts0 + tid + 1) + ld_l2(offs_b + ts0 + tid + 1) - zd)
                       : exp2(offs_a[ts0 + tid + 1] + offs_b[ts0 + tid + 1] - zd);
    __syncthreads();
    if (gamma && !dead) {
      // occupancies as the sweeps stored them: Q floats at the head of the slot's row (of Q doubles) -- the beta buffer's up
      // to the middle slot, the alpha buffer's beyond
      const int n = nr * Q;
      constexpr int U = 8;  // (16: no different -- 0.3136 against 0.3107 ms a step)
      for (int i0 = tid; i0 < n; i0 += U * NT) {
        float gv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int i = min(i0 + k * NT, n - 1);
          const int r = (int)(((float)i + 0.5f) * inv_q), q = i - r * Q, sl = ts0 + 1 + r;
          const float* row = reinterpret_cast<const float*>((sl <= mid ? beta_d : alpha_d) + (int64_t)sl * Q);
          gv[k] = LIVE ? ld_l2(row + q) : row[q];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int i = i0 + k * NT;
          const int r = (int)(((float)i + 0.5f) * inv_q), q = i - r * Q;
          if (i < n) {
            const int kk = lab[q];
            if (kk >= 0 && gv[k] != 0.f) atomicAdd(&acc[r * Kmax + kk], gv[k]);
          }
        }
      }
    } else if (!dead) {
      const double* asrc = alpha_d + (int64_t)(ts0 + 1) * Q;
      const double* bsrc = beta_d + (int64_t)(ts0 + 1) * Q;
      const int n = nr * Q;
      if (LIVE) {
        // (the L1-bypassing loads are relaxed atomics, which the compiler does not move across the LDS atomics below:
        // left to it, every pair of loads is waited for before the next is issued -- batches of 16 by hand)
        constexpr int U = 8;  // (4: one more workgroup per SIMD, 0.414 against 0.409 ms at cfg4)
        for (int i0 = tid; i0 < n; i0 += U * NT) {
          double av[U], bv[U];
#pragma unroll
          for (int k = 0; k < U; ++k) {
            const int i = min(i0 + k * NT, n - 1);
            av[k] = ld_l2(asrc + i), bv[k] = ld_l2(bsrc + i);
          }
#pragma unroll
          for (int k = 0; k < U; ++k) {
            const int i = i0 + k * NT;
            const int r = (int)(((float)i + 0.5f) * inv_q), q = i - r * Q;  // (exact for i < 2^20)
            const double g = av[k] * bv[k];
            if (i < n) {
              const int kk = lab[q];
              if (kk >= 0 && g != 0.0) atomicAdd(&acc[r * Kmax + kk], (float)(g * corr[r]));
            }
          }
        }
      } else {
#pragma unroll 4
        for (int i = tid; i < n; i += NT) {
          const int r = (int)(((float)i + 0.5f) * inv_q), q = i - r * Q;  // (exact for i < 2^20)
          const double g = asrc[i] * bsrc[i];
          const int k = lab[q];
          if (k >= 0 && g != 0.0) atomicAdd(&acc[r * Kmax + k], (float)(g * corr[r]));
        }
      }
    }
    __syncthreads();
    LIVE_T(t_p1);
    // (eight rows a trip for the workgroups beside the sweeps -- stream_grad_rows<8, true>, twice the bytes in flight for the
    // same registers -- changes nothing: a tile's rows 23.9 us either way, the step 0.342 against 0.338 ms)
    if (LIVE && prewritten && !dead) {
      // the base values are this workgroup's own, in L2: + cf x the occupancy of the row's label columns (the same two
      // roundings as the single pass: -cf softmax, then + cf acc)
      // (sixteen elements a thread at a time, all their loads in flight together: a load behind a store to the same
      // array is a load the compiler must wait for -- one round trip to L2 per element, 15 us for a tile's 6 000)
      float* gdst = dx + ((int64_t)b * T + ts0) * C;
      const float inv_k = 1.f / (float)max(K, 1);
      const int n = nr * K;
      constexpr int UP = 16;
      for (int i0 = tid; i0 < n; i0 += UP * NT) {
        float* pv[UP];
        float have[UP], add[UP];
#pragma unroll
        for (int j = 0; j < UP; ++j) {
          const int i = min(i0 + j * NT, n - 1);
          const int r = (int)(((float)i + 0.5f) * inv_k), k = i - r * K;
          pv[j] = gdst + (int64_t)r * C + colk[k];
          add[j] = i0 + j * NT < n ? cf * acc[r * Kmax + k] : 0.f;
          have[j] = ld_l2(pv[j]);
        }
#pragma unroll
        for (int j = 0; j < UP; ++j)
          if (add[j] != 0.f) *pv[j] = have[j] + add[j];
      }
    } else {
      stream_grad_rows(b, ts0, nr, T, C, Kmax, tid, NT, dx, x, row_lse, accumulate, dead, cf, acc, colmap);
    }
    LIVE_T(t_p2);
    if (LIVE) {
      LIVE_ADD(4, t_p0, t_p1);
      LIVE_ADD(5, t_p1, t_p2);
      LIVE_ADD(6, 0ull, 1ull);
      LIVE_TILE(b, ts0, t_w0, t_p0, t_p1, t_p2);
    }
  }
}
// LDS of occ_grad_tiles (bytes)
static size_t occ_lds_bytes(const wfl_lattice_desc& d, int TS, int C) {
  return 18 * 8 + (((size_t)TS * d.max_labels + 1) & ~(size_t)1) * 4 + 8 * (size_t)TS +
         2 * (((size_t)d.max_states + 3) & ~(size_t)3) + 2 * (((size_t)C + 3) & ~(size_t)3) + 4 * (size_t)d.max_labels + 16;
}

__global__ void __launch_bounds__(256)
    occ_grad_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats, int T, int C,
                    const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ logz,
                    const float* __restrict__ coef, const float* __restrict__ gout, int accumulate,
                    const float* __restrict__ x, const float* __restrict__ row_lse, float* __restrict__ dx,
                    int rows_per_block, int TS, int64_t tail, int nch1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y;
  const UttView u = make_view(d, ints, floats, b, T);
  const int t_begin = blockIdx.x * rows_per_block;
  occ_grad_tiles<false>(d, u, b, T, C, alpha, beta, logz, coef, gout, accumulate, x, row_lse, dx, t_begin,
                        min(T, t_begin + rows_per_block), TS, tail, nch1, smem, OccLive{});
}

// The occupancy gradient BESIDE the sweeps.  The sweeps of a batch of B utterances are 2 B workgroups that each run for
// the whole launch at the pace of one dependent frame after the other, on a chip with 256 CUs; the gradient is a
// bandwidth-bound pass over alpha, beta and the emissions that can start in the middle of the utterance as soon as the
// two sweeps have crossed and then follows them outwards.  Three launches on two streams (wfl_lattice_forward_grad):
//     stream      prob_chain_pub_kernel   the sweeps, publishing a progress word per (utterance, direction);
//                                         ids [0, 2 Bp): direction id / Bp, utterance id % Bp (Bp = B rounded up to 8),
//                                         so that the two sweeps of an utterance land on the same XCD (ids are dealt to
//                                         the XCDs round-robin)
//     side        occ_gate_kernel         one wave: waits until every sweep workgroup has announced itself (from then on
//                                         nothing the gradient does can keep a sweep off the chip), then sorts the
//                                         utterances by the XCD their sweeps run on and resets the job counters
//     side        occ_live_kernel         persistent 256-thread workgroups (their own register budget: three per CU resident; a
//                                         kernel that also held the sweeps' code would get 1 workgroup per CU): each
//                                         reads the id of the XCD it runs on and draws (tile, utterance) jobs of THAT
//                                         XCD, tiles in middle-out order -- the sweeps' plain stores are in that XCD's
//                                         L2 (see occ_grad_tiles)
// The kernels themselves do not rely on how workgroups are dealt to XCDs: an utterance whose sweeps ended up on
// different XCDs is marked in `bad`, one whose XCD hosts no gradient workgroup shows in the job counters: the general
// gradient kernel of wfl_lattice_grad_rest serves both (served_beside_the_sweeps).
template <int MAXT>
__global__ void __launch_bounds__(MAXT)
    prob_chain_pub_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats,
                          const float* __restrict__ xg, int T, int rows_per_chunk, const float* __restrict__ weights,
                          float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ logz, int64_t tail,
                          int nch1, int Bp, uint32_t token, int mitm_req) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x % Bp;
  if (b >= d.B) return;
  __builtin_amdgcn_s_setprio(2);
  // the gradient workgroups leave a CU alone while a sweep runs on it (occ_live_kernel)
  uint32_t* busy = occ_header(d, alpha, tail, nch1).busy + cu_key();
  if (threadIdx.x == 0) __hip_atomic_store(busy, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef WFL_LIVE_STATS
  if (threadIdx.x == 0 && blockIdx.x < 512) g_sweep[blockIdx.x][0] = wall_clock64();
  WFL_LOG(2);
#endif
  prob_chain_body<MAXT, true>(d, ints, floats, xg, T, rows_per_chunk, weights, alpha, beta, logz, tail, nch1, b,
                              blockIdx.x / Bp, smem, token, mitm_req);
#ifdef WFL_LIVE_STATS
  if (threadIdx.x == 0 && blockIdx.x < 512) g_sweep[blockIdx.x][1] = wall_clock64();
  WFL_LOG(3);
#endif
  if (threadIdx.x == 0) __hip_atomic_store(busy, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `verdict` (one word per call, outside the sweeps' buffers): the launch's token once the header below is complete, its
// complement if the gate gave up -- what occ_live_kernel looks at before it touches the header.  `host_words` (pinned):
// [0] gates that gave up, [1] gates that went through, counted for the host's back-off (wfl_lattice_diagnostics).
__global__ void __launch_bounds__(64)
    occ_gate_kernel(wfl_lattice_desc d, float* __restrict__ alpha, float* __restrict__ beta, int64_t tail, int nch1,
                    uint32_t token, int force_bad, uint32_t* __restrict__ host_words, uint32_t* __restrict__ verdict,
                    int ntiles, int max_spins) {
  const int lane = threadIdx.x;
  const uint64_t* pa = reinterpret_cast<const uint64_t*>(reinterpret_cast<double*>(alpha + tail) + prog_offset_doubles(d, nch1));
  const uint64_t* pb = reinterpret_cast<const uint64_t*>(reinterpret_cast<double*>(beta + tail) + prog_offset_doubles(d, nch1));
  const OccHeader h = occ_header(d, alpha, tail, nch1);
  // First every sweep of the launch must have announced itself -- until then NOTHING is written to alpha / beta: the
  // buffers may be a previous call's, whose gradient kernel (wfl_lattice_grad_rest) may read its header until the
  // stream reaches this call's sweeps.  The wait is bounded (max_spins polls of ~2 us: the side stream was forked
  // from the caller's stream right in front of the sweeps, so they are microseconds away unless something runs the
  // streams' kernels one after the other -- a counter-collecting profiler, AMD_SERIALIZE_KERNEL, a debugger).  On a
  // give-up the gate writes its verdict and the host counter and ends: the gradient workgroups behind it see the
  // verdict and end too, the header keeps whatever token it had, and wfl_lattice_grad_rest -- which believes
  // "served beside the sweeps" only under THIS launch's token -- computes every row.
  constexpr int kMaxRounds = 4;  // (B <= 256: the launch is only taken while every sweep gets a CU of its own)
  int xcds[kMaxRounds];
  bool gave_up = false;
#pragma unroll
  for (int r = 0; r < kMaxRounds; ++r) {
    const int b = r * 64 + lane;
    int xcd = -1;  // -1: not for the in-flight gradient
    if (b < d.B) {
      bool seen = false;
      for (int spin = 0; spin < max_spins && !seen; ++spin) {
        const uint64_t va = __hip_atomic_load(pa + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t vb = __hip_atomic_load(pb + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(va >> 32) == token && (uint32_t)(vb >> 32) == token) {
          seen = true;
          const uint32_t xa = ((uint32_t)va >> 28) & 15u, xb = ((uint32_t)vb >> 28) & 15u;
          const bool skip = ((uint32_t)va & 0x0fffffffu) == kProgSkip || ((uint32_t)vb & 0x0fffffffu) == kProgSkip;
          if (!skip) xcd = (xa == xb && xa < 8 && !force_bad) ? (int)xa : -2;
        } else {
          __builtin_amdgcn_s_sleep(64);
        }
      }
      gave_up |= !seen;
    }
    xcds[r] = xcd;
    if (__any(gave_up)) break;  // (the rest would only wait as long again)
  }
  if (__any(gave_up)) {
    if (lane == 0) {
      __hip_atomic_store(verdict, ~token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(host_words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  int count[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (wave-uniform)
#pragma unroll
  for (int r = 0; r < kMaxRounds; ++r) {
    const int b = r * 64 + lane, xcd = xcds[r];
    if (b < d.B && xcd == -2) h.bad[b] = token;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const uint64_t m = __ballot(xcd == x);
      if (xcd == x) h.list[x * d.B + count[x] + __popcll(m & ((1ull << lane) - 1))] = b;
      count[x] += __popcll(m);
    }
  }
  if (lane < 8) h.next[lane] = 0, h.next_base[lane] = 0;
  if (lane == 0) h.meta[0] = token, h.meta[1] = (uint32_t)ntiles;
#pragma unroll
  for (int x = 0; x < 8; ++x)
    if (lane == 0) h.nx[x] = count[x];
  if (lane == 0) {
    __hip_atomic_store(verdict, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(host_words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void __launch_bounds__(256, 3)  // (three waves a SIMD, i.e. three workgroups a CU: at most 168 VGPRs -- at 169 a third fewer)
    occ_live_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats, int T, int C,
                    float* __restrict__ alpha, float* __restrict__ beta, const float* __restrict__ coef,
                    const float* __restrict__ x, const float* __restrict__ row_lse, float* __restrict__ dx, int rows_o,
                    int ntiles, int rows_per_chunk, int64_t tail, int nch1, uint32_t token,
                    const uint32_t* __restrict__ verdict, int mitm_req, int two_phase) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ uint32_t job_s;
  // (the gate in front of this kernel on the side stream: anything but the launch's token means it gave up and the
  // header is not this launch's -- nothing to do here, wfl_lattice_grad_rest computes every row)
  if (__hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token) return;
  const OccHeader h = occ_header(d, alpha, tail, nch1);
  const uint32_t me = xcc_id() & 7u;
  const int nx = h.nx[me];
  const double* ta = reinterpret_cast<const double*>(alpha + tail) + prog_offset_doubles(d, nch1);
  const double* tb = reinterpret_cast<const double*>(beta + tail) + prog_offset_doubles(d, nch1);
  // middle-out: mid, mid - 1, mid + 1, mid - 2, ... and the rest of the longer side
  const int mid = ntiles / 2, nlo = mid, nhi = ntiles - mid, pair = min(nlo, nhi);
#ifndef WFL_LIVE_SHARE_CU
  // A sweep is a chain of dependent frames that any neighbour on its CU slows down, and the launch ends when the
  // last sweep does: workgroups that find one on their CU sit out until it is done; the others -- half the chip at
  // the Transducer benchmark's 128 sweeps -- draw the jobs meanwhile.
  LIVE_T(t_b0);
  if (threadIdx.x == 0) {
    const uint32_t* busy = h.busy + cu_key();
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (__hip_atomic_load(busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token) break;
      __builtin_amdgcn_s_sleep(127);
    }
  }
  LIVE_T(t_b1);
  LIVE_ADD(0, t_b0, t_b1);
#endif
#ifdef WFL_LIVE_STATS
  unsigned long long njobs_done = 0;
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg[blockIdx.x][0] = t_b0, g_wg[blockIdx.x][1] = t_b1;
#endif
  if (nx <= 0) return;
  auto tile_of = [&](uint32_t job, int& b) {  // job -> (tile, utterance), tiles in middle-out order
    const int r = (int)(job / (uint32_t)nx);
    b = h.list[me * d.B + (int)(job % (uint32_t)nx)];
    if (r < 2 * pair) return (r & 1) ? mid - 1 - (r >> 1) : mid + (r >> 1);
    return nhi > nlo ? mid + pair + (r - 2 * pair) : mid - 1 - pair - (r - 2 * pair);
  };
  // ---- first phase: the rows' base values.  A row of the gradient is -cf softmax(x) plus the occupancies of a few label
  // columns (16 of 1001 at the Transducer benchmark), and only the latter wait for the sweeps: the workgroups stream the
  // former -- all of the step's gradient traffic but a line per (frame, label) -- while the sweeps are in their first
  // half and there is nothing else for them to do (the first occupancies appear at the crossing, 85 us into the launch),
  // and mark each tile in `based`.  With whole rows written per tile behind the sweeps the gradient ran at the capacity
  // of its workgroups from the crossing on and was 80 us behind when the sweeps ended (B = 64: 858 of 1 600 tiles,
  // profiles/r06_live_timeline.txt).
  const uint32_t total = (uint32_t)nx * (uint32_t)ntiles;
  auto live_of = [&](int b, int tile) {
    OccLive live;
    live.prog_a = reinterpret_cast<const uint64_t*>(ta) + b;
    live.prog_b = reinterpret_cast<const uint64_t*>(tb) + b;
    live.bad = h.bad + b;
    live.token = token, live.R = rows_per_chunk, live.force_bad = 0, live.mitm = mitm_req;
    live.based = two_phase ? h.based + (int64_t)b * ntiles + tile : nullptr;
    return live;
  };
  // ---- first phase: the rows' base values.  A row of the gradient is -cf softmax(x) plus the occupancies of its label
  // columns, and only the latter wait for the sweeps: the workgroups stream the former -- the step's gradient traffic --
  // while the sweeps are in their first half and there is nothing else for them to do (the first occupancies appear at
  // the crossing, 85 us into the launch), and mark each tile in `based`.  With whole rows written per tile behind the
  // sweeps the gradient ran at the capacity of its workgroups from the crossing on and was 80 us behind when the sweeps
  // ended (B = 64: 858 of 1 600 tiles; profiles/r06_live_timeline.txt).
  // (Every base job is drawn before any occupancy job: a tile that waits for its base rows waits for a workgroup that
  // is writing them.  Occupancy jobs first where they are ready -- a look at the head job, then a ticket or a
  // compare-and-swap -- was built twice: the looks cost more than the order gains, 0.40 ms and 2.1 ms a step.)
  // (two_phase = 0, WFL_LATTICE_TWO_PHASE=0: whole rows per tile behind the sweeps as until round 6 -- 13 us a step slower at
  // the Transducer benchmark, 200 MB less traffic: the label columns' lines are not read and written a second time)
  for (; two_phase;) {
    __syncthreads();
    if (threadIdx.x == 0) job_s = __hip_atomic_fetch_add(h.next_base + me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t job = job_s;
    if (job >= total) break;
    int b;
    const int tile = tile_of(job, b);
    const int t_begin = tile * rows_o, nr = min(T, t_begin + rows_o) - t_begin;
    stream_grad_rows(b, t_begin, nr, T, C, d.max_labels, threadIdx.x, blockDim.x, dx, x, row_lse, 0, false, coef ? coef[b] : 1.f,
                     nullptr, nullptr, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this thread's rows are in L2 ...)
    __syncthreads();                                   // (... and every thread's)
    if (threadIdx.x == 0)
      __hip_atomic_store(h.based + (int64_t)b * ntiles + tile, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- second phase: the occupancies of the tiles' label columns, added to their base rows, tile by tile behind the sweeps
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) job_s = __hip_atomic_fetch_add(h.next + me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t job = job_s;
#ifdef WFL_LIVE_STATS
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg[blockIdx.x][2] = wall_clock64(), g_wg[blockIdx.x][3] = njobs_done++;
#endif
    if (job >= total) return;
    int b;
    const int tile = tile_of(job, b);
    const UttView u = make_view(d, ints, floats, b, T);
    const OccLive live = live_of(b, tile);
    const int t_begin = tile * rows_o;
    occ_grad_tiles<true>(d, u, b, T, C, alpha, beta, nullptr, coef, nullptr, 0, x, row_lse, dx, t_begin,
                         min(T, t_begin + rows_o), rows_o, tail, nch1, smem, live);
  }
}

__global__ void __launch_bounds__(256)
    grad_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats,
                const float* __restrict__ xg, int T, int C, const float* __restrict__ weights,
                const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ logz,
                const float* __restrict__ coef, const float* __restrict__ coef_w, const float* __restrict__ gout,
                int accumulate, const float* __restrict__ x, const float* __restrict__ row_lse,
                float* __restrict__ dx, float* __restrict__ dW, int rows_per_block, int TS, int64_t tail, int nch1,
                int R, int skip_band, int skip_occ) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y, tid = threadIdx.x, NT = blockDim.x;
  WFL_MARK_BEGIN(2);
  WFL_LOG(12);
  const UttView u = make_view(d, ints, floats, b, T);
  const int Q = u.Q, A = u.A, E = u.E, K = u.K, Kmax = d.max_labels;
  // The dense rows never pass through LDS: posteriors are accumulated per (frame, distinct label)
  // in a compact tile, and the rows are streamed out as base value (0, the existing gradient, or
  // the softmax term of the fused log_softmax backward) plus the accumulator of the column's label
  // slot, looked up in a column -> slot map.
  // alpha / beta rows of the tile: doubles -- probabilities for utterances swept in the probability domain
  // (fmt[b] == kFmtProb), log scores for the general sweep's (run_chain: ChainVal)
  double* ald = (double*)smem;                              // [TS+1][Qmax]
  double* bed = ald + (size_t)(TS + 1) * d.max_states;      // [TS+1][Qmax]
  float* xr = (float*)(bed + (size_t)(TS + 1) * d.max_states);  // [TS][Kmax]: log scores | factors
  float* acc = xr + (size_t)TS * Kmax;                      // [TS][Kmax] (only if dx)
  float* dwacc = acc + (dx ? (size_t)TS * Kmax : 0);        // [A + E] (only if dW)
  const size_t nae = (size_t)d.max_arcs + d.max_eps;
  int* lead = (int*)(dwacc + (dW ? nae : 0));               // [A + E, rounded up to even] (only if dW): first arc with the same weight id
  int2* sarc = (int2*)(lead + (dW ? ((nae + 1) & ~(size_t)1) : 0));  // [A] by-slot {src | dst << 16, w - z}
  int* sptr = (int*)(sarc + (dx ? d.max_arcs : 0));         // [K + 1]
  // work items of the emission gradient: a slot's arc list in chunks of at most kChunk arcs, so that
  // the blank column of a CTC-like acceptor (two in-arcs per blank state: hundreds of arcs in ONE
  // slot) is spread over many threads instead of serialising the tile
  int2* chunk = (int2*)(sptr + (dx ? ((Kmax + 3) & ~1) : 0));  // [NC] {slot, first arc}; NC <= K + A / kChunk (8-byte aligned)
  double* corr_d = (double*)(chunk + (dx ? Kmax + d.max_arcs / kChunk + 1 : 0));  // [TS] probability domain: 2^(offsets - log2 Z);
                                                                                  // log domain: offs_alpha(t) + offs_beta(t+1) - log Z
  double* corr_eps = corr_d + 33;       // [TS+1] log domain: both offsets at slot t (epsilon arcs)
  int16_t* colmap = (int16_t*)(corr_eps + 34);  // [C] (only if dx)
  // scores are stored relative to per-chunk double offsets (run_chain): slot s of alpha belongs to chunk (s-1)/R of
  // the forward sweep, slot s of beta to chunk (T-1-s)/R of the backward sweep, the boundary slots to offset 0
  const double* offs_a = reinterpret_cast<const double*>(alpha + tail) + (int64_t)b * nch1;
  const double* offs_b = reinterpret_cast<const double*>(beta + tail) + (int64_t)b * nch1;
  const double zd = reinterpret_cast<const double*>(alpha + tail)[(int64_t)d.B * nch1 + b];  // ln Z | log2 Z (prob)
  const int32_t* fmt = reinterpret_cast<const int32_t*>(reinterpret_cast<const double*>(alpha + tail) + (int64_t)d.B * (nch1 + 1));
  const float* wrefs = reinterpret_cast<const float*>(fmt + d.B);
  const bool prob = fmt[b] == kFmtProb;
  if (fmt[b] == kFmtOcc) {
    // sweeps that met in the middle left occupancies, not both vectors (run_chain_prob): only the occupancy gradient reads
    // them -- the workgroups beside the sweeps (skip_occ == 2, unless they say otherwise) or occ_grad_kernel in this call
    // (skip_occ == 1); what neither served is computed right here, the same tiles
    if (skip_occ == 1 || (skip_occ == 2 && served_beside_the_sweeps(d, alpha, tail, nch1, b)) || !dx) return;
    const int tb = blockIdx.x * rows_per_block;
    occ_grad_tiles<false>(d, u, b, T, C, alpha, beta, logz, coef, gout, accumulate, x, row_lse, dx, tb, min(T, tb + rows_per_block),
                          TS, tail, nch1, smem, OccLive{});
    return;
  }
  if (skip_band) {  // band_grad_kernel served this utterance (the same test decides there)
    const int band_ok = band_in_arcs(u, weights, tid).ok;
    if (__syncthreads_and(band_ok) && prob && band_shape(d, u)) return;
  }
  // occ_grad_kernel served this utterance (the same test decides there); skip_occ == 2: the gradient workgroups beside
  // the sweeps did, unless they say otherwise
  if (skip_occ && occ_eligible(u, prob) && (skip_occ != 2 || served_beside_the_sweeps(d, alpha, tail, nch1, b))) return;
  const float wref = prob ? wrefs[b] : 0.f;
  const float* fgp = xg + xg_main_dev(d, T);                    // probability-domain factors of the gathered rows
  const float* rmaxp = fgp + xg_main_dev(d, T) + (int64_t)b * T;  // their references
  const double* alpha_d = reinterpret_cast<const double*>(alpha);
  const double* beta_d = reinterpret_cast<const double*>(beta);
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = coef ? coef[b] * g0 : g0;
  const float z = logz[b];
  const bool dead = !(z > WFL_NEG_INF) || !(z < __builtin_inff());  // no accepting path: zero gradient
  const int t_begin = blockIdx.x * rows_per_block;
  const int t_end = min(T, t_begin + rows_per_block);
  const float inv_q = 1.f / (float)max(Q, 1);
  auto fdiv = [](int i, float inv) { return (int)(((float)i + 0.5f) * inv); };
  if (dW) {
    // Arcs that share a learnable weight (every blank self-loop of a CTC-like alignment under a bigram model: 45 per
    // utterance at the n-gram benchmark) are summed HERE before anything goes to the global gradient: with one global
    // atomic per arc, 21 600 of a launch's 106 000 landed on ONE address and the launch took 173 us for 9 us of work
    // (same-address atomics are served one after the other: ~8 ns each).  lead[a]: the first arc with a's weight id.
    int* widc = reinterpret_cast<int*>(dwacc);  // (the ids, for the search; zeroed below)
    for (int a = tid; a < A + E; a += NT) widc[a] = a < A ? u.arc_wid[a] : u.eps_wid[a - A];
    __syncthreads();
    const bool search = A + E <= 1024;  // (quadratic in the arc count: beyond, every arc is its own leader)
    for (int a = tid; a < A + E; a += NT) {
      int l = a;
      if (search) {
        const int wid = widc[a];
        for (int c = 0; c < a; ++c)
          if (widc[c] == wid) {
            l = c;
            break;
          }
      }
      lead[a] = l;
    }
    __syncthreads();
    for (int a = tid; a < A + E; a += NT) dwacc[a] = 0.f;
  }
  if (dx) {
    for (int c = tid; c < C; c += NT) colmap[c] = -1;
    __syncthreads();
    for (int k = tid; k < K; k += NT) colmap[u.labels[k]] = (int16_t)k;
    for (int k = tid; k <= K; k += NT) sptr[k] = u.slot_ptr[k];
    __syncthreads();
    if (tid == 0) {  // chunk table (a few hundred entries at most, once per workgroup) -- from the LDS copy of the slot
                     // pointers: read from global memory here, every slot was two dependent round trips of ONE thread
                     // (K = 83 labels: ~150 of the kernel's 208 us at the n-gram benchmark's shape)
      int nc = 0;
      for (int k = 0; k < K; ++k) {
        const int a1 = sptr[k + 1];
        for (int a0 = sptr[k]; a0 < a1; a0 += kChunk) chunk[nc++] = make_int2(k, a0);
      }
      sptr[Kmax + 1] = nc;
    }
    for (int j = tid; j < A; j += NT) {
      const int a = u.slot_arc[j];
      const int wid = u.arc_wid[a];
      float w = u.arc_w[a];
      if (weights && wid >= 0) w += nan_to_neg(weights[wid]);
      if (prob) w = fast_exp(nan_to_neg(w) - wref);  // the arc's factor (run_chain_prob)
      sarc[j] = make_int2(u.arc_src[a] | (u.arc_dst[a] << 16), __float_as_int(w));
    }
  }
  for (int ts0 = t_begin; ts0 < t_end; ts0 += TS) {
    const int nr = min(TS, t_end - ts0);
    __syncthreads();
    // flat, unrolled copy loops: the global loads of several iterations are in flight together
    // (idx / n by float reciprocal: exact for idx < 2^20, see fdiv)
    {
      const int n = (nr + 1) * Q;
      {  // (doubles in both formats)
        const double* asrc = alpha_d + u.ab_base + (int64_t)ts0 * Q;
        const double* bsrc = beta_d + u.ab_base + (int64_t)ts0 * Q;
#pragma unroll 4
        for (int i = tid; i < n; i += NT) {
          const int r = fdiv(i, inv_q), q = i - r * Q;
          const double av = asrc[i], bv = bsrc[i];
          ald[r * d.max_states + q] = av;
          bed[r * d.max_states + q] = bv;
        }
      }
      const float* xsrc = (prob ? fgp : xg) + u.xg_base + (int64_t)ts0 * Kmax;
#pragma unroll 4
      for (int i = tid; i < nr * Kmax; i += NT) {
        xr[i] = xsrc[i];
        if (dx) acc[i] = 0.f;
      }
      if (tid <= nr) {
        const int sl = ts0 + tid;
        if (prob) {
          // gamma_t(arc) = p_alpha[t][src] wf f_t[slot] p_beta[t+1][dst] * 2^(offs_a[t] + offs_b[t+1] + (r_t + wref) log2e - log2 Z)
          if (tid < nr)
            corr_d[tid] = exp2(offs_a[sl] + offs_b[sl + 1] + ((double)rmaxp[sl] + (double)wref) * kLog2e_d - zd);
          // an epsilon arc at slot t: p_alpha[t][src] factor p_beta[t][dst] * 2^(offs_a[t] + offs_b[t] - log2 Z)
          if (E > 0) corr_eps[tid] = exp2(offs_a[sl] + offs_b[sl] - zd);
        } else {
          const double oa = offs_a[sl == 0 ? 0 : 1 + (sl - 1) / R];
          corr_eps[tid] = oa + offs_b[sl == T ? 0 : 1 + (T - 1 - sl) / R] - zd;
          if (tid < nr) corr_d[tid] = oa + offs_b[sl + 1 == T ? 0 : 1 + (T - 2 - sl) / R] - zd;
        }
      }
    }
    __syncthreads();
    if (!dead) {
      // emission gradient: one thread per (frame, emission slot) sums the posteriors of the slot's
      // arcs (by-slot CSR staged in LDS) -- no atomics, all 256 lanes busy
      if (dx) {
        const int NC = sptr[Kmax + 1];
        const float inv_nc = 1.f / (float)max(NC, 1);
        for (int i = tid; i < nr * NC; i += NT) {
          const int r = fdiv(i, inv_nc);
          const int2 ch = chunk[i - r * NC];
          const int k = ch.x, j1 = min(ch.y + kChunk, sptr[k + 1]);
          float sum = 0.f;
          if (prob) {
            const double* pa = ald + r * d.max_states;
            const double* pb = bed + (r + 1) * d.max_states;
            double dsum = 0.0;
            for (int j = ch.y; j < j1; ++j) {
              const int2 a = sarc[j];
              dsum = fma(pa[a.x & 0xffff] * pb[(unsigned)a.x >> 16], (double)__int_as_float(a.y), dsum);
            }
            sum = (float)(dsum * (corr_d[r] * (double)xr[r * Kmax + k]));
          } else {
            // log posterior of an arc: a sum of doubles (the two scores are ~1e3 apart from each other and from log Z
            // after a few hundred frames), rounded to float only as the argument of the exponential
            const double* pa = ald + r * d.max_states;
            const double* pb = bed + (r + 1) * d.max_states;
            const double xv = (double)xr[r * Kmax + k] + corr_d[r];
            for (int j = ch.y; j < j1; ++j) {
              const int2 a = sarc[j];
              const double v = pa[a.x & 0xffff] + pb[(unsigned)a.x >> 16] + (xv + (double)__int_as_float(a.y));
              sum += fast_exp((float)v);  // exp(-inf) = 0
            }
          }
          if (sptr[k + 1] - sptr[k] <= kChunk)
            acc[r * Kmax + k] = sum;  // the slot's only chunk
          else if (sum != 0.f)
            atomicAdd(&acc[r * Kmax + k], sum);
        }
      }
      // learnable-weight gradient: one arc per thread, frames of the tile in the inner loop
      if (dW) {
        for (int a = tid; a < A; a += NT) {
          const int wid = u.arc_wid[a];
          if (wid < 0) continue;
          const float w = u.arc_w[a] + (weights ? nan_to_neg(weights[wid]) : 0.f);
          const float* px = xr + u.arc_slot[a];
          float wsum = 0.f;
          if (prob) {
            const double* pa = ald + u.arc_src[a];
            const double* pb = bed + d.max_states + u.arc_dst[a];
            double ds = 0.0;
            for (int r = 0; r < nr; ++r)
              ds = fma(pa[r * d.max_states] * pb[r * d.max_states], corr_d[r] * (double)px[r * Kmax], ds);
            wsum = (float)(ds * (double)fast_exp(nan_to_neg(w) - wref));
          } else {
            const double* pa = ald + u.arc_src[a];
            const double* pb = bed + d.max_states + u.arc_dst[a];
            for (int r = 0; r < nr; ++r)
              wsum += fast_exp((float)(pa[r * d.max_states] + pb[r * d.max_states] + ((double)px[r * Kmax] + corr_d[r] + (double)w)));
          }
          if (wsum != 0.f) dwacc[a] += wsum;  // this thread owns dwacc[a]
        }
      }
      if (dW && E > 0) {
        const int nslots = nr + ((ts0 + nr == T) ? 1 : 0);  // epsilon slots t = ts0 .. (T included once)
        for (int i = tid; i < nslots * E; i += NT) {
          const int r = i / E, e = i - r * E;
          const int wid = u.eps_wid[e];
          if (wid < 0) continue;
          if (prob) {
            const double g = ald[r * d.max_states + u.eps_src[e]] * bed[r * d.max_states + u.eps_dst[e]] * (corr_eps[r] * eps_factor(u, weights, e));
            if (g != 0.0) atomicAdd(&dwacc[A + e], (float)g);
            continue;
          }
          const float w = u.eps_w[e] + (weights ? nan_to_neg(weights[wid]) : 0.f);
          const float v = (float)(ald[r * d.max_states + u.eps_src[e]] + bed[r * d.max_states + u.eps_dst[e]] + (corr_eps[r] + (double)w));
          if (v > WFL_NEG_INF) atomicAdd(&dwacc[A + e], fast_exp(v));
        }
      }
    }
    if (dx) {
      __syncthreads();
      stream_grad_rows(b, ts0, nr, T, C, Kmax, tid, NT, dx, x, row_lse, accumulate, dead, cf, acc, colmap);
    }
  }
  if (dW) {
    __syncthreads();
    for (int a = tid; a < A + E; a += NT) {
      const int l = lead[a];
      const float g = dwacc[a];
      if (l != a && g != 0.f) atomicAdd(&dwacc[l], g);  // (LDS; leaders are only added to, the others only read)
    }
    __syncthreads();
    const float cw = (coef_w ? coef_w[b] : 1.f) * g0;
    for (int a = tid; a < A + E; a += NT) {
      if (lead[a] != a) continue;
      const int wid = a < A ? u.arc_wid[a] : u.eps_wid[a - A];
      const float g = dwacc[a];
      if (wid >= 0 && g != 0.f) atomicAdd(&dW[wid], g * cw);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stage 3 for banded acceptors swept in the probability domain (ASG force alignment, chains without skips: every arc
// comes from the state itself or from its neighbour, at most 64 states): one WAVE per block of 16 frames, lane = state,
// no barrier.  The general kernel below stages double-precision tiles of alpha and beta in LDS and walks arc lists
// behind three barriers per tile -- 170-250 us for the force-alignment lattice of the ASG benchmark, next to which
// the sweeps take 120; here a lane reads its own alpha (the neighbour's through a DPP wave shift) and beta doubles,
// forms the two posteriors of its in-arcs, adds them into a compact [16][labels] tile (repeated labels share a
// slot: LDS float atomics) and the wave expands the tile into the dense rows.  Transition gradients accumulate in
// two registers per lane over all blocks of the wave: one global atomic per arc and wave.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    band_grad_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats,
                     const float* __restrict__ xg, int T, int C, const float* __restrict__ weights,
                     const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ logz,
                     const float* __restrict__ coef, const float* __restrict__ coef_w, const float* __restrict__ gout,
                     int accumulate, float* __restrict__ dx, float* __restrict__ dW, int64_t tail, int nch1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef WFL_BANDGRAD_PRIO
#define WFL_BANDGRAD_PRIO 0
#endif
  if (WFL_BANDGRAD_PRIO) __builtin_amdgcn_s_setprio(WFL_BANDGRAD_PRIO);
  constexpr int RB = 16;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const UttView u = make_view(d, ints, floats, b, T);
  const int Q = u.Q, Kmax = d.max_labels;
  const double* offs_a = reinterpret_cast<const double*>(alpha + tail) + (int64_t)b * nch1;
  const double* offs_b = reinterpret_cast<const double*>(beta + tail) + (int64_t)b * nch1;
  const double zd = reinterpret_cast<const double*>(alpha + tail)[(int64_t)d.B * nch1 + b];  // log2 Z
  const int32_t* fmt = reinterpret_cast<const int32_t*>(reinterpret_cast<const double*>(alpha + tail) + (int64_t)d.B * (nch1 + 1));
  const float wref = reinterpret_cast<const float*>(fmt + d.B)[b];
  const BandArcs arcs = band_in_arcs(u, weights, lane);
  const bool shape = fmt[b] == kFmtProb && band_shape(d, u);
  if (!__syncthreads_and(arcs.ok) || !shape) return;  // the general kernel takes this utterance (same test there)
  float* acc = reinterpret_cast<float*>(smem) + (size_t)wave * RB * Kmax;  // [RB][Kmax], this wave's
  int16_t* colmap = reinterpret_cast<int16_t*>(reinterpret_cast<float*>(smem) + (size_t)4 * RB * Kmax);  // [C]
  if (dx) {
    for (int c = tid; c < C; c += 256) colmap[c] = -1;
    __syncthreads();
    for (int k = tid; k < u.K; k += 256) colmap[u.labels[k]] = (int16_t)k;
    __syncthreads();
  }
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = coef ? coef[b] * g0 : g0;
  const float cw = (coef_w ? coef_w[b] : 1.f) * g0;
  const float z = logz[b];
  const bool dead = !(z > WFL_NEG_INF) || !(z < __builtin_inff());  // no accepting path: zero gradient
  const double wfs = arcs.has_self ? (double)fast_exp(arcs.w_self - wref) : 0.0;  // (the sweeps' factors)
  const double wfa = arcs.has_adj ? (double)fast_exp(arcs.w_adj - wref) : 0.0;
  // (lanes without a state read the last state's column and multiply it by arc factors of zero: every lane stays
  // active, so that the per-frame words can be read from ANY lane below -- under `lane < Q` the compiler computes them
  // for those lanes only)
  const double* A_ = reinterpret_cast<const double*>(alpha) + u.ab_base + min(lane, Q - 1);
  const double* B_ = reinterpret_cast<const double*>(beta) + u.ab_base + min(lane, Q - 1);
  const float* fgu = xg + xg_main_dev(d, T) + u.xg_base;                         // factors of the gathered rows
  const float* rmu = xg + 2 * xg_main_dev(d, T) + (int64_t)b * T;                // their references
  const bool one_slot = arcs.slot_self == arcs.slot_adj;
  double sum_s = 0.0, sum_a = 0.0;
  const int nblk = (T + RB - 1) / RB, nw = gridDim.x * 4;
  for (int kb = blockIdx.x * 4 + wave; kb < nblk; kb += nw) {
    const int t0 = kb * RB, n = min(RB, T - t0);
    if (dx)
      for (int i = lane; i < RB * Kmax; i += 64) acc[i] = 0.f;
    if (!dead) {
      // gamma_t(arc) = p_alpha[t][src] wf f_t[slot] p_beta[t+1][dst] 2^e_t,
      // e_t = offs_a[t] + offs_b[t+1] + (r_t + wref) log2e - log2 Z: lane j holds frame t0 + j's as mantissa x 2^exponent
      double e = 0.0;
      if (lane < n) e = offs_a[t0 + lane] + offs_b[t0 + lane + 1] + ((double)rmu[t0 + lane] + (double)wref) * kLog2e_d - zd;
      e = fmin(fmax(e, -2000.0), 2000.0);
      const double ef = floor(e);
      const float cm = __builtin_amdgcn_exp2f((float)(e - ef));
      const int ce = (int)ef;
      // (eight frames at a time: sixteen frames of operands in flight cost the fourth wave per SIMD)
      constexpr int HB = RB / 2;
#pragma unroll 1
      for (int h = 0; h < RB; h += HB) {
        if (h >= n) break;
        double pa[HB], pb[HB];
        float fs[HB], fa[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {  // (rows past the end: the last one again, not consumed)
          const int t = t0 + min(h + j, n - 1);
          pa[j] = A_[(int64_t)t * Q], pb[j] = B_[(int64_t)(t + 1) * Q];
          fs[j] = fgu[(int64_t)t * Kmax + arcs.slot_self], fa[j] = fgu[(int64_t)t * Kmax + arcs.slot_adj];
        }
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          if (h + j < n) {
            const int lo = __double2loint(pa[j]), hi = __double2hiint(pa[j]);  // the neighbour's alpha (lane 0: none)
            const double pn = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false),
                                               __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false));
            const double cj = ldexp((double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(cm), h + j)),
                                    __builtin_amdgcn_readlane(ce, h + j));
            const double right = pb[j] * cj;
            const double ps = pa[j] * wfs * (double)fs[j] * right;
            const double pd = pn * wfa * (double)fa[j] * right;
            sum_s += ps, sum_a += pd;
            if (dx) {
              float* row = acc + (h + j) * Kmax;
              if (one_slot) {
                const float g = (float)(ps + pd);
                if (g != 0.f) atomicAdd(&row[arcs.slot_self], g);
              } else {
                if (ps != 0.0) atomicAdd(&row[arcs.slot_self], (float)ps);
                if (pd != 0.0) atomicAdd(&row[arcs.slot_adj], (float)pd);
              }
            }
          }
        }
      }
    }
    if (dx) {
      // the block's rows are contiguous: dense row value = cf * (the tile entry of the column's label slot)
      float* g = dx + ((int64_t)b * T + t0) * C;
      const int total = n * C;
      const float inv_c = 1.f / (float)C;
      if ((C & 3) == 0) {
        for (int i4 = lane; i4 < (total >> 2); i4 += 64) {
          const int i = i4 << 2;
          const int r = (int)(((float)i + 0.5f) * inv_c), c = i - r * C;  // (exact for i < 2^20)
          float4 v = accumulate ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
          const int k0 = colmap[c], k1 = colmap[c + 1], k2 = colmap[c + 2], k3 = colmap[c + 3];
          if (k0 >= 0) v.x += cf * acc[r * Kmax + k0];
          if (k1 >= 0) v.y += cf * acc[r * Kmax + k1];
          if (k2 >= 0) v.z += cf * acc[r * Kmax + k2];
          if (k3 >= 0) v.w += cf * acc[r * Kmax + k3];
          *reinterpret_cast<float4*>(g + i) = v;
        }
      } else {
        for (int i = lane; i < total; i += 64) {
          const int r = (int)(((float)i + 0.5f) * inv_c), c = i - r * C;
          const int k = colmap[c];
          float v = accumulate ? g[i] : 0.f;
          if (k >= 0) v += cf * acc[r * Kmax + k];
          g[i] = v;
        }
      }
    }
  }
  if (dW && !dead && lane < Q) {
    // (measured at the ASG benchmark with these two compiled out: the kernel and the step do not change)
    if (arcs.wid_self >= 0 && sum_s != 0.0) atomicAdd(&dW[arcs.wid_self], (float)sum_s * cw);
    if (arcs.wid_adj >= 0 && sum_a != 0.0) atomicAdd(&dW[arcs.wid_adj], (float)sum_a * cw);
  }
}

// ------------------------------------------------------------------------------------------------
// tropical back-trace: one thread per utterance
// ------------------------------------------------------------------------------------------------
__global__ void backtrace_kernel(wfl_lattice_desc d, const int32_t* __restrict__ ints, const float* __restrict__ floats,
                                 const float* __restrict__ alpha, const int32_t* __restrict__ bptr, int T,
                                 int32_t* __restrict__ path, int32_t* __restrict__ path_len, int path_stride) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const UttView u = make_view(d, ints, floats, b, T);
  const int Q = u.Q, A = u.A;
  int best = -1;
  float bs = WFL_NEG_INF;
  for (int q = 0; q < Q; ++q) {
    const float v = alpha[u.ab_base + (int64_t)T * Q + q] + u.accept_w[q];
    if (v > bs) bs = v, best = q;
  }
  int32_t* out = path + (int64_t)b * path_stride;
  int n = 0;
  if (best >= 0) {
    int q = best, t = T;
    while (n < path_stride) {
      const int bp = bptr[u.ab_base + (int64_t)t * Q + q];
      if (bp < 0) break;
      if (bp < A) {
        out[n++] = u.arc_orig[bp];
        q = u.arc_src[bp];
        --t;
      } else {
        out[n++] = u.eps_orig[bp - A];
        q = u.eps_src[bp - A];
      }
    }
    for (int i = 0, j = n - 1; i < j; ++i, --j) {
      const int tmp = out[i];
      out[i] = out[j], out[j] = tmp;
    }
  } else {
    n = -1;  // no accepting path
  }
  path_len[b] = n;
}

__global__ void reduce_loss_kernel(const float* __restrict__ vals, const float* __restrict__ minus,
                                   const float* __restrict__ scale, int B, float sign, int accumulate,
                                   float* __restrict__ out) {
  __shared__ float red[64];
  WFL_LOG(13);
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x)
    s += sign * (scale ? scale[b] : 1.f) * (minus ? vals[b] - minus[b] : vals[b]);
  s = block_reduce_sum(s, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s / (float)B;
}

// out[r] = logsumexp_c x[r, c] (NaN = -inf): the forward half of a fused log_softmax.  One wave per row,
// RU rows per iteration with all NV * RU loads of a lane issued before the first use (16 in flight per
// lane: the kernel is one streaming read of x and needs the bytes in flight to reach HBM speed).
template <int NV, int RU>
__global__ void __launch_bounds__(256) row_lse_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RU; r0 < rows; r0 += nw * RU) {
    float v[RU][NV];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float* row = x + min(r0 + u, rows - 1) * C;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[u][i] = c < C ? row[c] : WFL_NEG_INF;
      }
    }
    float m[RU], sum[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      m[u] = WFL_NEG_INF;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[u][i] = nan_to_neg(v[u][i]), m[u] = fmaxf(m[u], v[u][i]);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) m[u] = wave_all_max(m[u]);  // (DPP; the __shfl reductions go through LDS)
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      sum[u] = 0.f;
      if (m[u] > WFL_NEG_INF) {
#pragma unroll
        for (int i = 0; i < NV; ++i) sum[u] += fast_exp(v[u][i] - m[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) sum[u] = wave_all_sum(sum[u]);
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (lane == 0 && r0 + u < rows) out[r0 + u] = m[u] > WFL_NEG_INF ? m[u] + fast_log(sum[u]) : WFL_NEG_INF;
  }
}

// any C: two passes over the row (the second one hits L2)
__global__ void __launch_bounds__(256) row_lse_wide_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                            float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const float* row = x + r * C;
    float m = WFL_NEG_INF;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, nan_to_neg(row[c]));
    m = wave_max(m);
    float s = 0.f;
    if (m > WFL_NEG_INF)
      for (int c = lane; c < C; c += 64) s += fast_exp(nan_to_neg(row[c]) - m);
    s = wave_sum(s);
    if (lane == 0) out[r] = m > WFL_NEG_INF ? m + fast_log(s) : WFL_NEG_INF;
  }
}

// out[r] = first index of the row's maximum (NaN = -inf): one wave per row, the row in registers for C <= 64 NV
template <int NV, int RU>
__global__ void __launch_bounds__(256) row_argmax_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                          int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RU; r0 < rows; r0 += nw * RU) {
    float v[RU][NV];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float* row = x + min(r0 + u, rows - 1) * C;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[u][i] = c < C ? row[c] : WFL_NEG_INF;
      }
    }
    float m[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      m[u] = WFL_NEG_INF;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[u][i] = nan_to_neg(v[u][i]), m[u] = fmaxf(m[u], v[u][i]);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) m[u] = wave_all_max(m[u]);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      int first = 0x3fffffff;
#pragma unroll
      for (int i = NV - 1; i >= 0; --i)
        if (v[u][i] == m[u] && lane + 64 * i < C) first = lane + 64 * i;
      first = -wave_all_max_int(-first);
      if (lane == 0 && r0 + u < rows) out[r0 + u] = first;
    }
  }
}
__global__ void __launch_bounds__(256) row_argmax_wide_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                               int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const float* row = x + r * C;
    float m = WFL_NEG_INF;
    int first = 0x3fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = nan_to_neg(row[c]);
      if (v > m || first == 0x3fffffff) m = v, first = c;  // (ascending c: the first of equals stays)
    }
    const float mm = wave_all_max(m);
    first = -wave_all_max_int(-(m == mm ? first : 0x3fffffff));
    if (lane == 0) out[r] = first;
  }
}

// dst[0..nbytes) = src[0..nbytes): `src` is pinned host memory read through its device-visible address (a kernel
// launch never waits for the stream to drain; hipMemcpyAsync from pinned memory sometimes does, see wfl_upload)
__global__ void __launch_bounds__(256) upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                      int64_t nbytes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0) {
    const int64_t done = n16 * 16;
    if ((int64_t)threadIdx.x < nbytes - done)
      reinterpret_cast<uint8_t*>(dst)[done + threadIdx.x] = reinterpret_cast<const uint8_t*>(src)[done + threadIdx.x];
  }
}

// v *= s[0], skipped when s[0] == 1 (the usual upstream gradient of a scalar loss)
__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ v, int64_t n4, int64_t n, const float* __restrict__ s) {
  const float f = s[0];
  if (f == 1.f) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 t = v4[i];
    t.x *= f, t.y *= f, t.z *= f, t.w *= f;
    v4[i] = t;
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] *= f;
}

}  // namespace wfl

using namespace wfl;

extern "C" {

static int64_t xg_main(const wfl_lattice_desc& d, int T) { return (((int64_t)d.B * T * d.max_labels) + 3) & ~(int64_t)3; }

// Launch shape of the chain kernel: threads per workgroup and emission rows per chunk (also the renormalisation
// interval, so the gradient kernel needs the same number).
static void chain_config(const wfl_lattice_desc& d, int& nt, int& rpc) {
  nt = chain_threads(d);
  rpc = std::max(2, std::min(16, nt * kPre / std::max(1, d.max_labels)) & ~1);  // even (run_chain)
}
// scores (float | double) [..] | double offs[B][nch1] | double Z[B] | int32 fmt[B] | float wref[B]   (see chain_kernel)
static int64_t ab_main_elems(const wfl_lattice_desc& d, int T) {  // (float units; room for doubles)
  return 2 * (d.shared ? (int64_t)d.B * (T + 1) * d.max_states : (int64_t)(T + 1) * d.total_states);
}
static void ab_tail(const wfl_lattice_desc& d, int T, int64_t& tail, int& nch1) {
  nch1 = T + 1;  // one offset per time slot (probability domain); the log domain uses one per chunk
  tail = ab_main_elems(d, T);
}

int wfl_lattice_workspace(const wfl_lattice_desc* d, int T, int64_t* xg_elems, int64_t* ab_elems) {
  if (!d || T < 0) {
    set_error("lattice_workspace: bad arguments");
    return WFL_ERR_INVALID;
  }
  // xg: log-domain rows | probability-domain factors of the same rows | one reference per row
  if (xg_elems) *xg_elems = 2 * xg_main(*d, T) + (int64_t)d->B * T;
  if (ab_elems) {
    int64_t tail;
    int nch1;
    ab_tail(*d, T, tail, nch1);
    // (+ kDumpDoubles doubles behind the tail: where the lanes without a state of the unrolled sweeps "store")
    // (+ behind that: a progress word per utterance, and -- alpha only -- `bad` and the OccHeader lists)
    // (... + the in-flight gradient's header: OccHeader, its per-tile words last)
    *ab_elems = tail + 2 * ((int64_t)d->B * nch1 + d->B) + 2 * (int64_t)d->B + 2 + 2 * kDumpDoubles + 12 * (int64_t)d->B + 32 + 2048 + 4 +
                16 + (int64_t)d->B * (T / kLiveTile + 2);
  }
  return WFL_OK;
}

// where, in the alpha buffer (float units), the int32 [B] sweep formats of a log-semiring forward pass live
int wfl_lattice_formats_offset(const wfl_lattice_desc* d, int T, int64_t* offset) {
  if (!d || T < 0 || !offset) {
    set_error("lattice_formats_offset: bad arguments");
    return WFL_ERR_INVALID;
  }
  int64_t tail;
  int nch1;
  ab_tail(*d, T, tail, nch1);
  *offset = tail + 2 * ((int64_t)d->B * nch1 + d->B);  // behind offs[B][nch1] and Z[B] (doubles)
  return WFL_OK;
}

static int check_desc(const wfl_lattice_desc* d, const char* who) {
  if (!d || d->B <= 0) {
    set_error("%s: empty batch", who);
    return WFL_ERR_INVALID;
  }
  if (d->max_states > 65535 || d->max_labels > 65535) {
    set_error("%s: lattice too large (states=%d labels=%d; limit 65535)", who, d->max_states, d->max_labels);
    return WFL_ERR_UNSUPPORTED;
  }
  return WFL_OK;
}

int wfl_lattice_gather(const wfl_lattice_desc* d, const int32_t* ints, const float* x, int T, int C, float* xg,
                       float* row_lse, void* stream) {
  if (int rc = check_desc(d, "lattice_gather")) return rc;
  if (T <= 0) return WFL_OK;
  auto launch = [&](auto kern, int ru) {
    dim3 grid((unsigned)std::min(1024, (T + 4 * ru - 1) / (4 * ru)), (unsigned)d->B);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, *d, ints, x, T, C, xg, row_lse, xg + xg_main(*d, T),
                       xg + 2 * xg_main(*d, T));
  };
  if (!row_lse && d->max_labels <= 64)
    launch(gather_kernel, 16);  // (its one-label-per-lane path: 64 rows per workgroup and trip, a few trips per wave)
  else if (!row_lse || C > 1024)
    launch(gather_kernel, 1);
  else if (C <= 128)
    launch(gather_lse_kernel<2, 4>, 4);
  else if (C <= 256)
    launch(gather_lse_kernel<4, 2>, 2);
  else if (C <= 512)
    launch(gather_lse_kernel<8, 1>, 1);
  else
    launch(gather_lse_kernel<16, 1>, 1);  // (two rows per wave in flight: 55 instead of 50 us at cfg4 -- 302 MB at 6 TB/s as it is)
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

// One side stream and a join event per device for the gradient that runs beside the sweeps (created on first
// use, never destroyed: they must outlive every stream they were waited on from, and static destruction order is not
// ours to choose).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t join = nullptr;
  hipEvent_t fork = nullptr;       // recorded on the caller's stream in front of the sweeps; the side stream waits for it
  uint32_t* verdicts = nullptr;    // device: one word per call, a ring (occ_gate_kernel -> occ_live_kernel)
  uint32_t next_verdict = 0;
  std::mutex mu;
};
constexpr uint32_t kVerdictRing = 256;
static SideStream* side_stream_of_device() {
  static std::mutex mu;
  static auto* table = new std::map<int, SideStream*>();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto it = table->find(dev);
  if (it != table->end()) return it->second;
  auto* s = new SideStream();
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->join, wfl::order_event_flags()) != hipSuccess ||
      hipEventCreateWithFlags(&s->fork, wfl::order_event_flags()) != hipSuccess ||
      hipMalloc((void**)&s->verdicts, kVerdictRing * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(s->verdicts, 0, kVerdictRing * sizeof(uint32_t)) != hipSuccess) {
    delete s;
    s = nullptr;
  }
  (*table)[dev] = s;
  return s;
}
// The gradient beside the sweeps needs kernels of two streams to run at the same time.  Whether they do is only known
// afterwards: the gate kernel counts its give-ups and its successes in two pinned host words, the host reads them at
// the next call (no synchronisation) and backs off -- the next 1, 2, 4, ... 1024 calls take the plain path, then one
// call tries again; a success resets the back-off.  An environment that announces serialised launches switches the
// path off up front.  All of it is visible through wfl_lattice_diagnostics.
struct LiveState {
  std::mutex mu;
  uint32_t* host = nullptr;  // pinned: [0] give-ups, [1] successes (written by occ_gate_kernel)
  bool env_serial = false;
  uint32_t seen_gave_up = 0, seen_ok = 0;
  uint32_t backoff_left = 0, backoff_len = 0;
  uint64_t attempts = 0, skipped = 0;
  int max_spins = 1 << 11;  // gate polls of ~2 us each: ~4 ms (WFL_LATTICE_GATE_SPINS) -- a process's first launch of the
                            // sweeps loads their code, which takes longer than a millisecond
  bool fork = true;        // the side stream forks from the caller's right in front of the sweeps (without: the round-3 protocol, whose gate waited out the stream's backlog)
};
// (one per DEVICE, next to its side stream: a give-up on one GPU -- a profiler attached to it, a first launch that loads
// code -- says nothing about the others)
static LiveState& live_state() {
  static std::mutex mu;
  static auto* per_device = new std::map<int, LiveState*>();
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  LiveState*& st = (*per_device)[dev];
  if (!st) st = [] {
    auto* s = new LiveState();
    uint32_t* p = nullptr;
    if (hipHostMalloc((void**)&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
      p[0] = p[1] = 0;
      s->host = p;
    }
    for (const char* name : {"AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING"}) {
      const char* e = getenv(name);
      if (e && atoi(e) != 0) s->env_serial = true;
    }
    return s;
  }();
  return *st;
}
// May this call try the gradient beside the sweeps?  (accounts for what earlier gates reported)
static bool live_try(LiveState& ls) {
  std::lock_guard<std::mutex> lock(ls.mu);
  if (!ls.host || ls.env_serial) return false;
  const uint32_t gave = *(volatile uint32_t*)&ls.host[0], ok = *(volatile uint32_t*)&ls.host[1];
  if (ok != ls.seen_ok) ls.seen_ok = ok, ls.backoff_len = 0;
  if (gave != ls.seen_gave_up) {
    ls.seen_gave_up = gave;
    ls.backoff_len = ls.backoff_len ? std::min(ls.backoff_len * 2, 1024u) : 1u;
    ls.backoff_left = ls.backoff_len;
  }
  if (ls.backoff_left > 0) {
    --ls.backoff_left, ++ls.skipped;
    return false;
  }
  ++ls.attempts;
  return true;
}
static int device_cus() {
  int dev = 0, n = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  return n > 0 ? n : 256;
}

// what wfl_lattice_forward_grad adds to the sweeps' launch
struct InLaunchGrad {
  int C;
  const float* coef;
  const float* x;
  const float* row_lse;
  float* dx;
  int done;        // out: 1 if the launch computed the occupancy gradient
  int defer_join;  // in: the caller joins the side stream itself (wfl_lattice_side_join), behind more of its own launches
};
static int lattice_forward_impl(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T,
                                const float* weights, int semiring, float* alpha, float* beta, int32_t* bptr, float* logz,
                                void* stream, InLaunchGrad* g);

int wfl_lattice_forward(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T,
                        const float* weights, int semiring, float* alpha, float* beta, int32_t* bptr, float* logz,
                        void* stream) {
  return lattice_forward_impl(d, ints, floats, xg, T, weights, semiring, alpha, beta, bptr, logz, stream, nullptr);
}

int wfl_lattice_forward_grad(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T,
                             int C, const float* weights, float* alpha, float* beta, float* logz, const float* coef,
                             const float* x, const float* row_lse, float* dx, int* in_launch, void* stream) {
  if (!beta || !dx || !in_launch || (row_lse != nullptr) != (x != nullptr)) {
    set_error("lattice_forward_grad: beta, dx and in_launch are required; x and row_lse go together");
    return WFL_ERR_INVALID;
  }
  InLaunchGrad g{C, coef, x, row_lse, dx, 0, *in_launch == 2};
  const int rc = lattice_forward_impl(d, ints, floats, xg, T, weights, WFL_SEMIRING_LOG, alpha, beta, nullptr, logz, stream, &g);
  *in_launch = g.done;
  return rc;
}

int wfl_lattice_side_join(void* stream) {
  SideStream* side = side_stream_of_device();
  if (!side) return WFL_OK;
  std::lock_guard<std::mutex> lock(side->mu);
  WFL_HIP_CHECK(hipEventRecord(side->join, side->stream));
  WFL_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, side->join, 0));
  return WFL_OK;
}

static int lattice_forward_impl(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T,
                                const float* weights, int semiring, float* alpha, float* beta, int32_t* bptr, float* logz,
                                void* stream, InLaunchGrad* g) {
  if (int rc = check_desc(d, "lattice_forward")) return rc;
  if (!alpha || !logz) {
    set_error("lattice_forward: alpha and logz are required");
    return WFL_ERR_INVALID;
  }
  if (d->max_labels > 4 * 256) {
    set_error("lattice_forward: %d distinct labels per utterance (limit 1024)", d->max_labels);
    return WFL_ERR_UNSUPPORTED;
  }
  int nt, rpc, nch1;
  int64_t tail;
  chain_config(*d, nt, rpc);
  ab_tail(*d, T, tail, nch1);
  const size_t lds = chain_lds_bytes(*d, rpc);
  if (lds > (size_t)kLdsBytes) {
    set_error("lattice_forward: acceptor needs %zu B of LDS (limit %d): %d arcs, %d states", lds, kLdsBytes,
              d->max_arcs, d->max_states);
    return WFL_ERR_UNSUPPORTED;
  }
  if (semiring == WFL_SEMIRING_LOG) {
    dim3 grid((unsigned)d->B, beta ? 2u : 1u);
    auto k = chain_kernel<WFL_SEMIRING_LOG>;
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)k, (int)lds));
    static const int log_only = [] {  // WFL_LATTICE_DOMAIN=log: the fp32 log-domain sweeps throughout (A/B tests)
      const char* e = getenv("WFL_LATTICE_DOMAIN");
      return (e && std::string(e) == "log") ? 1 : 0;
    }();
    SideStream* join_side = nullptr;
    if (!log_only) {
      // probability-domain sweeps of every utterance whose acceptor allows it ...
      auto launch_prob = [&](auto kern) {
        if (lds > 48 * 1024)
          (void)wfl::set_max_dynamic_lds((const void*)kern, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(nt), lds, (hipStream_t)stream, *d, ints, floats, xg, T, rpc, weights, alpha,
                           beta, logz, tail, nch1);
      };
      // The occupancy gradient beside the sweeps (prob_chain_pub_kernel + occ_gate_kernel + occ_live_kernel): uniform-
      // label acceptors without epsilon arcs, full 16-frame chunks (the sweeps' publication counts on them).
      // WFL_LATTICE_FUSED_BADXCD=1: every utterance is reported as swept on two XCDs (tests of the fall-back).
      constexpr int fused_env = 1;
      constexpr int fused_tile = kLiveTile;  // frames per gradient job (16 / 24 / 48 re-measured in round 5 with the sweeps at 160 us: all within 1 %)
      // persistent gradient workgroups per CU: as many as are resident at once (four waves of 130 VGPRs each: three per
      // CU).  More only queue behind those and start when the jobs are gone; measured at the Transducer benchmark with
      // the sweeps at 160 us: 2 -> 0.399 ms, 3 -> 0.362, 4 -> 0.364, 5 (the value until then) -> 0.369, 8 -> 0.37
      constexpr int fused_wgs = 3;
      const char* bad_env = getenv("WFL_LATTICE_FUSED_BADXCD");
      uint32_t token = 0;
      int nt_o = 0;
      // (only while every sweep gets a CU of its own at once: beyond that the gate would wait for the LAST round of
      // sweeps to be placed, i.e. the gradient would start when the sweeps are nearly over)
      if (g && fused_env && beta && T > 0 && nt >= 256 && nt <= 512 && rpc == 16 && d->max_eps == 0 && d->max_labels <= 32767 &&
          2 * ((d->B + 7) & ~7) <= device_cus()) {
        const int ntiles = (T + fused_tile - 1) / fused_tile;
        const int rows_o = (T + ntiles - 1) / ntiles;
        nt_o = (T + rows_o - 1) / rows_o;
        const size_t plds = (size_t)std::max(d->max_states, nt) * 16 + 64 * 4 + prob_rows_floats(*d, rpc, nt) * 4 + (size_t)2 * nt * 4 + 64;
        const size_t olds = occ_lds_bytes(*d, rows_o, g->C);
        // (not while the stream is being captured into a graph: the side stream's launches would not be part of it)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
        LiveState& ls = live_state();
        SideStream* side = olds <= (size_t)kLdsBytes && cap == hipStreamCaptureStatusNone && d->B <= 256 && live_try(ls)
                               ? side_stream_of_device()
                               : nullptr;
        // WFL_LATTICE_FUSED_SERIAL=1 (tests): the gate is launched on the caller's stream IN FRONT of the sweeps, as a
        // stack that runs kernels one after the other would order them -- it cannot see them and gives up
        static const bool serial_test = [] {
          const char* e = getenv("WFL_LATTICE_FUSED_SERIAL");
          return e && atoi(e) != 0;
        }();
        if (side) {
          static std::atomic<uint32_t> counter{0};
          do token = (counter.fetch_add(1) + 1) * 2654435761u; while (token == 0);
          const int Bp = (d->B + 7) & ~7;
          hipStream_t main_s = (hipStream_t)stream;
          std::lock_guard<std::mutex> lock(side->mu);  // (the events are the device's, not the call's)
          // The side stream forks from the caller's stream right in front of the sweeps: the gate then waits for THIS
          // launch's token in the progress words for microseconds (the sweeps are the next thing the caller's stream
          // runs), whatever backlog the stream has -- without the fork it waited out the backlog, and its give-up had to
          // be seconds away.  It writes nothing before every sweep has announced itself.
          uint32_t* verdict = side->verdicts + (side->next_verdict++ % kVerdictRing);
          hipStream_t gate_s = serial_test ? main_s : side->stream;
          if (ls.fork && !serial_test) {
            WFL_HIP_CHECK(hipEventRecord(side->fork, main_s));
            WFL_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
          }
          auto launch_gate = [&]() {
            hipLaunchKernelGGL(occ_gate_kernel, dim3(1), dim3(64), 0, gate_s, *d, alpha, beta, tail, nch1, token,
                               bad_env && atoi(bad_env) == 1 ? 1 : 0, ls.host, verdict, nt_o, ls.max_spins);
          };
          if (serial_test) launch_gate();
          // the sweeps meet in the middle and store occupancies beyond it (run_chain_prob), from kMitmFrames frames on:
          // the crossing and the conversion of the partial chunks are a fixed cost on the sweeps' path, the gradient
          // workgroups' saving grows with T (step at the Transducer benchmark's batch: 320 frames 0.176 either way, 480
          // frames 0.235 -> 0.229 ms, 800 frames 0.352 -> 0.337).  WFL_LATTICE_MITM=0 keeps both vectors everywhere,
          // =2 meets from 64 frames on (A/B, tests; read per call: tests flip it inside one process)
          constexpr int kMitmFrames = 320;
          const char* mitm_e = getenv("WFL_LATTICE_MITM");
          const int mitm_env = mitm_e ? atoi(mitm_e) : 1;
          const int mitm_req = (mitm_env != 0 && T >= (mitm_env == 2 ? 64 : kMitmFrames) && !weights) ? 1 : 0;
          // the gradient workgroups write the rows' base values first and add the occupancies behind the sweeps
          // (occ_live_kernel); WFL_LATTICE_TWO_PHASE=0: whole rows per tile (read per call)
          const char* tp_e = getenv("WFL_LATTICE_TWO_PHASE");
          const int two_phase = (tp_e && atoi(tp_e) == 0) ? 0 : 1;
          auto launch_pub = [&](auto kern) {
            if (plds > 48 * 1024) (void)wfl::set_max_dynamic_lds((const void*)kern, (int)plds);
            hipLaunchKernelGGL(kern, dim3((unsigned)(2 * Bp)), dim3(nt), plds, main_s, *d, ints, floats, xg, T, rpc, weights,
                               alpha, beta, logz, tail, nch1, Bp, token, mitm_req);
          };
          if (nt <= 256)
            launch_pub(prob_chain_pub_kernel<256>);
          else
            launch_pub(prob_chain_pub_kernel<512>);
          if (!serial_test) launch_gate();
          if (olds > 48 * 1024) (void)wfl::set_max_dynamic_lds((const void*)occ_live_kernel, (int)olds);
          const int64_t jobs = (int64_t)d->B * nt_o;
          const unsigned wgs = (unsigned)std::max<int64_t>(8, std::min<int64_t>(jobs, (int64_t)fused_wgs * device_cus()));
          hipLaunchKernelGGL(occ_live_kernel, dim3(wgs), dim3(256), olds, side->stream, *d, ints, floats, T, g->C, alpha, beta,
                             g->coef, g->x, g->row_lse, g->dx, rows_o, nt_o, rpc, tail, nch1, token, verdict, mitm_req, two_phase);
          WFL_HIP_CHECK(hipEventRecord(side->join, side->stream));
          join_side = side;  // (joined below, behind the certificate: it and the log-domain launch overlap the gradient's tail)
          g->done = 1;
        }
      }
      if (g && g->done) {
      } else if (nt == 128 && d->max_states <= 64)
        launch_prob(prob_chain_kernel<128>);  // chain wave + loader wave (the banded sweep)
      else if (nt <= 256)
        launch_prob(prob_chain_kernel<256>);
      else if (nt <= 512)
        launch_prob(prob_chain_kernel<512>);
      else
        launch_prob(prob_chain_kernel<1024>);
      if (beta)
        hipLaunchKernelGGL(prob_certify_kernel, dim3((unsigned)d->B, 8u), dim3(256), 0, (hipStream_t)stream, *d, ints,
                           floats, T, alpha, beta, tail, nch1, nt, (g && g->done) ? 1 : 0, weights);
    }
    // ... then the log-domain sweeps of the rest (and of utterances whose two sweeps disagree: the certificate)
    hipLaunchKernelGGL(k, grid, dim3(nt), lds, (hipStream_t)stream, *d, ints, floats, xg, T, rpc, weights, alpha,
                       beta, (int32_t*)nullptr, logz, tail, nch1, log_only ? 1 : 2);
    // (the gradient beside the sweeps: joined only here -- the certificate and the log-domain launch, which normally
    // finds nothing to do, ran under its tail.  An utterance the log-domain launch re-sweeps while gradient workgroups
    // still read its alpha / beta gets rows of garbage from them; wfl_lattice_grad_rest overwrites exactly those.)
    if (join_side && !(g && g->defer_join)) WFL_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, join_side->join, 0));
  } else if (semiring == WFL_SEMIRING_TROPICAL) {
    if (!bptr) {
      set_error("lattice_forward: tropical semiring needs a back-pointer buffer");
      return WFL_ERR_INVALID;
    }
    dim3 grid((unsigned)d->B, 1u);
    auto k = chain_kernel<WFL_SEMIRING_TROPICAL>;
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)k, (int)lds));
    hipLaunchKernelGGL(k, grid, dim3(nt), lds, (hipStream_t)stream, *d, ints, floats, xg, T, rpc, weights, alpha,
                       (float*)nullptr, bptr, logz, tail, nch1, 1);
  } else {
    set_error("lattice_forward: unknown semiring %d", semiring);
    return WFL_ERR_INVALID;
  }
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

static int lattice_grad_impl(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T, int C,
                             const float* weights, const float* alpha, const float* beta, const float* logz, const float* coef,
                             const float* coef_w, const float* gout, int accumulate, const float* x, const float* row_lse,
                             float* dx, float* dW, void* stream, int occ_in_launch);

int wfl_lattice_grad(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T, int C,
                     const float* weights, const float* alpha, const float* beta, const float* logz, const float* coef,
                     const float* coef_w, const float* gout, int accumulate, const float* x, const float* row_lse,
                     float* dx, float* dW, void* stream) {
  return lattice_grad_impl(d, ints, floats, xg, T, C, weights, alpha, beta, logz, coef, coef_w, gout, accumulate, x, row_lse,
                           dx, dW, stream, 0);
}

// What is left of the gradient after wfl_lattice_forward_grad reported in_launch = 1: the utterances the occupancy
// gradient does not serve (and those the certificate sent to the log-domain sweeps), by the general kernel, which
// overwrites their rows of dx.  The rows the launch wrote were computed with grad_output = 1: scale them first if it
// is not (wfl_scale), then call this with the real gout.
int wfl_lattice_grad_rest(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T, int C,
                          const float* weights, const float* alpha, const float* beta, const float* logz, const float* coef,
                          const float* gout, const float* x, const float* row_lse, float* dx, void* stream) {
  return lattice_grad_impl(d, ints, floats, xg, T, C, weights, alpha, beta, logz, coef, nullptr, gout, 0, x, row_lse, dx,
                           nullptr, stream, 1);
}

static int lattice_grad_impl(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* xg, int T, int C,
                             const float* weights, const float* alpha, const float* beta, const float* logz, const float* coef,
                             const float* coef_w, const float* gout, int accumulate, const float* x, const float* row_lse,
                             float* dx, float* dW, void* stream, int occ_in_launch) {
  if (int rc = check_desc(d, "lattice_grad")) return rc;
  if ((row_lse != nullptr) != (x != nullptr)) {
    set_error("lattice_grad: the fused log-softmax backward needs both x and row_lse");
    return WFL_ERR_INVALID;
  }
  if (!alpha || !beta || !logz || (!dx && !dW)) {
    set_error("lattice_grad: alpha, beta, logz and at least one output are required");
    return WFL_ERR_INVALID;
  }
  if (T <= 0) return WFL_OK;
  // frames per LDS sub-tile: alpha, beta, gathered emissions and the per-label accumulators of TS
  // frames; ~40 KiB at most so that several workgroups are co-resident (each one is a load ->
  // barrier -> compute -> barrier -> stream-out sequence, overlap comes from co-residency).  Measured on MI355X
  // (kernel us at 24 / 32 / 40 / 48 KiB): Transducer cfg4 (263 states, 917 arcs: ONE frame per tile at 24 KiB)
  // 301 / 230 / 239 / 283; ASG force alignment alone 81 / 112 / 78 / 78, under the denominator sweeps 220 / 198 / 174.
  const size_t row_bytes = 16 * (size_t)d->max_states + 4 * (size_t)d->max_labels * (dx ? 2 : 1);  // (alpha, beta: doubles)
  const size_t nae = (size_t)d->max_arcs + d->max_eps;  // (learnable-weight accumulators + the arcs' leaders: grad_kernel)
  const size_t fixed = 16 * (size_t)d->max_states + (dW ? 4 * nae + 4 * ((nae + 1) & ~(size_t)1) : 0) +
                       (dx ? 8 * (size_t)d->max_arcs + 4 * (((size_t)d->max_labels + 3) & ~(size_t)1) +
                                8 * ((size_t)d->max_labels + d->max_arcs / kChunk + 1) + 2 * (size_t)C
                          : 0) +
                       8 * (33 + 34) + 64;  // (+ the per-row offset corrections of at most 32 + 1 rows: doubles)
  if (dx && d->max_labels > 32767) {
    set_error("lattice_grad: %d distinct labels per utterance (limit 32767)", d->max_labels);
    return WFL_ERR_UNSUPPORTED;
  }
  int nt_chain, rpc, nch1;
  int64_t tail;
  chain_config(*d, nt_chain, rpc);
  ab_tail(*d, T, tail, nch1);
  constexpr size_t budget = 40 * 1024;  // LDS per workgroup that sizes the tiles
  int TS = fixed + row_bytes < budget ? (int)((budget - fixed) / row_bytes) : 1;
  TS = std::max(1, std::min(TS, 32));
  const size_t lds = fixed + row_bytes * TS;
  if (lds > (size_t)kLdsBytes) {
    set_error("lattice_grad: needs %zu B of LDS (limit %d)", lds, kLdsBytes);
    return WFL_ERR_UNSUPPORTED;
  }
  // Every workgroup takes about the same time and they all fit the chip at once only up to
  // (resident workgroups per CU) x 256: a grid slightly above that costs a whole second round
  // (measured: 1856 workgroups on 1536 slots = 2 x 250 us).  Size the grid to ONE round.
  int per_cu = 1;
  {  // (the occupancy query costs a few microseconds of host time: remember the last answer)
    static std::mutex mu;
    static size_t last_lds = ~(size_t)0;
    static int last_per_cu = 1;
    std::lock_guard<std::mutex> lock(mu);
    if (lds != last_lds) {
      int n = 1;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, grad_kernel, 256, lds) != hipSuccess || n < 1) n = 1;
      last_lds = lds, last_per_cu = n;
    }
    per_cu = last_per_cu;
  }
  const int slots = per_cu * 256;
  int blocks_t = std::max(1, std::min((T + TS - 1) / TS, slots / std::max(1, d->B)));
  int rows_per_block = (T + blocks_t - 1) / blocks_t;
  blocks_t = (T + rows_per_block - 1) / rows_per_block;
  // banded acceptors swept in the probability domain go to band_grad_kernel; the general launch skips them
  const char* band_env = getenv("WFL_LATTICE_BAND_GRAD");  // (0: the general kernel for everything -- tests, measurements)
  const bool band_off = band_env && atoi(band_env) == 0;
  const int band = !band_off && d->max_states <= 64 && d->max_labels <= 64 && d->max_eps == 0 && !row_lse;
  if (band) {
    const int nblk = (T + 15) / 16;
    const unsigned bx = (unsigned)std::max(1, std::min((nblk + 3) / 4, (1024 + d->B - 1) / d->B));
    const size_t blds = (size_t)4 * 16 * d->max_labels * 4 + (size_t)2 * C + 16;
    hipLaunchKernelGGL(band_grad_kernel, dim3(bx, (unsigned)d->B), dim3(256), blds, (hipStream_t)stream, *d, ints, floats,
                       xg, T, C, weights, alpha, beta, logz, coef, coef_w, gout, accumulate, dx, dW, tail, nch1);
    WFL_LAUNCH_CHECK();
  }
  // uniform-label acceptors without learnable weights (the Transducer's alignment graphs, long CTC targets): the
  // emission gradient from state occupancies (occ_grad_kernel); the general launch skips what it served
  const int occ = dx && !dW && !band && d->max_eps == 0 && d->max_labels <= 32767;
  int occ_done = occ_in_launch ? 2 : 0;
  if (occ && !occ_in_launch) {
    const int wgs_t = std::max(1, std::min((T + 15) / 16, 2048 / std::max(1, d->B)));
    const int rows_o = (T + wgs_t - 1) / wgs_t;
    const int blocks_o = (T + rows_o - 1) / rows_o;
    const int ts_o = std::min(32, rows_o);
    const size_t olds = occ_lds_bytes(*d, ts_o, C);
    if (olds <= (size_t)kLdsBytes) {
      if (olds > 48 * 1024) WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)occ_grad_kernel, (int)olds));
      hipLaunchKernelGGL(occ_grad_kernel, dim3((unsigned)blocks_o, (unsigned)d->B), dim3(256), olds, (hipStream_t)stream, *d,
                         ints, floats, T, C, alpha, beta, logz, coef, gout, accumulate, x, row_lse, dx, rows_o, ts_o, tail,
                         nch1);
      WFL_LAUNCH_CHECK();
      occ_done = 1;
    }
  }
  if (lds > 48 * 1024)
    WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)grad_kernel, (int)lds));
  hipLaunchKernelGGL(grad_kernel, dim3((unsigned)blocks_t, (unsigned)d->B), dim3(256), lds, (hipStream_t)stream, *d,
                     ints, floats, xg, T, C, weights, alpha, beta, logz, coef, coef_w, gout, accumulate, x, row_lse,
                     dx, dW, rows_per_block, TS, tail, nch1, rpc, band, occ_done);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_lattice_backtrace(const wfl_lattice_desc* d, const int32_t* ints, const float* floats, const float* alpha,
                          const int32_t* bptr, int T, int32_t* path, int32_t* path_len, int path_stride, void* stream) {
  if (int rc = check_desc(d, "lattice_backtrace")) return rc;
  hipLaunchKernelGGL(backtrace_kernel, dim3((unsigned)((d->B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, *d, ints,
                     floats, alpha, bptr, T, path, path_len, path_stride);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

#ifdef WFL_LIVE_STATS
int wfl_debug_live_timeline(unsigned long long* tiles, unsigned int* ntiles, unsigned long long* sweeps) {
  int rc = (int)hipMemcpyFromSymbol(tiles, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * 4096 * 5);
  rc |= (int)hipMemcpyFromSymbol(ntiles, HIP_SYMBOL(g_tl_n), sizeof(unsigned int));
  rc |= (int)hipMemcpyFromSymbol(sweeps, HIP_SYMBOL(g_sweep), sizeof(unsigned long long) * 512 * 2);
  unsigned int z = 0;
  rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_n), &z, sizeof(z));
  return rc;
}
int wfl_debug_live_workgroups(unsigned long long* wgs) {
  return (int)hipMemcpyFromSymbol(wgs, HIP_SYMBOL(g_wg), sizeof(unsigned long long) * 1024 * 4);
}
int wfl_debug_log(unsigned long long* log, unsigned int* n) {  // read and reset
  int rc = (int)hipMemcpyFromSymbol(log, HIP_SYMBOL(g_log), sizeof(unsigned long long) * 1024);
  rc |= (int)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_log_n), sizeof(unsigned int));
  unsigned int z = 0;
  rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_log_n), &z, sizeof(z));
  return rc;
}
int wfl_debug_marks(unsigned long long* marks) {  // read and reset
  int rc = (int)hipMemcpyFromSymbol(marks, HIP_SYMBOL(g_marks), sizeof(unsigned long long) * 16);
  unsigned long long z[16];
  for (int i = 0; i < 8; ++i) z[2 * i] = ~0ull, z[2 * i + 1] = 0ull;
  rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_marks), z, sizeof(z));
  return rc;
}
int wfl_debug_live_stats(unsigned long long* out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_live), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_live), z, sizeof(z));
  }
  return rc;
}
#endif
// diagnostic: resident workgroups per CU of the gradient kernel for a given dynamic LDS size
int wfl_debug_grad_occupancy(int lds_bytes) {
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, grad_kernel, 256, (size_t)lds_bytes) != hipSuccess) return -1;
  return n;
}

int wfl_row_lse(const float* x, int64_t rows, int C, float* out, void* stream) {
  if (!x || !out || rows < 0 || C <= 0) {
    set_error("row_lse: bad arguments");
    return WFL_ERR_INVALID;
  }
  if (rows == 0) return WFL_OK;
  auto launch = [&](auto kern, int ru) {
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 4 * ru - 1) / (4 * ru), 1 << 16);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, rows, C, out);
  };
  if (C <= 64)
    launch(row_lse_kernel<1, 16>, 16);
  else if (C <= 128)
    launch(row_lse_kernel<2, 8>, 8);
  else if (C <= 256)
    launch(row_lse_kernel<4, 4>, 4);
  else if (C <= 512)
    launch(row_lse_kernel<8, 2>, 2);
  else if (C <= 1024)
    launch(row_lse_kernel<16, 1>, 1);
  else
    launch(row_lse_wide_kernel, 1);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_lattice_diagnostics(uint64_t* out, int n) {
  if (!out || n < 0) {
    set_error("lattice_diagnostics: bad arguments");
    return WFL_ERR_INVALID;
  }
  LiveState& ls = live_state();
  std::lock_guard<std::mutex> lock(ls.mu);
  const uint64_t v[8] = {ls.attempts,
                         ls.host ? (uint64_t) * (volatile uint32_t*)&ls.host[0] : 0,
                         ls.host ? (uint64_t) * (volatile uint32_t*)&ls.host[1] : 0,
                         ls.skipped,
                         ls.backoff_left,
                         ls.env_serial ? 1u : 0u,
                         (uint64_t)ls.max_spins,
                         ls.fork ? 1u : 0u};
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return WFL_OK;
}

int wfl_row_argmax(const float* x, int64_t rows, int C, int32_t* out, void* stream) {
  if (!x || !out || rows < 0 || C <= 0) {
    set_error("row_argmax: bad arguments");
    return WFL_ERR_INVALID;
  }
  if (rows == 0) return WFL_OK;
  auto launch = [&](auto kern, int ru) {
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 4 * ru - 1) / (4 * ru), 1 << 16);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, rows, C, out);
  };
  if (C <= 64)
    launch(row_argmax_kernel<1, 16>, 16);
  else if (C <= 128)
    launch(row_argmax_kernel<2, 8>, 8);
  else if (C <= 256)
    launch(row_argmax_kernel<4, 4>, 4);
  else if (C <= 512)
    launch(row_argmax_kernel<8, 2>, 2);
  else if (C <= 1024)
    launch(row_argmax_kernel<16, 1>, 1);
  else
    launch(row_argmax_wide_kernel, 1);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_upload(void* dst, const void* src_pinned, int64_t nbytes, void* stream) {
  if (!dst || !src_pinned || nbytes < 0 || (((uintptr_t)dst | (uintptr_t)src_pinned) & 15)) {
    set_error("upload: bad arguments (both buffers must be 16-byte aligned)");
    return WFL_ERR_INVALID;
  }
  if (nbytes == 0) return WFL_OK;
  const int64_t n16 = nbytes / 16;
  const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (n16 + 255) / 256));
  hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(src_pinned),
                     reinterpret_cast<uint4*>(dst), n16, nbytes);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_scale(float* v, int64_t n, const float* s, void* stream) {
  if (!v || !s || n < 0 || (((uintptr_t)v) & 15)) {
    set_error("scale: bad arguments (v must be 16-byte aligned)");
    return WFL_ERR_INVALID;
  }
  if (n == 0) return WFL_OK;
  // (grid-stride; 512 workgroups keep 2 MB in flight -- enough for HBM speed when there is something to scale -- and
  // cost half of what 2048 did when s[0] == 1 and every workgroup returns at once: that launch sits on the critical
  // path of every backward of a scalar loss)
  const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(512, (n / 4 + 255) / 256));
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, n / 4, n, s);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_reduce_loss(const float* vals, const float* minus, const float* scale, int B, float sign, int accumulate,
                    float* out, void* stream) {
  if (B <= 0 || !vals || !out) {
    set_error("reduce_loss: bad arguments");
    return WFL_ERR_INVALID;
  }
  hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, vals, minus, scale, B, sign,
                     accumulate, out);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

}  // extern "C"
