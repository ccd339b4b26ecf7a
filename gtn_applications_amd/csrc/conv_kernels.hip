// ConvTransduce1D (criterions/transducer.py:351-556): for every window of `ks` frames (stride
// `stride`) of every utterance and every lexicon entry k, the forward (or Viterbi) score of
//   intersect(window emissions, make_kernel_graph(lexicon[k]))            transducer.py:489-500
// and the gradient of those B*Tout*K scores w.r.t. the emissions and the (optional) per-arc kernel
// parameters (transducer.py:514-552).  The kernel graphs have a fixed CTC-like shape
// (transducer.py:351-367): state 2i = blank before sub-token i, state 2i+1 = sub-token i, arcs
//   2i->2i (blank)  2i->2i+1 (tok_i)  2i+1->2i+1 (tok_i, unless spike)  2i+1->2i+2 (blank)
//   2i-1->2i+1 (tok_i, if blank_optional and tok_{i-1} != tok_i)
// start {0}, accept {2L} plus {2L-1} if blank_optional.  They are never built on the device: a row of
// 16 lanes runs one (window, entry) dynamic program with lane i owning states 2i and 2i+1, the one
// cross-lane value per frame comes from a DPP row shift (rows of 16 lanes are isolated by the
// hardware), and a workgroup (16 rows) walks all K entries of one window, whose emissions are staged
// once in LDS.  Millions of tiny independent DPs: throughput-, not latency-bound.
#include "device_common.h"

namespace wfl {

constexpr int kConvMaxKs = 16;   // frames per window (register arrays are sized for it)
constexpr int kConvTab = 36;     // int32 words per lexicon entry: L, skip mask, tok[16], arc base[16], pad

template <int CTRL>
__device__ __forceinline__ float row_dpp(float fill, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// lane i of a 16-lane row receives lane i-1 / i+1 (row_shr:1 = 0x111, row_shl:1 = 0x101); edge lanes get `fill`
__device__ __forceinline__ float row_prev(float v, float fill) { return row_dpp<0x111>(fill, v); }
__device__ __forceinline__ float row_next(float v, float fill) { return row_dpp<0x101>(fill, v); }
__device__ __forceinline__ float row_sum16(float v) {  // every lane of the row receives the row's sum
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == WFL_NEG_INF) return WFL_NEG_INF;
  return m + fast_log(fast_exp(a - m) + fast_exp(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == WFL_NEG_INF) return WFL_NEG_INF;
  return m + fast_log(fast_exp(a - m) + fast_exp(b - m) + fast_exp(c - m));
}

struct ConvLane {  // what lane i of a row knows about entry k
  int L, tok;
  bool has_b, has_t;                       // states 2i / 2i+1 exist
  float w_bself, w_tbprev, w_bt, w_tself, w_skip;  // weights of the five arcs ENTERING the lane's states
  int a_bself, a_tbprev, a_bt, a_tself, a_skip;    // their indices in kernel_params (-1: arc absent)
};

__device__ __forceinline__ ConvLane conv_lane(const int32_t* __restrict__ ktab, int k, int i, int spike,
                                              const float* __restrict__ params) {
  const int32_t* e = ktab + (size_t)k * kConvTab;
  ConvLane c;
  c.L = e[0];
  const int skipmask = e[1];
  c.has_b = i <= c.L, c.has_t = i < c.L;
  c.tok = c.has_t ? e[2 + i] : 0;
  const int ns = spike ? 0 : 1;
  const int base_i = c.has_t ? e[18 + i] : 0, base_p = (i >= 1 && i - 1 < c.L) ? e[18 + i - 1] : 0;
  c.a_bself = !c.has_b ? -1 : (i == 0 ? e[18 + 16] : base_p + 2 + ns);
  c.a_tbprev = (c.has_b && i >= 1) ? base_p + 1 + ns : -1;
  c.a_bt = c.has_t ? base_i : -1;
  c.a_tself = (c.has_t && ns) ? base_i + 1 : -1;
  c.a_skip = (c.has_t && ((skipmask >> i) & 1)) ? base_i + 3 + ns : -1;
  auto wt = [&](int a) { return a < 0 ? WFL_NEG_INF : (params ? nan_to_neg(params[a]) : 0.f); };
  c.w_bself = wt(c.a_bself), c.w_tbprev = wt(c.a_tbprev), c.w_bt = wt(c.a_bt);
  c.w_tself = wt(c.a_tself), c.w_skip = wt(c.a_skip);
  return c;
}

// one frame of the alpha sweep; returns the new (blank, token) scores, optionally the argmax codes
template <int SR>
__device__ __forceinline__ void conv_alpha_step(const ConvLane& c, float ab, float al, float xb, float xl, float& nb,
                                                float& nl, int& code_b, int& code_l) {
  const float pal = row_prev(al, WFL_NEG_INF);
  const float b0 = pal + c.w_tbprev, b1 = ab + c.w_bself;            // in-arc order: lower arc id first
  const float l0 = ab + c.w_bt, l1 = al + c.w_tself, l2 = pal + c.w_skip;
  if (SR == WFL_SEMIRING_LOG) {
    nb = lse2(b0, b1) + xb;
    nl = lse3(l0, l1, l2) + xl;
  } else {
    code_b = b1 > b0 ? 1 : 0;
    nb = fmaxf(b0, b1) + xb;
    float m = l0;
    code_l = 0;
    if (l1 > m) m = l1, code_l = 1;
    if (l2 > m) m = l2, code_l = 2;
    nl = m + xl;
  }
  if (!c.has_b) nb = WFL_NEG_INF;
  if (!c.has_t) nl = WFL_NEG_INF;
}

// score of the window: accept states 2L (lane L, blank) and, if blank_optional, 2L-1 (lane L-1, token)
template <int SR>
__device__ __forceinline__ float conv_score(const ConvLane& c, float ab, float al, int i, int blank_optional,
                                            int& best_is_tok) {
  const float fb = __shfl(ab, c.L, 16);
  const float fl = (blank_optional && c.L > 0) ? __shfl(al, max(c.L - 1, 0), 16) : WFL_NEG_INF;
  best_is_tok = fl >= fb && fl > WFL_NEG_INF;  // ties: the lower state id (2L-1) wins
  if (SR == WFL_SEMIRING_LOG) return lse2(fb, fl);
  return fmaxf(fb, fl);
}

template <int SR>
__global__ void __launch_bounds__(256)
    conv_forward_kernel(const float* __restrict__ x, int T, int C, const int32_t* __restrict__ ktab, int K, int ks,
                        int stride, int blank, int spike, int blank_optional, const float* __restrict__ params,
                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float xwin[];  // [ks][C]
  const int win = blockIdx.x, b = blockIdx.y, Tout = gridDim.x;
  const int tid = threadIdx.x, row = tid >> 4, i = tid & 15;
  const float* src = x + ((int64_t)b * T + (int64_t)win * stride) * C;
  for (int e = tid; e < ks * C; e += 256) xwin[e] = nan_to_neg(src[e]);
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int k = min(k0 + row, K - 1);
    const ConvLane c = conv_lane(ktab, k, i, spike, params);
    float ab = i == 0 ? 0.f : WFL_NEG_INF, al = WFL_NEG_INF;
#pragma unroll
    for (int f = 0; f < kConvMaxKs; ++f) {
      if (f < ks) {
        float nb, nl;
        int cb, cl;
        conv_alpha_step<SR>(c, ab, al, xwin[f * C + blank], xwin[f * C + c.tok], nb, nl, cb, cl);
        ab = nb, al = nl;
      }
    }
    int bt;
    const float s = conv_score<SR>(c, ab, al, i, blank_optional, bt);
    if (i == 0 && k0 + row < K) out[((int64_t)b * Tout + win) * K + k] = s;
  }
}

template <int SR>
__global__ void __launch_bounds__(256)
    conv_grad_kernel(const float* __restrict__ x, int T, int C, const int32_t* __restrict__ ktab, int K, int ks,
                     int stride, int blank, int spike, int blank_optional, const float* __restrict__ params,
                     const float* __restrict__ delta, float* __restrict__ dx, float* __restrict__ dparams) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xwin = smem;            // [ks][C]
  float* rows = smem + ks * C;   // [ks][C] gradient of this window, summed over all entries
  const int win = blockIdx.x, b = blockIdx.y, Tout = gridDim.x;
  const int tid = threadIdx.x, row = tid >> 4, i = tid & 15;
  const float* src = x + ((int64_t)b * T + (int64_t)win * stride) * C;
  for (int e = tid; e < ks * C; e += 256) xwin[e] = nan_to_neg(src[e]), rows[e] = 0.f;
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int k = min(k0 + row, K - 1);
    const bool real = k0 + row < K;
    const ConvLane c = conv_lane(ktab, k, i, spike, params);
    const float dl = real ? delta[((int64_t)b * Tout + win) * K + k] : 0.f;
    // alpha sweep, history in registers: hb[f], hl[f] = scores BEFORE frame f
    float hb[kConvMaxKs], hl[kConvMaxKs];
    int codes_b = 0, codes_l = 0;
    float ab = i == 0 ? 0.f : WFL_NEG_INF, al = WFL_NEG_INF;
#pragma unroll
    for (int f = 0; f < kConvMaxKs; ++f) {
      hb[f] = ab, hl[f] = al;
      if (f < ks) {
        float nb, nl;
        int cb = 0, cl = 0;
        conv_alpha_step<SR>(c, ab, al, xwin[f * C + blank], xwin[f * C + c.tok], nb, nl, cb, cl);
        ab = nb, al = nl;
        codes_b |= cb << f, codes_l |= cl << (2 * f);
      }
    }
    int best_is_tok;
    const float z = conv_score<SR>(c, ab, al, i, blank_optional, best_is_tok);
    const bool alive = z > WFL_NEG_INF && dl != 0.f;  // block-row uniform (z, dl are per row)
    float g_bself = 0.f, g_tbprev = 0.f, g_bt = 0.f, g_tself = 0.f, g_skip = 0.f;  // sums over frames
    if (SR == WFL_SEMIRING_LOG) {
      // beta sweep with arc posteriors; beta_ks = 0 on accept states
      float bb = (i == c.L) ? 0.f : WFL_NEG_INF;
      float bl = (blank_optional && c.L > 0 && i == c.L - 1) ? 0.f : WFL_NEG_INF;
#pragma unroll
      for (int f = kConvMaxKs - 1; f >= 0; --f) {
        if (f < ks) {
          const float xb = xwin[f * C + blank], xl = xwin[f * C + c.tok];
          const float pal = row_prev(hl[f], WFL_NEG_INF);
          // posteriors of the arcs entering this lane's states at frame f
          const float eb = xb + bb - z, el = xl + bl - z;
          const float p_bself = alive ? fast_exp(hb[f] + c.w_bself + eb) : 0.f;
          const float p_tbprev = alive ? fast_exp(pal + c.w_tbprev + eb) : 0.f;
          const float p_bt = alive ? fast_exp(hb[f] + c.w_bt + el) : 0.f;
          const float p_tself = alive ? fast_exp(hl[f] + c.w_tself + el) : 0.f;
          const float p_skip = alive ? fast_exp(pal + c.w_skip + el) : 0.f;
          g_bself += p_bself, g_tbprev += p_tbprev, g_bt += p_bt, g_tself += p_tself, g_skip += p_skip;
          const float gtok = dl * (p_bt + p_tself + p_skip);
          const float gblank = row_sum16(dl * (p_bself + p_tbprev));
          if (c.has_t && gtok != 0.f) atomicAdd(&rows[f * C + c.tok], gtok);
          if (i == 0 && gblank != 0.f) atomicAdd(&rows[f * C + blank], gblank);
          // beta of the states before frame f: out-arcs; the neighbour's terms arrive by row shift
          const float to_b = c.w_tbprev + xb + bb, to_l = c.w_skip + xl + bl;  // arcs leaving lane i-1's token
          const float nb = lse2(c.w_bself + xb + bb, c.w_bt + xl + bl);
          const float nl = lse3(c.w_tself + xl + bl, row_next(to_b, WFL_NEG_INF), row_next(to_l, WFL_NEG_INF));
          bb = c.has_b ? nb : WFL_NEG_INF;
          bl = c.has_t ? nl : WFL_NEG_INF;
        }
      }
    } else {
      // tropical: delta flows along the single best path (transducer.py:492-495 + gtn.backward)
      int cur = best_is_tok ? c.L - 1 : c.L, is_tok = best_is_tok;  // row-uniform
#pragma unroll
      for (int f = kConvMaxKs - 1; f >= 0; --f) {
        if (f < ks && alive) {
          const int cb = (__shfl(codes_b, cur, 16) >> f) & 1, cl = (__shfl(codes_l, cur, 16) >> (2 * f)) & 3;
          const bool me = i == cur;
          if (is_tok) {
            if (me) {
              atomicAdd(&rows[f * C + c.tok], dl);
              if (cl == 0) g_bt += 1.f; else if (cl == 1) g_tself += 1.f; else g_skip += 1.f;
            }
            if (cl == 0) is_tok = 0;            // came from blank state 2*cur
            else if (cl == 2) cur -= 1;         // skip arc from token cur-1
          } else {
            if (me) {
              atomicAdd(&rows[f * C + blank], dl);
              if (cb == 0) g_tbprev += 1.f; else g_bself += 1.f;
            }
            if (cb == 0) cur -= 1, is_tok = 1;  // came from token cur-1
          }
        }
      }
    }
    if (dparams && alive && real) {
      if (c.a_bself >= 0 && g_bself != 0.f) atomicAdd(&dparams[c.a_bself], dl * g_bself);
      if (c.a_tbprev >= 0 && g_tbprev != 0.f) atomicAdd(&dparams[c.a_tbprev], dl * g_tbprev);
      if (c.a_bt >= 0 && g_bt != 0.f) atomicAdd(&dparams[c.a_bt], dl * g_bt);
      if (c.a_tself >= 0 && g_tself != 0.f) atomicAdd(&dparams[c.a_tself], dl * g_tself);
      if (c.a_skip >= 0 && g_skip != 0.f) atomicAdd(&dparams[c.a_skip], dl * g_skip);
    }
  }
  __syncthreads();
  // windows overlap when stride < ks: accumulate into dx (zeroed by the host wrapper)
  float* dst = dx + ((int64_t)b * T + (int64_t)win * stride) * C;
  for (int e = tid; e < ks * C; e += 256) {
    const float v = rows[e];
    if (v != 0.f) atomicAdd(&dst[e], v);
  }
}

}  // namespace wfl

using namespace wfl;

namespace wfl {

// ------------------------------------------------------------------------------------------------
// STC: the alphabet augmentation of stc.py:199-220 in one launch each way (the reference does it with six torch ops over
// [B, T, C] and autograd keeps their intermediates).  From log-probabilities x [T, B, C] (the module's input layout) and
// the K selected classes of a batch (the blank, 0, first):
//     out[b, t, k]         = x[t, b, select[k]]                                  k < K
//     out[b, t, K]         = lse = logsumexp_{c >= 1} x[t, b, c]                 <star>
//     out[b, t, K + k]     = lse + log1p(1e-7 - exp(x[t, b, select[k]] - lse))   1 <= k < K    <star> \ token
// and its gradient
//     dx[t, b, c] = [c >= 1] p_c A + [inv[c] >= 0] g[inv[c]] - [inv[c] >= 1] g[K + inv[c]] w_{inv[c]}
//     p_c = exp(x_c - lse),  w_k = u_k / (1 + 1e-7 - u_k),  u_k = exp(x[select[k]] - lse),  A = g[K] + sum_{k >= 1} g[K + k] (1 + w_k)
// One wave per (t, b) row; the row's log-sum-exp is kept for the backward.  expf / log1pf, not the fast intrinsics: the
// result is compared with torch's to 1e-6.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stc_augment_kernel(const float* __restrict__ x, int T, int B, int C,
                                                         const int32_t* __restrict__ select, int K, float* __restrict__ out,
                                                         float* __restrict__ lse_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // t * B + b
  if (row >= (int64_t)T * B) return;
  const int t = (int)(row / B), b = (int)(row % B);
  const float* xr = x + row * C;
  float m = WFL_NEG_INF;
  for (int c = 1 + lane; c < C; c += 64) m = fmaxf(m, nan_to_neg(xr[c]));
  m = wave_all_max(m);
  float sum = 0.f;
  if (m > WFL_NEG_INF)
    for (int c = 1 + lane; c < C; c += 64) sum += expf(nan_to_neg(xr[c]) - m);
  sum = wave_all_sum(sum);
  const float lse = m > WFL_NEG_INF ? m + logf(sum) : WFL_NEG_INF;
  float* o = out + ((int64_t)b * T + t) * 2 * K;
  if (lane == 0) {
    o[K] = lse;
    lse_out[row] = lse;
  }
  for (int k = lane; k < K; k += 64) {
    const float v = xr[select[k]];
    o[k] = v;
    if (k >= 1) o[K + k] = lse + log1pf(1e-7f - expf(v - lse));
  }
}

__global__ void __launch_bounds__(256) stc_augment_grad_kernel(const float* __restrict__ x, int T, int B, int C,
                                                              const int32_t* __restrict__ select, const int32_t* __restrict__ inv,
                                                              int K, const float* __restrict__ lse_in, const float* __restrict__ g,
                                                              float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)T * B) return;
  const int t = (int)(row / B), b = (int)(row % B);
  const float* xr = x + row * C;
  const float* gr = g + ((int64_t)b * T + t) * 2 * K;
  const float lse = lse_in[row];
  float acc = 0.f;
  for (int k = 1 + lane; k < K; k += 64) {
    const float u = expf(xr[select[k]] - lse);
    acc += gr[K + k] * (1.f + u / (1.f + 1e-7f - u));
  }
  const float A = gr[K] + wave_all_sum(acc);
  float* d = dx + row * C;
  for (int c = lane; c < C; c += 64) {
    const float xc = xr[c];
    float v = c >= 1 ? expf(xc - lse) * A : 0.f;
    const int k = inv[c];
    if (k >= 0) v += gr[k];
    if (k >= 1) {
      const float u = expf(xc - lse);
      v -= gr[K + k] * (u / (1.f + 1e-7f - u));
    }
    d[c] = v;
  }
}
}  // namespace wfl

extern "C" {

static int conv_check(const float* x, int B, int T, int C, const int32_t* ktab, int K, int ks, int stride, int blank,
                      const char* who, int& Tout, size_t lds_rows) {
  if (!x || !ktab || B <= 0 || C <= 0 || K <= 0 || stride <= 0 || blank < 0 || blank >= C) {
    set_error("%s: bad arguments", who);
    return WFL_ERR_INVALID;
  }
  if (T < ks) {  // transducer.py:468-470
    set_error("%s: input (%d) too short for kernel (%d)", who, T, ks);
    return WFL_ERR_INVALID;
  }
  if (ks < 1 || ks > kConvMaxKs) {
    set_error("%s: kernel size %d not supported (1..%d)", who, ks, kConvMaxKs);
    return WFL_ERR_UNSUPPORTED;
  }
  if (lds_rows * ks * C * 4 > (size_t)kLdsBytes) {
    set_error("%s: window of %d frames x %d classes does not fit LDS", who, ks, C);
    return WFL_ERR_UNSUPPORTED;
  }
  Tout = (T - ks) / stride + 1;
  return WFL_OK;
}

int wfl_conv_forward(const float* x, int B, int T, int C, const int32_t* ktab, int K, int ks, int stride, int blank,
                     int flags, const float* params, int semiring, float* out, void* stream) {
  int Tout = 0;
  if (int rc = conv_check(x, B, T, C, ktab, K, ks, stride, blank, "conv_forward", Tout, 1)) return rc;
  if (!out) {
    set_error("conv_forward: out is required");
    return WFL_ERR_INVALID;
  }
  const size_t lds = (size_t)ks * C * 4;
  const dim3 grid((unsigned)Tout, (unsigned)B);
  const int spike = flags & WFL_CONV_SPIKE ? 1 : 0, bo = flags & WFL_CONV_BLANK_OPTIONAL ? 1 : 0;
  auto launch = [&](auto kern) -> int {
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)kern, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, x, T, C, ktab, K, ks, stride, blank, spike, bo,
                       params, out);
    return WFL_OK;
  };
  if (int rc = semiring == WFL_SEMIRING_LOG ? launch(conv_forward_kernel<WFL_SEMIRING_LOG>)
                                             : launch(conv_forward_kernel<WFL_SEMIRING_TROPICAL>))
    return rc;
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_conv_grad(const float* x, int B, int T, int C, const int32_t* ktab, int K, int ks, int stride, int blank,
                  int flags, const float* params, int semiring, const float* delta, float* dx, float* dparams,
                  void* stream) {
  int Tout = 0;
  if (int rc = conv_check(x, B, T, C, ktab, K, ks, stride, blank, "conv_grad", Tout, 2)) return rc;
  if (!delta || !dx) {
    set_error("conv_grad: delta and dx are required");
    return WFL_ERR_INVALID;
  }
  WFL_HIP_CHECK(hipMemsetAsync(dx, 0, (size_t)B * T * C * 4, (hipStream_t)stream));
  const size_t lds = (size_t)2 * ks * C * 4;
  const dim3 grid((unsigned)Tout, (unsigned)B);
  const int spike = flags & WFL_CONV_SPIKE ? 1 : 0, bo = flags & WFL_CONV_BLANK_OPTIONAL ? 1 : 0;
  auto launch = [&](auto kern) -> int {
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(wfl::set_max_dynamic_lds((const void*)kern, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, x, T, C, ktab, K, ks, stride, blank, spike, bo,
                       params, delta, dx, dparams);
    return WFL_OK;
  };
  if (int rc = semiring == WFL_SEMIRING_LOG ? launch(conv_grad_kernel<WFL_SEMIRING_LOG>)
                                             : launch(conv_grad_kernel<WFL_SEMIRING_TROPICAL>))
    return rc;
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_stc_augment(const float* x, int T, int B, int C, const int32_t* select, int K, float* out, float* lse, void* stream) {
  if (!x || !select || !out || !lse || T <= 0 || B <= 0 || C <= 1 || K <= 0 || K > C) {
    wfl::set_error("stc_augment: bad arguments (T=%d B=%d C=%d K=%d)", T, B, C, K);
    return WFL_ERR_INVALID;
  }
  const int64_t rows = (int64_t)T * B;
  hipLaunchKernelGGL(wfl::stc_augment_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, T, B, C, select, K,
                     out, lse);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_stc_augment_grad(const float* x, int T, int B, int C, const int32_t* select, const int32_t* inv, int K, const float* lse,
                         const float* g, float* dx, void* stream) {
  if (!x || !select || !inv || !lse || !g || !dx || T <= 0 || B <= 0 || C <= 1 || K <= 0 || K > C) {
    wfl::set_error("stc_augment_grad: bad arguments (T=%d B=%d C=%d K=%d)", T, B, C, K);
    return WFL_ERR_INVALID;
  }
  const int64_t rows = (int64_t)T * B;
  hipLaunchKernelGGL(wfl::stc_augment_grad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, T, B, C,
                     select, inv, K, lse, g, dx);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

}  // extern "C"
