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
#include "device_common.h"

namespace wfl {

struct DenseLds {
  float* W;    // [(C+1) * ldw]
  float* a0;   // [C]
  float* a1;   // [C]
  float* x0;   // [C]
  float* x1;   // [C]
  float* pm;   // [R*C] partial max
  float* ps;   // [R*C] partial sum (or arg for tropical)
  float* red;  // [64]
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
                            float* __restrict__ out, int32_t* __restrict__ bptr, float* __restrict__ logz) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int R = C <= NT ? NT / C : 1;       // partial reductions per state
  const int span = (C + R - 1) / R;         // terms per partial
  const float* xb = x + (int64_t)b * T * C;
  float* ob = out + (int64_t)b * T * C;
  const int t0 = DIR == 0 ? 0 : T - 1;
  float* cur = (t0 & 1) ? L.a1 : L.a0;
  for (int i = tid; i < C; i += NT) {
    const float v = DIR == 0 ? nan_to_neg(xb[i]) + L.W[i] : 0.f;
    cur[i] = v;
    ob[(int64_t)t0 * C + i] = v;
    if (SR == WFL_SEMIRING_TROPICAL) bptr[(int64_t)b * T * C + i] = -1;
  }
  if (DIR == 1 && T > 0)  // the beta sweep consumes x[t+1]; stage row T-1 for the first step
    for (int i = tid; i < C; i += NT) (((T - 1) & 1) ? L.x1 : L.x0)[i] = nan_to_neg(xb[(int64_t)(T - 1) * C + i]);
  if (DIR == 0 && T > 1)
    for (int i = tid; i < C; i += NT) L.x1[i] = nan_to_neg(xb[(int64_t)C + i]);
  __syncthreads();
  for (int step = 1; step < T; ++step) {
    const int t = DIR == 0 ? step : T - 1 - step;       // slot being produced
    const int tf = DIR == 0 ? t - 1 : t + 1;            // slot read
    const float* from = (tf & 1) ? L.a1 : L.a0;
    float* to = (t & 1) ? L.a1 : L.a0;
    const int tx = DIR == 0 ? t : t + 1;                // emissions row used by this step
    const float* xr = (tx & 1) ? L.x1 : L.x0;
    const int txn = DIR == 0 ? t + 1 : t;               // row the next step needs
    const bool has_next = step + 1 < T;
    float pre[4];
    if (has_next) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + j * NT;
        if (i < C) pre[j] = xb[(int64_t)txn * C + i];
      }
    }
    // ---- partial reductions
    for (int idx = tid; idx < R * C; idx += NT) {
      const int s = idx % C, r = idx / C;  // s: state being produced, r: which slice of the other index
      const int j0 = r * span, j1 = min(C, j0 + span);
      float m = WFL_NEG_INF;
      int am = -1;
      for (int j = j0; j < j1; ++j) {
        const float v = DIR == 0 ? from[j] + L.W[(1 + s) * ldw + j] : L.W[(1 + j) * ldw + s] + xr[j] + from[j];
        if (v > m) m = v, am = j;
      }
      float sum = 0.f;
      if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF)
        for (int j = j0; j < j1; ++j) {
          const float v = DIR == 0 ? from[j] + L.W[(1 + s) * ldw + j] : L.W[(1 + j) * ldw + s] + xr[j] + from[j];
          sum += fast_exp(v - m);
        }
      L.pm[idx] = m;
      L.ps[idx] = SR == WFL_SEMIRING_LOG ? sum : __int_as_float(am);
    }
    __syncthreads();
    // ---- merge partials, add the emission (alpha only), publish
    for (int s = tid; s < C; s += NT) {
      float m = L.pm[s];
      int am = SR == WFL_SEMIRING_LOG ? 0 : __float_as_int(L.ps[s]);
      for (int r = 1; r < R; ++r) {
        const float v = L.pm[r * C + s];
        if (v > m) {
          m = v;
          if (SR != WFL_SEMIRING_LOG) am = __float_as_int(L.ps[r * C + s]);
        }
      }
      float val = m;
      if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF) {
        float sum = 0.f;
        for (int r = 0; r < R; ++r) sum += L.ps[r * C + s] * fast_exp(L.pm[r * C + s] - m);
        val = m + fast_log(sum);
      }
      if (DIR == 0) val += xr[s];
      to[s] = val;
      ob[(int64_t)t * C + s] = val;
      if (SR == WFL_SEMIRING_TROPICAL) bptr[((int64_t)b * T + t) * C + s] = am;
    }
    if (has_next) {
      float* xn = (txn & 1) ? L.x1 : L.x0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + j * NT;
        if (i < C) xn[i] = nan_to_neg(pre[j]);
      }
    }
    __syncthreads();
  }
  if (DIR == 0 && logz && T > 0) {
    const float* fin = ((T - 1) & 1) ? L.a1 : L.a0;
    float m = WFL_NEG_INF;
    for (int i = tid; i < C; i += NT) m = fmaxf(m, fin[i]);
    m = blk_max(m, L.red);
    float z = m;
    if (SR == WFL_SEMIRING_LOG && m > WFL_NEG_INF) {
      float s = 0.f;
      for (int i = tid; i < C; i += NT) s += fast_exp(fin[i] - m);
      s = blk_sum(s, L.red);
      z = m + fast_log(s);
    }
    if (tid == 0) logz[b] = z;
  }
}

template <int SR>
__global__ void __launch_bounds__(256)
    dense_chain_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C, int ldw,
                       float* __restrict__ alpha, float* __restrict__ beta, int32_t* __restrict__ bptr,
                       float* __restrict__ logz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, NT = blockDim.x;
  const int R = C <= NT ? NT / C : 1;
  DenseLds L;
  float* p = (float*)smem;
  L.W = p, p += (size_t)(C + 1) * ldw;
  L.a0 = p, p += C;
  L.a1 = p, p += C;
  L.x0 = p, p += C;
  L.x1 = p, p += C;
  L.pm = p, p += (size_t)R * C;
  L.ps = p, p += (size_t)R * C;
  L.red = p;
  for (int i = tid; i < (C + 1) * C; i += NT) L.W[(i / C) * ldw + (i % C)] = nan_to_neg(W[i]);
  __syncthreads();
  if (dir == 0)
    dense_chain<SR, 0>(L, x, b, T, C, ldw, alpha, bptr, logz);
  else
    dense_chain<WFL_SEMIRING_LOG, 1>(L, x, b, T, C, ldw, beta, nullptr, nullptr);
}

static size_t dense_chain_lds(int C, int ldw) {
  const int R = C <= 256 ? 256 / C : 1;
  return 4 * ((size_t)(C + 1) * ldw + 4 * (size_t)C + 2 * (size_t)R * C + 64) + 64;
}

// ------------------------------------------------------------------------------------------------
// gradient
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ void __launch_bounds__(256)
    dense_grad_kernel(const float* __restrict__ x, const float* __restrict__ W, int T, int C,
                      const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ logz,
                      const float* __restrict__ coef, const float* __restrict__ coef_w, const float* __restrict__ gout,
                      int accumulate, float* __restrict__ dx, float* __restrict__ partial, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ap = (float*)smem;  // [C] alpha_{t-1}
  float* xb = ap + C;        // [C] x_t + beta_t - logZ
  const int b = blockIdx.y, tid = threadIdx.x, NT = 256;
  const float g0 = gout ? gout[0] : 1.f;
  const float cf = (coef ? coef[b] : 1.f) * g0;
  const float cw = (coef_w ? coef_w[b] : 1.f) * g0;
  const float z = logz[b];
  const bool dead = !(z > WFL_NEG_INF) || !(z < __builtin_inff());
  const int t_begin = blockIdx.x * rows_per_block, t_end = min(T, t_begin + rows_per_block);
  const int npairs = C * C;
  float acc[NP], wreg[NP];
  int pij[NP];  // (i << 16) | j
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = tid + k * NT;
    acc[k] = 0.f;
    pij[k] = p < npairs ? ((p / C) << 16) | (p % C) : 0;
    wreg[k] = p < npairs ? nan_to_neg(W[C + p]) : WFL_NEG_INF;  // W[1+i, j]
  }
  const int64_t base = (int64_t)b * T * C;
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    for (int i = tid; i < C; i += NT) {
      const float a = alpha[base + (int64_t)t * C + i], be = beta[base + (int64_t)t * C + i];
      if (dx) {
        const float post = dead ? 0.f : fast_exp(a + be - z);
        const int64_t o = base + (int64_t)t * C + i;
        dx[o] = (accumulate ? dx[o] : 0.f) + cf * (post == post ? post : 0.f);
      }
      if (partial) {
        ap[i] = t > 0 ? alpha[base + (int64_t)(t - 1) * C + i] : WFL_NEG_INF;
        xb[i] = nan_to_neg(x[base + (int64_t)t * C + i]) + be - z;
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
    for (int i = tid; i < C; i += NT) {
      float v = 0.f;
      if (t_begin == 0 && !dead) {
        const float p = fast_exp(alpha[base + i] + beta[base + i] - z);
        v = (p == p) ? p * cw : 0.f;
      }
      dst[i] = v;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int p = tid + k * NT;
      if (p < npairs) dst[C + p] = acc[k] * cw;
    }
  }
}

__global__ void __launch_bounds__(256)
    dense_reduce_kernel(const float* __restrict__ partial, int nblk, int n, float* __restrict__ dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += partial[(int64_t)k * n + i];
  dW[i] += s;
}

__global__ void dense_backtrace_kernel(const float* __restrict__ alpha, const int32_t* __restrict__ bptr, int B, int T,
                                       int C, int32_t* __restrict__ path) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || T <= 0) return;
  const float* fin = alpha + ((int64_t)b * T + (T - 1)) * C;
  int cur = 0;
  float best = fin[0];
  for (int i = 1; i < C; ++i)
    if (fin[i] > best) best = fin[i], cur = i;
  int32_t* out = path + (int64_t)b * T;
  for (int t = T - 1; t >= 0; --t) {
    out[t] = cur;
    if (t > 0) cur = bptr[((int64_t)b * T + t) * C + cur];
    if (cur < 0) cur = 0;  // unreachable state (all -inf): keep the path well-formed
  }
}

static int dense_chunks(int B, int T) { return std::max(1, std::min(T, (512 + B - 1) / B)); }

}  // namespace wfl

using namespace wfl;

extern "C" {

static int dense_check(const float* x, const float* W, int B, int T, int C, const char* who) {
  if (!x || !W || B <= 0 || T <= 0 || C <= 0) {
    set_error("%s: bad arguments (B=%d T=%d C=%d)", who, B, T, C);
    return WFL_ERR_INVALID;
  }
  const int ldw = C | 1;
  if (dense_chain_lds(C, ldw) > (size_t)kLdsBytes) {
    set_error("%s: C=%d does not fit the LDS-resident transition matrix (limit about 190 classes)", who, C);
    return WFL_ERR_UNSUPPORTED;
  }
  return WFL_OK;
}

int wfl_dense_forward(const float* x, const float* W, int B, int T, int C, int semiring, float* alpha, float* beta,
                      int32_t* bptr, float* logz, void* stream) {
  if (int rc = dense_check(x, W, B, T, C, "dense_forward")) return rc;
  if (!alpha) {
    set_error("dense_forward: alpha is required");
    return WFL_ERR_INVALID;
  }
  const int ldw = C | 1;
  const size_t lds = dense_chain_lds(C, ldw);
  if (semiring == WFL_SEMIRING_LOG) {
    auto k = dense_chain_kernel<WFL_SEMIRING_LOG>;
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)B, beta ? 2u : 1u), dim3(256), lds, (hipStream_t)stream, x, W, T, C, ldw,
                       alpha, beta, (int32_t*)nullptr, logz);
  } else {
    if (!bptr) {
      set_error("dense_forward: tropical semiring needs a back-pointer buffer");
      return WFL_ERR_INVALID;
    }
    auto k = dense_chain_kernel<WFL_SEMIRING_TROPICAL>;
    if (lds > 48 * 1024)
      WFL_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)B, 1u), dim3(256), lds, (hipStream_t)stream, x, W, T, C, ldw, alpha,
                       (float*)nullptr, bptr, logz);
  }
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_dense_workspace(int B, int T, int C, int64_t* partial_elems) {
  if (B <= 0 || T <= 0 || C <= 0 || !partial_elems) {
    set_error("dense_workspace: bad arguments");
    return WFL_ERR_INVALID;
  }
  *partial_elems = (int64_t)B * dense_chunks(B, T) * (int64_t)(C + 1) * C;
  return WFL_OK;
}

int wfl_dense_grad(const float* x, const float* W, int B, int T, int C, const float* alpha, const float* beta,
                   const float* logz, const float* coef, const float* coef_w, const float* gout, int accumulate,
                   float* dx, float* dW, float* dW_partial, void* stream) {
  if (int rc = dense_check(x, W, B, T, C, "dense_grad")) return rc;
  if (!alpha || !beta || !logz || (!dx && !dW) || (dW && !dW_partial)) {
    set_error("dense_grad: missing buffers");
    return WFL_ERR_INVALID;
  }
  const int chunks = dense_chunks(B, T);
  const int rows = (T + chunks - 1) / chunks;
  const int np = (C * C + 255) / 256;
  const size_t lds = 8 * (size_t)C + 64;
  dim3 grid((unsigned)chunks, (unsigned)B);
  float* part = dW ? dW_partial : nullptr;
#define WFL_DENSE_GRAD(NP)                                                                                         \
  hipLaunchKernelGGL(dense_grad_kernel<NP>, grid, dim3(256), lds, (hipStream_t)stream, x, W, T, C, alpha, beta,    \
                     logz, coef, coef_w, gout, accumulate, dx, part, rows)
  if (np <= 4)
    WFL_DENSE_GRAD(4);
  else if (np <= 16)
    WFL_DENSE_GRAD(16);
  else if (np <= 40)
    WFL_DENSE_GRAD(40);
  else if (np <= 64)
    WFL_DENSE_GRAD(64);
  else if (!dW)
    WFL_DENSE_GRAD(4);  // emission gradient only: the pair accumulators are unused
  else {
    set_error("dense_grad: C=%d too large for the register-resident transition-gradient kernel", C);
    return WFL_ERR_UNSUPPORTED;
  }
#undef WFL_DENSE_GRAD
  WFL_LAUNCH_CHECK();
  if (dW) {
    const int n = (C + 1) * C;
    hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dW_partial, B * chunks, n, dW);
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
  if (int rc = wfl_dense_forward(x, W, B, T, C, WFL_SEMIRING_TROPICAL, alpha, nullptr, bptr, nullptr, stream)) return rc;
  hipLaunchKernelGGL(dense_backtrace_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, alpha,
                     bptr, B, T, C, path);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

}  // extern "C"
