// CTC fast path for gfx950: create_ctc_graph + intersect + forward_score + backward of
// criterions/ctc.py:15-94 without any graph.
//
// Lane mapping (one 64-lane wavefront per utterance and direction): lane i owns target position i,
// i.e. the blank state 2i ("ab") and the label state 2i+1 ("al") of the CTC label graph
// (ctc.py:18-27); lane L owns the trailing blank.  One frame of the recursion needs exactly one
// cross-lane value (al of lane i-1), fetched with a DPP wave shift -- no LDS on the dependent
// chain:
//     ab' = xb   + LSE(ab, al[i-1])
//     al' = xl_i + LSE(al, ab, skip_i ? al[i-1] : -inf)         skip_i = (y_i != y_{i-1})
// The beta sweep is the same recursion on the reversed target and reversed time (the CTC graph is
// mirror-symmetric), so one routine serves both directions; it stores the value BEFORE the
// emission is added, so that posterior(t, s) = alpha_t(s) * beta~_t(s) / Z needs no emission.
// Scores are kept in base-2 log units (v_exp_f32 / v_log_f32 are base 2): no multiplies on the
// dependent chain.  -inf is represented by a large finite sentinel so the chain is branch-free.
// Emissions are gathered straight from the [B,T,C] tensor (one dword per lane per frame, all
// addresses of a wave inside one C-float row) with an 8-frame register prefetch ring.
//
//   stage A  ctc_chain_kernel   grid (B, 2): alpha and beta chains run concurrently
//   stage B  ctc_grad_kernel    all CUs: one wave per (b, t) row: posteriors, label reduction in
//                               LDS, dense row store (zeros included, as ctc.py:75 returns)
#include "device_common.h"

namespace wfl {

constexpr float kNegBig = -1.0e30f;          // stands in for -inf on the chain
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kCtcPrefetch = 8;

__device__ __forceinline__ float wave_shr1(float v, float fill) {
  // lane i receives lane i-1's value; lane 0 receives `fill` (DPP wave_shr:1, bound_ctrl off)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

__device__ __forceinline__ float lse2_b2(float a, float b) {
  const float m = fmaxf(a, b);
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m));
}
__device__ __forceinline__ float lse3_b2(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) +
                                   __builtin_amdgcn_exp2f(c - m));
}

__device__ __forceinline__ float load_score(const float* p) {
  const float v = *p * kLog2e;
  return (v > kNegBig) ? v : kNegBig;  // NaN and -inf both become the sentinel (NaN policy)
}

// ws layout: [b][dir][t][pos] float2 with pos < P (= max_len + 1)
__global__ void __launch_bounds__(64)
    ctc_chain_kernel(const float* __restrict__ x, int T, int C, const int32_t* __restrict__ targets,
                     const int64_t* __restrict__ offsets, int P, int blank, float2* __restrict__ ws,
                     float* __restrict__ nll) {
  const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x;
  const int64_t o0 = offsets[b];
  const int L = (int)(offsets[b + 1] - o0);
  // this lane's label (reversed target for the beta sweep) and skip flag
  int y = -1, yprev = -1;
  if (lane < L) y = targets[o0 + (dir == 0 ? lane : L - 1 - lane)];
  if (lane >= 1 && lane - 1 < L) yprev = targets[o0 + (dir == 0 ? lane - 1 : L - lane)];
  const bool has_label = lane < L, has_blank = lane <= L;
  const bool skip = has_label && lane >= 1 && y != yprev;
  const float* xb_ptr = x + (int64_t)b * T * C;  // row base; + t*C + column
  const int col = has_label ? y : blank;
  float2* out = ws + ((int64_t)(b * 2 + dir) * T) * P;

  float ab = (lane == 0) ? 0.f : kNegBig;  // virtual slot "before the first frame"
  float al = kNegBig;
  float xl_r[kCtcPrefetch], xb_r[kCtcPrefetch];
#pragma unroll
  for (int j = 0; j < kCtcPrefetch; ++j) {
    const int t = dir == 0 ? j : T - 1 - j;
    if (j < T) {
      xl_r[j] = load_score(xb_ptr + (int64_t)t * C + col);
      xb_r[j] = load_score(xb_ptr + (int64_t)t * C + blank);
    }
  }
  for (int s0 = 0; s0 < T; s0 += kCtcPrefetch) {
#pragma unroll
    for (int j = 0; j < kCtcPrefetch; ++j) {
      const int step = s0 + j;
      if (step < T) {
        const int t = dir == 0 ? step : T - 1 - step;
        const float xl = has_label ? xl_r[j] : kNegBig;
        const float xb = has_blank ? xb_r[j] : kNegBig;
        const int sn = step + kCtcPrefetch;
        if (sn < T) {
          const int tn = dir == 0 ? sn : T - 1 - sn;
          xl_r[j] = load_score(xb_ptr + (int64_t)tn * C + col);
          xb_r[j] = load_score(xb_ptr + (int64_t)tn * C + blank);
        }
        const float pal = wave_shr1(al, kNegBig);
        const float nb = lse2_b2(ab, pal);
        const float nl = lse3_b2(al, ab, skip ? pal : kNegBig);
        ab = fmaxf(nb + xb, kNegBig);
        al = fmaxf(nl + xl, kNegBig);
        if (lane < P) out[(int64_t)t * P + lane] = dir == 0 ? make_float2(ab, al) : make_float2(nb, nl);
      }
    }
  }
  if (dir == 0) {
    // logZ = LSE(alpha_{T-1}[2L], alpha_{T-1}[2L-1]) = LSE(ab[L], al[L-1])   (ctc.py:21 accept states)
    const float a_last = __shfl(ab, L, 64);
    const float l_last = L > 0 ? __shfl(al, L - 1, 64) : kNegBig;
    if (lane == 0) {
      const float z2 = lse2_b2(a_last, l_last);
      nll[b] = (z2 > 0.5f * kNegBig) ? -z2 * kLn2 : __builtin_inff();
    }
  }
}

__global__ void __launch_bounds__(256)
    ctc_grad_kernel(int T, int C, const int32_t* __restrict__ targets, const int64_t* __restrict__ offsets, int P,
                    int blank, const float2* __restrict__ ws, const float* __restrict__ nll,
                    const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* row = (float*)smem + (size_t)wave * C;
  const int64_t o0 = offsets[b];
  const int L = (int)(offsets[b + 1] - o0);
  const int y = lane < L ? targets[o0 + lane] : blank;
  const float loss = nll[b];
  const bool dead = !(loss < __builtin_inff());
  const float z2 = -loss * kLog2e;
  const float cf = (coef ? coef[b] : 1.f) * (gout ? gout[0] : 1.f);
  const float2* al = ws + ((int64_t)(b * 2 + 0) * T) * P;
  const float2* be = ws + ((int64_t)(b * 2 + 1) * T) * P;
  for (int c = lane; c < C; c += 64) row[c] = 0.f;
  __syncthreads();
  for (int tb = blockIdx.x * 4; tb < T; tb += gridDim.x * 4) {  // uniform trip count: barriers inside
    const int t = tb + wave;
    const bool live = t < T && !dead;
    float gb = 0.f, gl = 0.f;
    if (live && lane <= L) {
      const float2 a = al[(int64_t)t * P + lane];
      // mirrored beta: blank state 2i <-> reversed position L-i; label of position i <-> L-1-i
      const float bb = be[(int64_t)t * P + (L - lane)].x;
      gb = __builtin_amdgcn_exp2f(a.x + bb - z2);
      if (lane < L) {
        const float bl = be[(int64_t)t * P + (L - 1 - lane)].y;
        gl = __builtin_amdgcn_exp2f(a.y + bl - z2);
      }
    }
    gb = wave_sum(gb);
    if (lane == 0 && gb != 0.f) atomicAdd(&row[blank], gb * cf);
    if (lane < L && gl != 0.f) atomicAdd(&row[y], gl * cf);
    __syncthreads();
    if (t < T) {
      float* dst = dx + ((int64_t)b * T + t) * C;
      for (int c = lane; c < C; c += 64) {
        dst[c] = row[c];
        row[c] = 0.f;
      }
    }
    __syncthreads();
  }
}

}  // namespace wfl

using namespace wfl;

extern "C" {

static int ctc_check(int B, int T, int C, int max_len, int blank, const char* who) {
  if (B <= 0 || T <= 0 || C <= 0 || blank < 0 || blank >= C || max_len < 0) {
    set_error("%s: bad arguments (B=%d T=%d C=%d blank=%d max_len=%d)", who, B, T, C, blank, max_len);
    return WFL_ERR_INVALID;
  }
  if (max_len + 1 > 64) {
    set_error("%s: target length %d needs more than one 64-lane wavefront (use the lattice engine)", who, max_len);
    return WFL_ERR_UNSUPPORTED;
  }
  return WFL_OK;
}

int wfl_ctc_workspace(int B, int T, int C, int max_len, int64_t* ws_elems) {
  if (!ws_elems || B <= 0 || T <= 0 || max_len < 0) {
    set_error("ctc_workspace: bad arguments");
    return WFL_ERR_INVALID;
  }
  (void)C;
  *ws_elems = (int64_t)B * 2 * T * (max_len + 1) * 2;
  return WFL_OK;
}

int wfl_ctc_forward(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets, int max_len,
                    int blank, float* ws, float* nll, void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_forward")) return rc;
  if (!x || !targets || !offsets || !ws || !nll) {
    set_error("ctc_forward: null buffer");
    return WFL_ERR_INVALID;
  }
  hipLaunchKernelGGL(ctc_chain_kernel, dim3((unsigned)B, 2u), dim3(64), 0, (hipStream_t)stream, x, T, C, targets,
                     offsets, max_len + 1, blank, (float2*)ws, nll);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

int wfl_ctc_grad(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets, int max_len,
                 int blank, const float* ws, const float* nll, const float* coef, const float* gout, float* dx,
                 void* stream) {
  if (int rc = ctc_check(B, T, C, max_len, blank, "ctc_grad")) return rc;
  if (!targets || !offsets || !ws || !nll || !dx) {
    set_error("ctc_grad: null buffer");
    return WFL_ERR_INVALID;
  }
  (void)x;
  const int blocks_t = std::max(1, std::min((T + 3) / 4, (4096 + B - 1) / B));
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)blocks_t, (unsigned)B), dim3(256), (size_t)4 * C * 4,
                     (hipStream_t)stream, T, C, targets, offsets, max_len + 1, blank, (const float2*)ws, nll, coef,
                     gout, dx);
  WFL_LAUNCH_CHECK();
  return WFL_OK;
}

}  // extern "C"
