// Autograd nodes of the criteria's hot operator paths, in C++ (module gtn_applications_amd._wfl_torch).
//
// The kernels of the CTC step take ~60 us at the reference's benchmark shape; a Python torch.autograd.Function
// around them costs more than that in interpreter time alone (apply() bookkeeping, the engine's call back into
// Python for backward, tensor wrapping of the saved state).  This file is the same operator -- counterpart of
// CTCLossFunction.forward / backward, /root/reference/criterions/ctc.py:31-93 -- as a torch::autograd::Function that
// calls the C ABI of libwfl.so (include/wfl.h) directly.  Host-side plumbing only: no arithmetic happens here.
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include "../../include/wfl.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void* current_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check(int rc, const char* what) { TORCH_CHECK(rc == WFL_OK, what, ": ", wfl_last_error()); }

// Loss and gradient of a CTC batch in ONE pipelined launch (wfl_ctc_forward_backward); backward only applies the
// upstream scalar to the gradient computed here (like torch's own CTC the gradient is produced eagerly).
//   staged: the uint8 device buffer of engine.CtcTargets (offsets | flat labels | per-utterance factors), addressed
//   by the byte offsets that follow; ws / nll: the per-stream scratch of engine.ctc_workspace; lse: optional row
//   log-sum-exps of x (fused log_softmax, ctc.py:107).
struct CtcStep : public torch::autograd::Function<CtcStep> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& staged, int64_t off_offsets,
                            int64_t off_flat, int64_t off_scale, int64_t off_coef, int64_t max_len, int64_t blank,
                            const at::Tensor& ws, const at::Tensor& nll, const c10::optional<at::Tensor>& lse) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 3,
                "ctc_step: x must be a contiguous float32 [B,T,C] device tensor");
    const auto B = x.size(0), T = x.size(1), C = x.size(2);
    at::Tensor dx = at::empty_like(x);
    at::Tensor loss = at::empty({}, x.options());
    const char* base = static_cast<const char*>(staged.data_ptr());
    const float* lse_p = lse.has_value() && lse->defined() ? lse->data_ptr<float>() : nullptr;
    check(wfl_ctc_forward_backward(x.data_ptr<float>(), (int)B, (int)T, (int)C,
                                   reinterpret_cast<const int32_t*>(base + off_flat),
                                   reinterpret_cast<const int64_t*>(base + off_offsets), (int)max_len, (int)blank,
                                   ws.data_ptr<float>(), nll.data_ptr<float>(),
                                   reinterpret_cast<const float*>(base + off_coef), nullptr, dx.data_ptr<float>(),
                                   reinterpret_cast<const float*>(base + off_scale), loss.data_ptr<float>(), lse_p,
                                   current_stream(x)),
          "ctc_step");
    ctx->saved_data["x"] = x.detach();
    ctx->saved_data["staged"] = staged;
    ctx->saved_data["ws"] = ws;
    ctx->saved_data["nll"] = nll;
    ctx->saved_data["dx"] = dx;
    if (lse_p) ctx->saved_data["lse"] = *lse;
    ctx->saved_data["ints"] = std::vector<int64_t>{off_offsets, off_flat, off_coef, max_len, blank};
    return loss;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::Tensor x = ctx->saved_data["x"].toTensor();
    at::Tensor g = grads[0].detach().reshape({1});
    if (!g.is_cuda() || g.scalar_type() != at::kFloat) g = g.to(x.device(), at::kFloat);
    auto it = ctx->saved_data.find("dx");
    at::Tensor dx;
    if (it != ctx->saved_data.end() && it->second.isTensor() && it->second.toTensor().defined()) {
      dx = it->second.toTensor();
      ctx->saved_data.erase(it);  // handed out (and scaled in place) once
      check(wfl_scale(dx.data_ptr<float>(), dx.numel(), g.data_ptr<float>(), current_stream(x)), "ctc_step backward");
    } else {
      // a second backward through a retained graph: the same launch again into a fresh buffer, the upstream scalar
      // applied by the kernel (with the same row log-sum-exps when the log_softmax is fused)
      const auto v = ctx->saved_data["ints"].toIntVector();
      at::Tensor staged = ctx->saved_data["staged"].toTensor(), ws = ctx->saved_data["ws"].toTensor(),
                 nll = ctx->saved_data["nll"].toTensor();
      const char* base = static_cast<const char*>(staged.data_ptr());
      auto l = ctx->saved_data.find("lse");
      const float* lse_p = l != ctx->saved_data.end() ? l->second.toTensor().data_ptr<float>() : nullptr;
      dx = at::empty_like(x);
      check(wfl_ctc_forward_backward(x.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)x.size(2),
                                     reinterpret_cast<const int32_t*>(base + v[1]),
                                     reinterpret_cast<const int64_t*>(base + v[0]), (int)v[3], (int)v[4],
                                     ws.data_ptr<float>(), nll.data_ptr<float>(),
                                     reinterpret_cast<const float*>(base + v[2]), g.data_ptr<float>(),
                                     dx.data_ptr<float>(), nullptr, nullptr, lse_p, current_stream(x)),
            "ctc_step backward");
    }
    return {dx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

at::Tensor ctc_step(const at::Tensor& x, const at::Tensor& staged, int64_t off_offsets, int64_t off_flat,
                    int64_t off_scale, int64_t off_coef, int64_t max_len, int64_t blank, const at::Tensor& ws,
                    const at::Tensor& nll, const c10::optional<at::Tensor>& lse) {
  return CtcStep::apply(x, staged, off_offsets, off_flat, off_scale, off_coef, max_len, blank, ws, nll, lse);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("ctc_step", &ctc_step, "CTC loss + eager gradient in one pipelined launch (C++ autograd node)");
}
