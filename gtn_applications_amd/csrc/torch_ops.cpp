// Autograd nodes of the criteria's hot operator paths, in C++ (module gtn_applications_amd._wfl_torch).
//
// The kernels of the CTC step take ~60 us at the reference's benchmark shape; a Python torch.autograd.Function
// around them costs more than that in interpreter time alone (apply() bookkeeping, the engine's call back into
// Python for backward, tensor wrapping of the saved state).  This file is the same operator -- counterpart of
// CTCLossFunction.forward / backward, /root/reference/criterions/ctc.py:31-93 -- as a torch::autograd::Function that
// calls the C ABI of libwfl.so (include/wfl.h) directly.  Host-side plumbing only: no arithmetic happens here.
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/functions/accumulate_grad.h>
#include <torch/csrc/autograd/python_variable.h>
#include <torch/extension.h>

#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/wfl.h"

namespace {
// ordering-only events of this device: no timestamp, a device-scope release when recorded (csrc/device_common.h has the
// reasoning and the WFL_ORDER_EVENTS switch; kept in step by hand: this file does not include the kernels' headers)
unsigned order_event_flags() {
  static const unsigned flags = [] {
    const char* v = getenv("WFL_ORDER_EVENTS");
    if (v && !strcmp(v, "nofence")) return (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
    if (v && !strcmp(v, "default")) return (unsigned)hipEventDisableTiming;
    return (unsigned)(hipEventDisableTiming | hipEventReleaseToDevice);
  }();
  return flags;
}

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
                            const at::Tensor& ws, const at::Tensor& nll, const c10::optional<at::Tensor>& lse,
                            int64_t n_labels, int64_t host_state) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 3,
                "ctc_step: x must be a contiguous float32 [B,T,C] device tensor");
    const auto B = x.size(0), T = x.size(1), C = x.size(2);
    at::Tensor dx = at::empty_like(x);
    at::Tensor loss = at::empty({}, x.options());
    const char* base = static_cast<const char*>(staged.data_ptr());
    const float* lse_p = lse.has_value() && lse->defined() ? lse->data_ptr<float>() : nullptr;
    // (what the step remembers between calls is the caller's: wfl_ctc_call in include/wfl.h)
    const wfl_ctc_call call{n_labels, reinterpret_cast<int32_t*>(host_state)};
    check(wfl_ctc_forward_backward_call(x.data_ptr<float>(), (int)B, (int)T, (int)C,
                                        reinterpret_cast<const int32_t*>(base + off_flat),
                                        reinterpret_cast<const int64_t*>(base + off_offsets), (int)max_len, (int)blank,
                                        ws.data_ptr<float>(), nll.data_ptr<float>(),
                                        reinterpret_cast<const float*>(base + off_coef), nullptr, dx.data_ptr<float>(),
                                        reinterpret_cast<const float*>(base + off_scale), loss.data_ptr<float>(), lse_p,
                                        &call, current_stream(x)),
          "ctc_step");
    ctx->saved_data["x"] = x.detach();
    ctx->saved_data["staged"] = staged;
    ctx->saved_data["ws"] = ws;
    ctx->saved_data["nll"] = nll;
    ctx->saved_data["dx"] = dx;
    if (lse_p) ctx->saved_data["lse"] = *lse;
    ctx->saved_data["ints"] = std::vector<int64_t>{off_offsets, off_flat, off_coef, max_len, blank, n_labels};
    return loss;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    TORCH_CHECK(ctx->saved_data.find("freed") == ctx->saved_data.end(),
                "Trying to backward through the graph a second time (or directly access saved tensors after they have "
                "already been freed). Saved intermediate values of the graph are freed when you call .backward() or "
                "autograd.grad(). Specify retain_graph=True if you need to backward through the graph a second time.");
    at::Tensor x = ctx->saved_data["x"].toTensor();
    if (!grads[0].defined())  // (the loss did not take part in what is being differentiated)
      return {at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
              at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
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
      const wfl_ctc_call call{v[5], nullptr};  // (a recomputation: not a step whose outcome is to be remembered)
      check(wfl_ctc_forward_backward_call(x.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)x.size(2),
                                          reinterpret_cast<const int32_t*>(base + v[1]),
                                          reinterpret_cast<const int64_t*>(base + v[0]), (int)v[3], (int)v[4],
                                          ws.data_ptr<float>(), nll.data_ptr<float>(),
                                          reinterpret_cast<const float*>(base + v[2]), g.data_ptr<float>(),
                                          dx.data_ptr<float>(), nullptr, nullptr, lse_p, &call, current_stream(x)),
            "ctc_step backward");
    }
    return {dx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

// ------------------------------------------------------------------------------------------------------------
// Targets of a batch, staged and uploaded without passing through Python objects: what engine.CtcTargets does
// (flatten list-of-int-lists -> [int64 offsets | int32 labels | six per-utterance factor arrays] in a pinned ring ->
// one wfl_upload -> small content-keyed cache), for the one operator whose kernels are shorter than that Python code.
// ------------------------------------------------------------------------------------------------------------
struct StagedTargets {
  at::Tensor dev_buf;  // uint8, device
  std::string key_bytes;  // [offsets | labels] as staged (confirms a cache hit byte for byte)
  int64_t B = 0, n = 0, max_len = 0, off_flat = 0, off_fac = 0;
  long label_min = 0, label_max = -1;
  hipStream_t up_stream = nullptr;  // the stream the upload was queued on ...
  hipEvent_t up_event = nullptr;    // ... and this batch's OWN event behind it (the staging slot's event is re-recorded
                                    // by later uploads, possibly on another stream)
  int slot = 0;
  StagedTargets() = default;
  StagedTargets(const StagedTargets&) = delete;
  StagedTargets& operator=(const StagedTargets&) = delete;
  ~StagedTargets() {
    if (up_event) free_events(dev_index).push_back(up_event);  // (creating an event costs more than recording one: kept for the next batch)
  }
  int dev_index = 0;  // (an event belongs to the device it was created on)
  static std::vector<hipEvent_t>& free_events(int dev) {
    // (never destroyed: the target caches are torn down by static destructors at exit, in no particular order, and
    // their entries come through here)
    static auto* pools = new std::unordered_map<int, std::vector<hipEvent_t>>();
    return (*pools)[dev];
  }
};

struct PinnedRing {  // reusable pinned staging buffers; a slot is reused after the upload that read it has completed
  static constexpr int kSlots = 8;
  at::Tensor buf[kSlots];
  hipEvent_t ev[kSlots] = {};
  int i = 0;
  uint8_t* next(int64_t need, int& slot) {
    slot = i = (i + 1) % kSlots;
    if (ev[slot]) (void)hipEventSynchronize(ev[slot]);
    if (!buf[slot].defined() || buf[slot].numel() < need) {
      int64_t cap = 1 << 18;
      while (cap < need) cap <<= 1;
      // every slot at once: a pinned allocation costs hundreds of microseconds, and allocating slot by slot would
      // spread eight of them over the first eight steps of a run instead of paying them in the first one
      for (int k = 0; k < kSlots; ++k)
        if (!buf[k].defined() || buf[k].numel() < cap) {
          if (ev[k]) (void)hipEventSynchronize(ev[k]);
          buf[k] = at::empty({cap}, at::TensorOptions().dtype(at::kByte).pinned_memory(true));
        }
    }
    return buf[slot].data_ptr<uint8_t>();
  }
  void uploaded(int slot, hipStream_t stream) {
    if (!ev[slot]) (void)hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming);
    (void)hipEventRecord(ev[slot], stream);
  }
};

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33, k *= 0xff51afd7ed558ccdULL, k ^= k >> 33, k *= 0xc4ceb9fe1a85ec53ULL, k ^= k >> 33;
  return k;
}
std::pair<uint64_t, uint64_t> hash128(const uint8_t* p, int64_t n) {  // MurmurHash3 x64_128 mixing steps
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = 0x9e3779b97f4a7c15ULL, h2 = 0xd1b54a32d192ed03ULL;
  const int64_t nb = n / 16;
  for (int64_t i = 0; i < nb; ++i) {
    uint64_t k1, k2;
    memcpy(&k1, p + 16 * i, 8), memcpy(&k2, p + 16 * i + 8, 8);
    k1 *= c1, k1 = rotl64(k1, 31), k1 *= c2, h1 ^= k1, h1 = rotl64(h1, 27), h1 += h2, h1 = h1 * 5 + 0x52dce729;
    k2 *= c2, k2 = rotl64(k2, 33), k2 *= c1, h2 ^= k2, h2 = rotl64(h2, 31), h2 += h1, h2 = h2 * 5 + 0x38495ab5;
  }
  uint64_t t1 = 0, t2 = 0;
  const int64_t rem = n - 16 * nb;
  if (rem > 8) memcpy(&t2, p + 16 * nb + 8, (size_t)(rem - 8));
  if (rem > 0) memcpy(&t1, p + 16 * nb, (size_t)(rem > 8 ? 8 : rem));
  t2 *= c2, t2 = rotl64(t2, 33), t2 *= c1, h2 ^= t2;
  t1 *= c1, t1 = rotl64(t1, 31), t1 *= c2, h1 ^= t1;
  h1 ^= (uint64_t)n, h2 ^= (uint64_t)n, h1 += h2, h2 += h1, h1 = fmix64(h1), h2 = fmix64(h2), h1 += h2, h2 += h1;
  return {h1, h2};
}

// The batch a device saw last, recognised by OBJECT IDENTITY: the reference's benchmarks (ctc_benchmark.py:26-31) hand the
// same list of int lists to every iteration, and flattening + hashing its 5 632 labels is 10 of the ~20 us the forward
// spends on the host -- on a step whose kernels take 45.  Python ints are immutable, so "the same int OBJECTS in the same
// places" means the same labels: one pointer comparison per label against the batch remembered here.  The entry holds a
// reference to every label object (an address it compares against cannot be recycled for another int) and is only
// filled when the content-keyed cache below HITS -- a run whose targets are new every step never pays for it.
struct LastBatch {
  std::vector<PyObject*> elems;  // owned references, row after row
  std::vector<Py_ssize_t> lens;
  std::shared_ptr<StagedTargets> st;
  void clear() {
    for (PyObject* o : elems) Py_DECREF(o);
    elems.clear(), lens.clear(), st.reset();
  }
  // the rows of `t` (lists / tuples only) hold exactly the remembered objects?
  bool matches(PyObject** rows, Py_ssize_t B) const {
    if (!st || (Py_ssize_t)lens.size() != B) return false;
    size_t k = 0;
    for (Py_ssize_t b = 0; b < B; ++b) {
      PyObject* r = rows[b];
      if (!PyList_Check(r) && !PyTuple_Check(r)) return false;
      const Py_ssize_t n = PySequence_Fast_GET_SIZE(r);
      if (n != lens[b]) return false;
      if (n && memcmp(PySequence_Fast_ITEMS(r), elems.data() + k, (size_t)n * sizeof(PyObject*)) != 0) return false;
      k += (size_t)n;
    }
    return true;
  }
  void remember(PyObject** rows, Py_ssize_t B, const std::shared_ptr<StagedTargets>& staged) {
    clear();
    for (Py_ssize_t b = 0; b < B; ++b) {
      PyObject* r = rows[b];
      if (!PyList_Check(r) && !PyTuple_Check(r)) {  // (tensor rows: their storage is mutable, identity proves nothing)
        clear();
        return;
      }
      const Py_ssize_t n = PySequence_Fast_GET_SIZE(r);
      PyObject** it = PySequence_Fast_ITEMS(r);
      lens.push_back(n);
      for (Py_ssize_t i = 0; i < n; ++i) {
        if (!PyLong_CheckExact(it[i])) {  // (an int subclass could answer differently next time)
          clear();
          return;
        }
        Py_INCREF(it[i]);
        elems.push_back(it[i]);
      }
    }
    st = staged;
  }
};

struct TargetCache {  // per device: ring + LRU of the last 64 distinct batches
  PinnedRing ring;
  LastBatch last;
  using Key = std::tuple<uint64_t, uint64_t, int64_t>;
  std::list<std::pair<Key, std::shared_ptr<StagedTargets>>> lru;
  std::map<Key, decltype(lru)::iterator> index;
};
std::unordered_map<int, TargetCache> g_targets;

// -> staged targets, or nullptr if `targets` is not a list / tuple of lists / tuples of ints (the caller falls back)
std::shared_ptr<StagedTargets> stage_targets(const py::handle& targets, const at::Device& dev) {
  PyObject* t = targets.ptr();
  if (!PyList_Check(t) && !PyTuple_Check(t)) return nullptr;
  const Py_ssize_t B = PySequence_Fast_GET_SIZE(t);
  PyObject** rows = PySequence_Fast_ITEMS(t);
  // rows: lists / tuples of ints (the benchmarks, `[t.tolist() for t in targets]`) or 1-D CPU int tensors (train.py)
  auto tensor_row = [](PyObject* r) -> const at::Tensor* {
    if (!THPVariable_Check(r)) return nullptr;
    const at::Tensor& v = THPVariable_Unpack(r);
    const bool ok = v.dim() == 1 && v.device().is_cpu() && (v.scalar_type() == at::kLong || v.scalar_type() == at::kInt);
    return ok ? &v : nullptr;
  };
  int64_t total = 0, max_len = 0;
  for (Py_ssize_t b = 0; b < B; ++b) {
    int64_t n;
    if (PyList_Check(rows[b]) || PyTuple_Check(rows[b]))
      n = PySequence_Fast_GET_SIZE(rows[b]);
    else if (const at::Tensor* v = tensor_row(rows[b]))
      n = v->numel();
    else
      return nullptr;
    total += n, max_len = std::max<int64_t>(max_len, n);
  }
  TargetCache& tc = g_targets[dev.index()];
  auto on_this_stream = [&](const std::shared_ptr<StagedTargets>& e) {
    const hipStream_t now = c10::hip::getCurrentHIPStream(dev.index()).stream();
    // reused on another stream than the one that uploaded it: order this stream behind the upload
    if (now != e->up_stream && e->up_event) (void)hipStreamWaitEvent(now, e->up_event, 0);
    return e;
  };
  if (tc.last.matches(rows, B)) return on_this_stream(tc.last.st);  // the same label objects as last time: nothing to stage
  const int64_t off_flat = 8 * (B + 1), off_fac = (off_flat + 4 * std::max<int64_t>(total, 1) + 7) & ~(int64_t)7;
  const int64_t nbytes = off_fac + 4 * B * 6;
  int slot;
  uint8_t* base = tc.ring.next(nbytes + 16, slot);
  int64_t* off = reinterpret_cast<int64_t*>(base);
  int32_t* flat = reinterpret_cast<int32_t*>(base + off_flat);
  long lo = 0, hi = -1;
  bool first = true;
  int64_t k = 0;
  auto put = [&](long v) {
    if (v > INT32_MAX || v < INT32_MIN) throw py::value_error("target label does not fit int32");
    if (first || v < lo) lo = v;
    if (first || v > hi) hi = v;
    first = false;
    flat[k++] = (int32_t)v;
  };
  for (Py_ssize_t b = 0; b < B; ++b) {
    off[b] = k;
    if (const at::Tensor* v = tensor_row(rows[b])) {
      const int64_t n = v->numel(), st = n ? v->stride(0) : 1;
      if (v->scalar_type() == at::kLong) {
        const int64_t* p = v->data_ptr<int64_t>();
        for (int64_t i = 0; i < n; ++i) put((long)p[i * st]);
      } else {
        const int32_t* p = v->data_ptr<int32_t>();
        for (int64_t i = 0; i < n; ++i) put((long)p[i * st]);
      }
      continue;
    }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(rows[b]);
    PyObject** it = PySequence_Fast_ITEMS(rows[b]);
    for (Py_ssize_t i = 0; i < n; ++i) {
      long v;
      PyObject* o = it[i];
#if PY_VERSION_HEX < 0x030C0000
      // (exact ints of one 30-bit digit -- every label there is -- without the call: the long layout of CPython <= 3.11;
      // from 3.12 on the digits live under long_value and Py_SIZE is not defined for ints: PyLong_AsLong below)
      if (PyLong_CheckExact(o) && (Py_SIZE(o) == 1 || Py_SIZE(o) == 0)) {
        v = Py_SIZE(o) ? (long)reinterpret_cast<PyLongObject*>(o)->ob_digit[0] : 0;
        put(v);
        continue;
      }
#endif
      v = PyLong_AsLong(o);
      if (v == -1 && PyErr_Occurred()) {
        PyErr_Clear();
        tc.ring.i = (tc.ring.i + PinnedRing::kSlots - 1) % PinnedRing::kSlots;  // slot not used
        return nullptr;  // not ints: the Python path normalises (numpy ints, ranges, ...)
      }
      put(v);
    }
  }
  off[B] = k;
  const int64_t nkey = off_flat + 4 * total;
  const auto h = hash128(base, nkey);
  const TargetCache::Key key{h.first, h.second, nkey};
  auto hit = tc.index.find(key);
  if (hit != tc.index.end() && (int64_t)hit->second->second->key_bytes.size() == nkey &&
      memcmp(hit->second->second->key_bytes.data(), base, (size_t)nkey) == 0) {
    tc.lru.splice(tc.lru.begin(), tc.lru, hit->second);
    tc.ring.i = (tc.ring.i + PinnedRing::kSlots - 1) % PinnedRing::kSlots;  // nothing was uploaded from the slot
    auto& e = hit->second->second;
    tc.last.remember(rows, B, e);  // (seen before, by content: next time its objects are recognised without the staging)
    return on_this_stream(e);
  }
  // per-utterance factors (engine._FACTORS order): scale_none, scale_mean, then both times +1/B and -1/B
  float* fac = reinterpret_cast<float*>(base + off_fac);
  const float inv_b = 1.0f / (float)(B > 0 ? B : 1);
  for (Py_ssize_t b = 0; b < B; ++b) {
    const float ln = (float)(off[b + 1] - off[b]);
    const float mean = ln > 0.f ? 1.0f / ln : 1.0f;
    fac[b] = 1.0f, fac[B + b] = mean, fac[2 * B + b] = inv_b, fac[3 * B + b] = mean * inv_b;
    fac[4 * B + b] = -inv_b, fac[5 * B + b] = mean * -inv_b;
  }
  auto st = std::make_shared<StagedTargets>();
  st->B = B, st->n = total, st->max_len = max_len, st->off_flat = off_flat, st->off_fac = off_fac;
  st->label_min = lo, st->label_max = hi;
  st->key_bytes.assign(reinterpret_cast<const char*>(base), (size_t)nkey);
  st->dev_buf = at::empty({nbytes}, at::TensorOptions().dtype(at::kByte).device(dev));
  const hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
  check(wfl_upload(st->dev_buf.data_ptr(), base, nbytes, (void*)stream), "stage_targets");
  tc.ring.uploaded(slot, stream);
  st->up_stream = stream, st->slot = slot;
  {
    st->dev_index = dev.index();
    auto& pool = StagedTargets::free_events(st->dev_index);
    if (!pool.empty())
      st->up_event = pool.back(), pool.pop_back();
    else if (hipEventCreateWithFlags(&st->up_event, hipEventDisableTiming) != hipSuccess)
      st->up_event = nullptr;
    if (st->up_event) (void)hipEventRecord(st->up_event, stream);
  }
  if (hit != tc.index.end()) {  // (same hash, different bytes: replace)
    tc.lru.erase(hit->second);
    tc.index.erase(hit);
  }
  tc.lru.emplace_front(key, st);
  tc.index[key] = tc.lru.begin();
  if (tc.lru.size() > 64) {
    tc.index.erase(tc.lru.back().first);
    tc.lru.pop_back();
  }
  return st;
}

struct WsKey {
  int dev;
  void* stream;
  int64_t B, T, C, L;
  bool operator<(const WsKey& o) const { return std::tie(dev, stream, B, T, C, L) < std::tie(o.dev, o.stream, o.B, o.T, o.C, o.L); }
};
std::map<WsKey, std::pair<at::Tensor, at::Tensor>> g_ws;  // (engine.ctc_workspace: scratch + nll per stream and shape)

// The step's memory between calls (wfl_ctc_call::host_state: two pinned int32 per stream and shape, written by the
// repair launch without anybody waiting).  Handed out from pinned pages that are never freed: a launch still in flight
// may write its word whatever happens to the workspace cache above.
struct HostStatePool {
  static constexpr size_t kWords = 1024;
  std::mutex mu;
  std::map<WsKey, int32_t*> table;
  std::vector<int32_t*> pages;
  size_t used = kWords;
};
HostStatePool& host_state_pool() {
  static auto* p = new HostStatePool();  // (never destroyed: see above)
  return *p;
}
int32_t* ctc_host_state(const WsKey& k) {
  HostStatePool& P = host_state_pool();
  std::lock_guard<std::mutex> lock(P.mu);
  auto it = P.table.find(k);
  if (it != P.table.end()) return it->second;
  // (a stream that is being captured into a graph: no allocation, no memory -- the captured step replays one choice)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)k.stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  if (P.used + 2 > HostStatePool::kWords) {
    void* p = nullptr;
    if (hipHostMalloc(&p, HostStatePool::kWords * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return nullptr;
    memset(p, 0, HostStatePool::kWords * sizeof(int32_t));
    P.pages.push_back(static_cast<int32_t*>(p));
    P.used = 0;
  }
  int32_t* w = P.pages.back() + P.used;
  P.used += 2;
  P.table[k] = w;
  return w;
}
// forget what the steps remembered: the next call of every shape starts with the lane-exponent step (tests that run
// unrelated data through one shape)
void ctc_reset_host_state() {
  HostStatePool& P = host_state_pool();
  std::lock_guard<std::mutex> lock(P.mu);
  for (int32_t* page : P.pages) memset(page, 0, HostStatePool::kWords * sizeof(int32_t));
}

// CTCLoss(log_probs, targets, blank, reduction) for the hot case, everything between the Python call and the launch
// in this one function.  Returns None when the case is not the hot one (the caller takes the Python path).
py::object ctc_loss_staged(const at::Tensor& x, const std::shared_ptr<StagedTargets>& st, int64_t blank, bool mean,
                           bool fused_lse, int64_t lim_len, int64_t lim_c, int64_t lim_c_long) {
  const auto B = x.size(0), T = x.size(1), C = x.size(2);
  if (!st) return py::none();
  if (!(st->max_len <= lim_len && C <= (st->max_len <= 63 ? lim_c : lim_c_long))) return py::none();
  if (st->B != B) throw py::value_error("got " + std::to_string(st->B) + " targets for a batch of " + std::to_string(B));
  if (st->label_min < 0 || st->label_max >= C) {
    const long bad = st->label_min < 0 ? st->label_min : st->label_max;
    throw py::value_error("CTCLoss: target label " + std::to_string(bad) + " is outside [0, " + std::to_string(C) +
                          ") (emissions have " + std::to_string(C) + " classes)");
  }
  if (blank < 0 || blank >= C)
    throw py::value_error("CTCLoss: blank index " + std::to_string(blank) + " is outside [0, " + std::to_string(C) + ")");
  void* stream = current_stream(x);
  const WsKey wk{x.device().index(), stream, B, T, C, st->max_len};
  auto w = g_ws.find(wk);
  if (w == g_ws.end()) {
    int64_t n = 0;
    check(wfl_ctc_workspace((int)B, (int)T, (int)C, (int)st->max_len, &n), "ctc_workspace");
    if (g_ws.size() >= 16) g_ws.clear();
    w = g_ws.emplace(wk, std::make_pair(at::empty({n}, x.options()), at::empty({B}, x.options()))).first;
  }
  c10::optional<at::Tensor> lse;
  if (fused_lse) {
    lse = at::empty({B, T}, x.options());
    check(wfl_row_lse(x.data_ptr<float>(), B * T, (int)C, lse->data_ptr<float>(), stream), "row_lse");
  }
  const int64_t fac = st->off_fac + 4 * B * (mean ? 1 : 0);  // scale_<reduction>; cneg_<reduction> is 4 arrays on
  return py::cast(CtcStep::apply(x, st->dev_buf, 0, st->off_flat, fac, fac + 16 * B, st->max_len, blank, w->second.first,
                                 w->second.second, lse, st->n, reinterpret_cast<int64_t>(ctc_host_state(wk))));
}

// CTCLoss(log_probs, targets, blank, reduction) for the hot case, everything between the Python call and the launch
// in this one function.  Returns None when the case is not the hot one (the caller takes the Python path).
py::object ctc_loss_lists(const at::Tensor& x, const py::handle& targets, int64_t blank, bool mean, bool fused_lse,
                          int64_t lim_len, int64_t lim_c, int64_t lim_c_long) {
  return ctc_loss_staged(x, stage_targets(targets, x.device()), blank, mean, fused_lse, lim_len, lim_c, lim_c_long);
}

std::shared_ptr<StagedTargets> stage_lists(const py::handle& targets, const at::Tensor& like) {
  return stage_targets(targets, like.device());
}

at::Tensor ctc_step(const at::Tensor& x, const at::Tensor& staged, int64_t off_offsets, int64_t off_flat,
                    int64_t off_scale, int64_t off_coef, int64_t max_len, int64_t blank, const at::Tensor& ws,
                    const at::Tensor& nll, const c10::optional<at::Tensor>& lse, int64_t n_labels, int64_t host_state) {
  return CtcStep::apply(x, staged, off_offsets, off_flat, off_scale, off_coef, max_len, blank, ws, nll, lse, n_labels,
                        host_state);
}

// `loss.backward()` for the loss a CtcStep node returned, without running THIS node on the autograd engine: the gradient
// was computed by the forward launch (for an upstream gradient of one, which is what a bare .backward() on a scalar
// means), so what is left is to hand dx on.
//   * Leaf emissions (ctc_benchmark.py's protocol on device-resident inputs): what AccumulateGrad would do -- dx becomes
//     the leaf's .grad (or is added to it).  No engine at all.
//   * Emissions that are some producer's OUTPUT (a model's, train.py:262-266; `.cuda()` of a host leaf,
//     ctc_benchmark.py:22): the engine is started AT the emissions' edge with dx as its root gradient
//     (x.backward(dx) without needing x): same graph below, same hooks on x and on everything under it -- but no
//     ones_like fill, no trip through this node and no scale launch.
// The engine's version of the first costs two thread hand-overs (device nodes run on the engine's device thread), a
// ones_like fill and the scale launch: 40-60 us of host time per step where the step's kernels take 45.  Returns false --
// the caller then takes the ordinary torch.Tensor.backward -- whenever anything is not exactly the plain case: the loss
// is not a fresh CtcStep output, hooks are registered on the loss or its node (or, for a leaf, on the emissions: the
// engine runs those), the gradient buffer has been handed out already.  Same results, same .grad semantics (first
// gradient: the buffer itself; later ones: added in place), same error on a second backward.
// Only what torch's public headers declare is read here: Node::{pre,post,tensor_pre,retains_grad}_hooks(),
// next_edge(), AccumulateGrad::variable / tensor_post_acc_grad_hooks(), Engine::execute (built_for_torch() below lets
// the Python side refuse this path under another torch than the one these headers came from).
bool ctc_fast_backward(const at::Tensor& loss) {
  auto fn = loss.grad_fn();
  auto* node = dynamic_cast<torch::autograd::CppNode<CtcStep>*>(fn.get());
  if (!node || loss.dim() != 0) return false;
  if (!fn->pre_hooks().empty() || !fn->post_hooks().empty() || !fn->tensor_pre_hooks().empty() ||
      !fn->retains_grad_hooks().empty())
    return false;
  const torch::autograd::Edge edge = fn->next_edge(0);
  if (!edge.function) return false;
  auto* acc = dynamic_cast<torch::autograd::AccumulateGrad*>(edge.function.get());
  at::Tensor x;
  if (acc) {
    if (!acc->pre_hooks().empty() || !acc->post_hooks().empty() || !acc->tensor_pre_hooks().empty() ||
        acc->tensor_post_acc_grad_hooks())
      return false;
    x = acc->variable;
    if (!x.defined() || !x.requires_grad()) return false;
  }
  auto& saved = node->ctx_.saved_data;
  if (saved.find("freed") != saved.end()) return false;  // (the ordinary path raises torch's error)
  auto it = saved.find("dx");
  if (it == saved.end() || !it->second.isTensor() || !it->second.toTensor().defined()) return false;
  at::Tensor dx = it->second.toTensor();
  saved.clear();  // what the engine's release of the graph does
  saved["freed"] = true;
  if (acc) {
    at::NoGradGuard no_grad;
    at::Tensor& grad = x.mutable_grad();
    if (!grad.defined())
      grad = std::move(dx);
    else
      grad.add_(dx);
    return true;
  }
  py::gil_scoped_release no_gil;  // (the engine takes the GIL itself where it runs Python nodes and hooks)
  torch::autograd::Engine::get_default_engine().execute({edge}, {std::move(dx)}, /*keep_graph=*/false,
                                                        /*create_graph=*/false, /*accumulate_grad=*/true, {});
  return true;
}

// the torch these nodes were compiled against (criterions/ctc.py compares it with the running one)
std::string built_for_torch() { return TORCH_VERSION; }


// ------------------------------------------------------------------------------------------------------------
// Every launch of an ASG step's forward in ONE native call: counterpart of ASGLossFunction.forward,
// /root/reference/criterions/asg.py:84-139 (the per-sample graph loop under gtn.parallel_for and the reduction), after
// the targets have been packed (engine.PackedLattice.asg_force_align, cached per batch).  The Python spelling of the same
// sequence (criterions/asg.py, kept as the fall-back when this module is missing and for phase timing) is ~15 tensor
// allocations, 8 ctypes calls, a stream context and three events: 180-205 us of interpreter time per step, against
// 450 us of kernels at the benchmark shape and ~100 us at a training batch of 8.  Host-side plumbing only.
//   numerator (force-aligned lattice, lattice engine) on `side_stream`, forked from the current stream; its gradient
//   for grad_output = 1 right behind its sweeps; denominator (dense engine) on the current stream; the loss reduction
//   after the numerator's sweeps; `early`: the denominator's gradient (+ the numerator's, as its addend) for
//   grad_output = 1 as well.
// Returns {loss, den_alpha, den_beta, den_logz, den_ws, dx_num, dw_num, dx, dW} (undefined where not asked for).
// ------------------------------------------------------------------------------------------------------------
struct EventRing {  // fork / join events of the native steps, per device (created on first use, never destroyed)
  static constexpr int kN = 32;
  hipEvent_t ev[kN] = {};
  unsigned next = 0;
  std::mutex mu;  // concurrent steps on one device (DataParallel threads, several streams): one event per take
  hipEvent_t take() {
    std::lock_guard<std::mutex> lock(mu);
    hipEvent_t& e = ev[next++ % kN];
    if (!e) TORCH_CHECK(hipEventCreateWithFlags(&e, order_event_flags()) == hipSuccess, "hipEventCreate");
    return e;
  }
};
EventRing& event_ring(int dev) {
  static std::mutex mu;
  static auto* rings = new std::map<int, EventRing>();
  std::lock_guard<std::mutex> lock(mu);
  return (*rings)[dev];
}
void order_after(hipStream_t waiter, hipStream_t signaller, int dev) {  // waiter's later work after signaller's earlier work
  hipEvent_t e = event_ring(dev).take();
  TORCH_CHECK(hipEventRecord(e, signaller) == hipSuccess && hipStreamWaitEvent(waiter, e, 0) == hipSuccess, "stream ordering");
}
void used_on(const at::Tensor& t, const c10::hip::HIPStream& s) {
  if (t.defined()) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), s);
}
float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

c10::hip::HIPStream& fill_stream(int dev) {  // a third stream per device: the step's one fill, off both critical paths
  static std::mutex mu;
  static auto* streams = new std::map<int, c10::hip::HIPStream>();
  std::lock_guard<std::mutex> lock(mu);
  auto it = streams->find(dev);
  if (it == streams->end()) it = streams->emplace(dev, c10::hip::getStreamFromPool(false, (c10::DeviceIndex)dev)).first;
  return it->second;
}

std::vector<at::Tensor> asg_forward(const at::Tensor& x, const at::Tensor& W, int64_t desc_ptr, const at::Tensor& ints,
                                    const at::Tensor& floats, const at::Tensor& scale, const at::Tensor& cpos,
                                    const at::Tensor& cneg, bool need_dx, bool need_dw, bool early, int64_t side_stream) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 3,
              "asg_forward: x must be a contiguous float32 [B,T,C] device tensor");
  TORCH_CHECK(W.is_cuda() && W.scalar_type() == at::kFloat && W.is_contiguous(), "asg_forward: W must be contiguous float32");
  const int dev = x.device().index();
  const int B = (int)x.size(0), T = (int)x.size(1), C = (int)x.size(2);
  const auto* d = reinterpret_cast<const wfl_lattice_desc*>(desc_ptr);
  const c10::hip::HIPStream main_s = c10::hip::getCurrentHIPStream(dev);
  const c10::hip::HIPStream side_s = c10::hip::getStreamFromExternal(reinterpret_cast<hipStream_t>(side_stream), dev);
  hipStream_t ms = main_s.stream(), ss = side_s.stream();
  const bool need_grad = need_dx || need_dw;
  early = early && need_grad;
  const auto f32 = x.options();
  EventRing& ring = event_ring(dev);
  // What sits on the caller's stream is what the step's time is made of: the denominator's sweeps, its gradient, the
  // reduction of the transition-gradient partials, the loss reduction.  The numerator and the zero fill of its
  // transition gradient run beside it.
  at::Tensor dx_num = need_dx ? at::empty_like(x) : at::Tensor();
  used_on(dx_num, side_s);
  at::Tensor dw_num;
  hipEvent_t filled = nullptr;
  if (need_dw) {
    const c10::hip::HIPStream& fs = fill_stream(dev);
    c10::hip::HIPStreamGuard guard(fs);  // (the buffer belongs to that stream: nothing of an earlier step can still be using it)
    dw_num = at::zeros_like(W);
    filled = ring.take();
    TORCH_CHECK(hipEventRecord(filled, fs.stream()) == hipSuccess, "hipEventRecord");
    used_on(dw_num, side_s), used_on(dw_num, main_s);
  }
  const int32_t* ip = ints.data_ptr<int32_t>();
  at::Tensor xg, al, be, lz;
  order_after(ss, ms, dev);
  {
    c10::hip::HIPStreamGuard guard(side_s);  // (the numerator's buffers belong to its stream: freed when this call returns)
    int64_t n_xg = 0, n_ab = 0;
    check(wfl_lattice_workspace(d, T, &n_xg, &n_ab), "asg_forward");
    xg = at::empty({std::max<int64_t>(n_xg, 1)}, f32);
    al = at::empty({std::max<int64_t>(n_ab, 1)}, f32);
    if (need_grad) be = at::empty({std::max<int64_t>(n_ab, 1)}, f32);
    lz = at::empty({B}, f32);
    check(wfl_lattice_gather(d, ip, x.data_ptr<float>(), T, C, xg.data_ptr<float>(), nullptr, ss), "asg_forward");
    check(wfl_lattice_forward(d, ip, floats.data_ptr<float>(), xg.data_ptr<float>(), T, W.data_ptr<float>(), WFL_SEMIRING_LOG,
                              al.data_ptr<float>(), fptr(be), nullptr, lz.data_ptr<float>(), ss),
          "asg_forward");
    if (filled) TORCH_CHECK(hipStreamWaitEvent(ss, filled, 0) == hipSuccess, "hipStreamWaitEvent");
    if (need_grad)
      check(wfl_lattice_grad(d, ip, floats.data_ptr<float>(), xg.data_ptr<float>(), T, C, W.data_ptr<float>(),
                             al.data_ptr<float>(), be.data_ptr<float>(), lz.data_ptr<float>(), cneg.data_ptr<float>(),
                             cneg.data_ptr<float>(), nullptr, 0, nullptr, nullptr, fptr(dx_num), fptr(dw_num), ss),
            "asg_forward");
  }
  hipEvent_t num_done = ring.take();
  TORCH_CHECK(hipEventRecord(num_done, ss) == hipSuccess, "hipEventRecord");
  int64_t n_part = 0, n_ws = 0;
  check(wfl_dense_workspace(B, T, C, &n_part, &n_ws), "asg_forward");
  at::Tensor da = at::empty({B, T, C}, f32), db = need_grad ? at::empty({B, T, C}, f32) : at::Tensor();
  at::Tensor dz = at::empty({B}, f32), ws = at::empty({n_ws}, f32.dtype(at::kByte));
  at::Tensor loss = at::empty({}, f32);
  at::Tensor dx, dW, part;
  if (early) {
    if (need_dx) dx = at::empty_like(x);
    if (need_dw) dW = at::empty_like(W), part = at::empty({n_part}, f32);
  }
  auto dense_forward = [&](int parts, hipStream_t st) {
    check(wfl_dense_forward_parts(x.data_ptr<float>(), W.data_ptr<float>(), B, T, C, WFL_SEMIRING_LOG, da.data_ptr<float>(),
                                  fptr(db), nullptr, dz.data_ptr<float>(), ws.data_ptr(), parts, st),
          "asg_forward");
  };
  auto dense_grad = [&](int parts, hipStream_t st) {
    check(wfl_dense_grad_parts(x.data_ptr<float>(), W.data_ptr<float>(), B, T, C, da.data_ptr<float>(), db.data_ptr<float>(),
                               dz.data_ptr<float>(), cpos.data_ptr<float>(), cpos.data_ptr<float>(), nullptr, 0, fptr(dx_num),
                               fptr(dw_num), fptr(dx), fptr(dW), fptr(part), ws.data_ptr(), parts, st),
          "asg_forward");
  };
  // One wait on this stream (for the numerator's launches) and nothing else between its kernels: a wait or an event
  // record between two launches keeps the second from being dispatched under the first one's tail -- ~6 us each,
  // measured (scripts/step_timeline.sh) -- so the loss reduction comes last, back to back with the gradient's
  // launches, rather than on the numerator's stream with an event each way.  (The log-domain launches stay here too:
  // beside the gradient kernel, which fills every SIMD's registers, an "empty" launch of 2 B workgroups only gets
  // through as that kernel's workgroups retire: measured 86 us.)
  dense_forward(WFL_DENSE_ALL, ms);
  TORCH_CHECK(hipStreamWaitEvent(ms, num_done, 0) == hipSuccess, "hipStreamWaitEvent");
  used_on(lz, main_s);
  if (early) dense_grad(WFL_DENSE_ALL, ms);
  check(wfl_reduce_loss(dz.data_ptr<float>(), lz.data_ptr<float>(), scale.data_ptr<float>(), B, 1.0f, 0, loss.data_ptr<float>(), ms),
        "asg_forward");
  return {loss, da, db, dz, ws, dx_num, dw_num, dx, dW};
}

// ------------------------------------------------------------------------------------------------------------
// The launches of a Transducer step WITHOUT a transition model in one native call: counterpart of
// TransducerLossFunction.forward, /root/reference/criterions/transducer.py:239-315, after the batch of alignment
// acceptors has been packed (wfl_transducer_pack_batch, cached per batch): gather (+ row log-sum-exps when the
// log_softmax of transducer.py:186-187 is fused), the sweeps -- with the emission gradient beside them when asked for and
// possible --, the loss reduction, the join.  Same sequence as criterions/transducer.py spells in Python (kept for
// phase timing, transition models and builds without this module).
// Returns ({loss, xg, alpha, beta, logz, row_lse, dx}, in_launch); dx undefined unless in_launch.
// ------------------------------------------------------------------------------------------------------------
std::pair<std::vector<at::Tensor>, bool> lattice_loss_forward(const at::Tensor& x, int64_t desc_ptr, const at::Tensor& ints,
                                                              const at::Tensor& floats, const at::Tensor& scale,
                                                              const at::Tensor& cneg, bool log_softmax, bool want_dx,
                                                              bool need_beta) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 3,
              "lattice_loss_forward: x must be a contiguous float32 [B,T,C] device tensor");
  const int B = (int)x.size(0), T = (int)x.size(1), C = (int)x.size(2);
  const auto* d = reinterpret_cast<const wfl_lattice_desc*>(desc_ptr);
  void* st = current_stream(x);
  const auto f32 = x.options();
  int64_t n_xg = 0, n_ab = 0;
  check(wfl_lattice_workspace(d, T, &n_xg, &n_ab), "lattice_loss_forward");
  at::Tensor xg = at::empty({std::max<int64_t>(n_xg, 1)}, f32), al = at::empty({std::max<int64_t>(n_ab, 1)}, f32);
  at::Tensor be = need_beta ? at::empty({std::max<int64_t>(n_ab, 1)}, f32) : at::Tensor();
  at::Tensor lz = at::empty({B}, f32), loss = at::empty({}, f32);
  at::Tensor lse = log_softmax ? at::empty({B, T}, f32) : at::Tensor();
  at::Tensor dx = (want_dx && need_beta) ? at::empty_like(x) : at::Tensor();
  const int32_t* ip = ints.data_ptr<int32_t>();
  check(wfl_lattice_gather(d, ip, x.data_ptr<float>(), T, C, xg.data_ptr<float>(), fptr(lse), st), "lattice_loss_forward");
  int flag = 0;
  if (dx.defined()) {
    flag = 2;  // (the join comes behind the loss reduction, which then runs under the gradient's tail)
    check(wfl_lattice_forward_grad(d, ip, floats.data_ptr<float>(), xg.data_ptr<float>(), T, C, nullptr, al.data_ptr<float>(),
                                   be.data_ptr<float>(), lz.data_ptr<float>(), cneg.data_ptr<float>(),
                                   log_softmax ? x.data_ptr<float>() : nullptr, fptr(lse), dx.data_ptr<float>(), &flag, st),
          "lattice_loss_forward");
  } else {
    check(wfl_lattice_forward(d, ip, floats.data_ptr<float>(), xg.data_ptr<float>(), T, nullptr, WFL_SEMIRING_LOG,
                              al.data_ptr<float>(), fptr(be), nullptr, lz.data_ptr<float>(), st),
          "lattice_loss_forward");
  }
  check(wfl_reduce_loss(lz.data_ptr<float>(), nullptr, scale.data_ptr<float>(), B, -1.0f, 0, loss.data_ptr<float>(), st),
        "lattice_loss_forward");
  const bool in_launch = flag != 0 && dx.defined();
  if (in_launch) check(wfl_lattice_side_join(st), "lattice_loss_forward");
  if (!in_launch) dx = at::Tensor();
  return {{loss, xg, al, be, lz, lse, dx}, in_launch};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("ctc_fast_backward", &ctc_fast_backward,
        "loss.backward() of a CtcStep loss without the autograd engine (false: not the plain case, use the engine)");
  m.def(
      "order_after",
      [](int64_t waiter, int64_t signaller, int dev) {
        order_after(reinterpret_cast<hipStream_t>(waiter), reinterpret_cast<hipStream_t>(signaller), dev);
      },
      "waiter's later work after signaller's earlier work (raw stream handles): a pooled device-scope event");
  m.def(
      "order_mark",
      [](int64_t signaller, int dev) {
        hipEvent_t e = event_ring(dev).take();
        TORCH_CHECK(hipEventRecord(e, reinterpret_cast<hipStream_t>(signaller)) == hipSuccess, "hipEventRecord");
        return reinterpret_cast<int64_t>(e);
      },
      "a pooled device-scope event recorded behind signaller's work so far (its handle: valid for the next 31 takes)");
  m.def(
      "order_wait",
      [](int64_t waiter, int64_t event) {
        TORCH_CHECK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(waiter), reinterpret_cast<hipEvent_t>(event), 0) == hipSuccess,
                    "hipStreamWaitEvent");
      },
      "waiter's later work after the event of order_mark");
  m.def("built_for_torch", &built_for_torch, "torch version whose headers this module was compiled against");
  m.def("ctc_step", &ctc_step, "CTC loss + eager gradient in one pipelined launch (C++ autograd node)");
  m.def("asg_forward", &asg_forward, "every launch of an ASG step's forward in one native call (criterions/asg.py)");
  m.def("lattice_loss_forward", &lattice_loss_forward,
        "gather, sweeps (+ the gradient beside them), loss reduction and join of a Transducer step without transitions");
  m.def("ctc_reset_host_state", &ctc_reset_host_state, "zero the steps' memory of which launch to start with");
  py::class_<StagedTargets, std::shared_ptr<StagedTargets>>(m, "StagedTargets")
      .def_readonly("B", &StagedTargets::B)
      .def_readonly("n", &StagedTargets::n)
      .def_readonly("max_len", &StagedTargets::max_len);
  m.def("stage_lists", &stage_lists, "stage + upload list-of-int-list targets (None if they are something else)");
  m.def("ctc_loss_staged", &ctc_loss_staged, "the second half of ctc_loss_lists (profiling: bracket the launch alone)");
  m.def("ctc_loss_lists", &ctc_loss_lists,
        "CTCLoss for list-of-int-list targets: staging, upload, checks and the pipelined launch in one native call");
}
