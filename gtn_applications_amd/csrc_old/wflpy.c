/* _wflpy: CPython helper of the operator layer (gtn_applications_amd/engine.py).  One job: turn the targets the
 * reference's criteria receive -- a list of int lists (benchmarks/ctc_benchmark.py:23-24, train.py's
 * `[t.tolist() for t in targets]`) -- into the flat int32 + int64 offsets layout of the C ABI without a
 * Python-level loop or a numpy nested-sequence conversion (5632 labels: ~150 us in numpy, ~25 us here).
 * Not part of libwfl.so: the C ABI stays free of Python; this is glue on the Python side of it. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* flatten_into(targets, flat_addr, flat_capacity, offsets_addr) -> (total, max_len, min_label, max_label)
 * targets: list/tuple of list/tuple of ints.  Writes int32 labels to flat_addr (capacity in elements) and
 * int64 offsets [B+1] to offsets_addr.  Returns None if the capacity is too small (nothing useful written)
 * and raises TypeError for anything that is not a sequence of int sequences (the caller falls back). */
static PyObject* flatten_into(PyObject* self, PyObject* args) {
  PyObject* targets;
  unsigned long long flat_addr, off_addr;
  Py_ssize_t capacity;
  if (!PyArg_ParseTuple(args, "OKnK", &targets, &flat_addr, &capacity, &off_addr)) return NULL;
  if (!PyList_Check(targets) && !PyTuple_Check(targets)) {
    PyErr_SetString(PyExc_TypeError, "targets must be a list or tuple");
    return NULL;
  }
  int32_t* flat = (int32_t*)(uintptr_t)flat_addr;
  int64_t* off = (int64_t*)(uintptr_t)off_addr;
  const Py_ssize_t B = PySequence_Fast_GET_SIZE(targets);
  PyObject** rows = PySequence_Fast_ITEMS(targets);
  Py_ssize_t total = 0, max_len = 0;
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* r = rows[b];
    if (!PyList_Check(r) && !PyTuple_Check(r)) {
      PyErr_SetString(PyExc_TypeError, "every target must be a list or tuple of ints");
      return NULL;
    }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(r);
    off[b] = (int64_t)total;
    total += n;
    if (n > max_len) max_len = n;
  }
  off[B] = (int64_t)total;
  if (total > capacity) Py_RETURN_NONE;
  long lo = 0, hi = -1;
  int first = 1;
  Py_ssize_t k = 0;
  for (Py_ssize_t b = 0; b < B; ++b) {
    PyObject* r = rows[b];
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(r);
    PyObject** it = PySequence_Fast_ITEMS(r);
    for (Py_ssize_t i = 0; i < n; ++i) {
      const long v = PyLong_AsLong(it[i]);
      if (v == -1 && PyErr_Occurred()) return NULL; /* not an int (or overflow) */
      if (v > INT32_MAX || v < INT32_MIN) {
        PyErr_SetString(PyExc_OverflowError, "target label does not fit int32");
        return NULL;
      }
      if (first || v < lo) lo = v;
      if (first || v > hi) hi = v;
      first = 0;
      flat[k++] = (int32_t)v;
    }
  }
  return Py_BuildValue("nnll", total, max_len, lo, hi);
}

/* factors_into(offsets_addr, B, fac_addr): the per-utterance loss / gradient factors of both reductions, six float
 * arrays of B back to back at fac_addr -- scale_none = 1, scale_mean = 1/len (1 for an empty target), then both
 * times +1/B and times -1/B (ctc.py:53-58,87; asg.py:116-121,171-179) -- from the int64 offsets [B+1]. */
static PyObject* factors_into(PyObject* self, PyObject* args) {
  unsigned long long off_addr, fac_addr;
  Py_ssize_t B;
  if (!PyArg_ParseTuple(args, "KnK", &off_addr, &B, &fac_addr)) return NULL;
  const int64_t* off = (const int64_t*)(uintptr_t)off_addr;
  float* fac = (float*)(uintptr_t)fac_addr;
  const float inv_b = 1.0f / (float)(B > 0 ? B : 1);
  for (Py_ssize_t b = 0; b < B; ++b) {
    const float ln = (float)(off[b + 1] - off[b]);
    const float mean = ln > 0.f ? 1.0f / ln : 1.0f;
    fac[b] = 1.0f, fac[B + b] = mean;
    fac[2 * B + b] = 1.0f * inv_b, fac[3 * B + b] = mean * inv_b;
    fac[4 * B + b] = 1.0f * -inv_b, fac[5 * B + b] = mean * -inv_b;
  }
  Py_RETURN_NONE;
}

/* content_key(addr, nbytes) -> (h1, h2): a 128-bit hash of the staged target bytes (the MurmurHash3 x64_128 mixing
 * steps), the key of the operator layer's small content cache.  Python's own hash of a 23 KB bytes object costs more
 * than staging the targets does; a hit is still confirmed byte for byte (same_bytes) before it is used. */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33, k *= 0xff51afd7ed558ccdULL, k ^= k >> 33, k *= 0xc4ceb9fe1a85ec53ULL, k ^= k >> 33;
  return k;
}
static PyObject* content_key(PyObject* self, PyObject* args) {
  unsigned long long addr;
  Py_ssize_t n;
  if (!PyArg_ParseTuple(args, "Kn", &addr, &n)) return NULL;
  const uint8_t* p = (const uint8_t*)(uintptr_t)addr;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = 0x9e3779b97f4a7c15ULL, h2 = 0xd1b54a32d192ed03ULL;
  const Py_ssize_t nb = n / 16;
  for (Py_ssize_t i = 0; i < nb; ++i) {
    uint64_t k1, k2;
    memcpy(&k1, p + 16 * i, 8), memcpy(&k2, p + 16 * i + 8, 8);
    k1 *= c1, k1 = rotl64(k1, 31), k1 *= c2, h1 ^= k1;
    h1 = rotl64(h1, 27), h1 += h2, h1 = h1 * 5 + 0x52dce729;
    k2 *= c2, k2 = rotl64(k2, 33), k2 *= c1, h2 ^= k2;
    h2 = rotl64(h2, 31), h2 += h1, h2 = h2 * 5 + 0x38495ab5;
  }
  uint64_t t1 = 0, t2 = 0;
  const Py_ssize_t rem = n - 16 * nb;
  if (rem > 8) memcpy(&t2, p + 16 * nb + 8, (size_t)(rem - 8));
  if (rem > 0) memcpy(&t1, p + 16 * nb, (size_t)(rem > 8 ? 8 : rem));
  t2 *= c2, t2 = rotl64(t2, 33), t2 *= c1, h2 ^= t2;
  t1 *= c1, t1 = rotl64(t1, 31), t1 *= c2, h1 ^= t1;
  h1 ^= (uint64_t)n, h2 ^= (uint64_t)n;
  h1 += h2, h2 += h1;
  h1 = fmix64(h1), h2 = fmix64(h2);
  h1 += h2, h2 += h1;
  return Py_BuildValue("KK", (unsigned long long)h1, (unsigned long long)h2);
}

/* same_bytes(addr, bytes) -> bool: memcmp of a staged buffer against the bytes a cache entry was made from */
static PyObject* same_bytes(PyObject* self, PyObject* args) {
  unsigned long long addr;
  PyObject* b;
  if (!PyArg_ParseTuple(args, "KS", &addr, &b)) return NULL;
  if (memcmp((const void*)(uintptr_t)addr, PyBytes_AS_STRING(b), (size_t)PyBytes_GET_SIZE(b)) == 0) Py_RETURN_TRUE;
  Py_RETURN_FALSE;
}

static PyMethodDef methods[] = {
    {"flatten_into", flatten_into, METH_VARARGS, "flatten list-of-int-lists targets into int32 flat + int64 offsets"},
    {"factors_into", factors_into, METH_VARARGS, "per-utterance loss / gradient factors from the offsets"},
    {"content_key", content_key, METH_VARARGS, "128-bit hash of a staged buffer"},
    {"same_bytes", same_bytes, METH_VARARGS, "memcmp of a staged buffer against a bytes object"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_wflpy", "operator-layer helpers of gtn_applications_amd", -1,
                                    methods};
PyMODINIT_FUNC PyInit__wflpy(void) { return PyModule_Create(&moddef); }
