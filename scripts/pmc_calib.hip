// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the wfl kernels
// (MI355X_MICROARCH.md, "HBM": FETCH_SIZE reports 1/2 of a 16 B/lane coalesced stream; other widths are
// "uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel moves a KNOWN number
// of bytes; scripts/pmc_calib.py divides the counter by it.
//   build:  hipcc --offload-arch=gfx950 -O3 scripts/pmc_calib.hip -o gpurun_out/pmc_calib
//   run:    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- gpurun_out/pmc_calib
//           (and again with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

// every kernel folds what it reads into one float per thread so that no load is dead code
__global__ void read16_stream(const float4* __restrict__ p, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1234.5f) sink[0] = acc;
}
__global__ void read4_stream(const float* __restrict__ p, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 1234.5f) sink[0] = acc;
}
// the CTC gather: one wave per row of C floats, lane k < K reads column cols[k] of its row (4 B/lane, K scattered
// columns of a row; rows are visited once)
__global__ void read4_gather(const float* __restrict__ p, size_t rows, int C, const int* __restrict__ cols, int K,
                             float* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int col = lane < K ? cols[lane] : -1;
  float acc = 0.f;
  for (size_t r = wave; r < rows; r += nw)
    if (col >= 0) acc += p[r * C + col];
  if (acc == 1234.5f) sink[0] = acc;
}
__global__ void write16_stream(float4* __restrict__ p, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void write4_stream(float* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}
// rows of C floats written as float4 where C*4 is not a multiple of 64 B (the 400-B gradient rows of cfg2)
__global__ void write_rows(float* __restrict__ p, size_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < rows; r += nw)
    if (lane * 4 < C) *reinterpret_cast<float4*>(p + r * C + lane * 4) = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: four times the Infinity Cache
  float *buf, *sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, bytes));
  const int C = 100, K = 45;
  std::vector<int> cols(K);
  for (int k = 0; k < K; ++k) cols[k] = (k * 37 + 11) % 98;  // 45 distinct columns (37 is coprime to 98)
  int* dcols;
  CK(hipMalloc(&dcols, K * sizeof(int)));
  CK(hipMemcpy(dcols, cols.data(), K * sizeof(int), hipMemcpyHostToDevice));
  const size_t rows = bytes / (C * sizeof(float));
  // lines a row's gather touches, for 64-B and 128-B lines (rows of 400 B are not line aligned: count over all rows)
  size_t l64 = 0, l128 = 0;
  {
    std::vector<char> seen;
    for (int sz : {64, 128}) {
      size_t total = 0;
      // the pattern repeats every lcm(400, sz) bytes; count over that many rows and scale
      const size_t period_rows = sz == 64 ? 4 : 8;  // 400*4 = 1600 = 25*64; 400*8 = 3200 = 25*128
      for (size_t r = 0; r < period_rows; ++r) {
        seen.assign(64, 0);
        const size_t base = r * C * 4;
        size_t first = base / sz;
        for (int k = 0; k < K; ++k) {
          size_t line = (base + (size_t)cols[k] * 4) / sz - first;
          if (!seen[line]) seen[line] = 1, ++total;
        }
      }
      (sz == 64 ? l64 : l128) = total * (rows / period_rows);
    }
  }
  printf("CALIB bytes=%zu rows=%zu gather_useful=%zu gather_lines64=%zu gather_lines128=%zu rowwrite_bytes=%zu\n", bytes,
         rows, rows * K * 4, l64 * 64, l128 * 128, rows * C * 4);
  const int grid = 256 * 8, block = 256;
  for (int rep = 0; rep < 3; ++rep) {
    read16_stream<<<grid, block>>>((const float4*)buf, bytes / 16, sink);
    read4_stream<<<grid, block>>>(buf, bytes / 4, sink);
    read4_gather<<<grid, block>>>(buf, rows, C, dcols, K, sink);
    write16_stream<<<grid, block>>>((float4*)buf, bytes / 16);
    write4_stream<<<grid, block>>>(buf, bytes / 4);
    write_rows<<<grid, block>>>(buf, rows, C);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
