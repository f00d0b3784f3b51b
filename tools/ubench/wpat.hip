// HBM write-pattern microbenchmark: how fast can Z[row][n1][k2] be written in 128-B / 256-B / 512-B
// segments with a 32 KiB stride (pass-1 pattern) versus fully contiguous rows?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
// grid: rows x (N2/SEG) tiles; block 256 threads; each block writes N1=256 segments of SEG elements
template<int SEG> __global__ void wk(v2f* Z, int N2, int ntiles, float val) {
  const int N1 = 256;
  int tile = blockIdx.x % ntiles; size_t row = blockIdx.x / ntiles;
  int col = threadIdx.x % SEG, g = threadIdx.x / SEG;      // g in [0, 256/SEG)
  v2f* base = Z + row * (size_t)N1 * N2 + tile * SEG + col;
  constexpr int RPT = SEG;  // rows per thread = 256 / (256/SEG) = SEG
  #pragma unroll 16
  for (int i = 0; i < RPT; ++i) { int n1 = g + (256 / SEG) * i; base[(size_t)n1 * N2] = (v2f){val + n1, val}; }
}
__global__ void wlin(v2f* Z, size_t n, float val) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) Z[i] = (v2f){val, val};
}
__global__ void rlin(const v2f* Z, size_t n, float* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t stride = (size_t)gridDim.x * blockDim.x;
  v2f acc = {0,0};
  for (; i < n; i += stride) acc += Z[i];
  if (acc.x == 123.456f) out[0] = acc.y;
}
int main() {
  const int N2 = 4096, N1 = 256; const size_t rows = 1024; // 8 GiB
  size_t n = rows * N1 * N2; v2f* Z; hipMalloc(&Z, n * 8); float* o; hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  auto t = [&](const char* name, auto f) { f(); hipEventRecord(e0); f(); f(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= 2; printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, n * 8.0 / ms / 1e9); };
  t("write linear 8B/lane", [&]{ wlin<<<256*8, 256>>>(Z, n, 1.f); });
  t("read  linear 8B/lane", [&]{ rlin<<<256*8, 256>>>(Z, n, o); });
  t("write 128B seg @32KiB", [&]{ wk<16><<<rows * (N2/16), 256>>>(Z, N2, N2/16, 1.f); });
  t("write 256B seg @32KiB", [&]{ wk<32><<<rows * (N2/32), 256>>>(Z, N2, N2/32, 1.f); });
  t("write 512B seg @32KiB", [&]{ wk<64><<<rows * (N2/64), 256>>>(Z, N2, N2/64, 1.f); });
  return 0;
}
