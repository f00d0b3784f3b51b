// Does a producer->consumer intermediate that fits the 256 MiB Infinity Cache move faster than HBM?
// alternate: write kernel (pass-1 pattern, 128-B segments @32 KiB) then read kernel (linear), on buffers of 32..1024 MiB
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void wk(v2f* Z, int N2, int ntiles, float val) {
  const int N1 = 256; int tile = blockIdx.x % ntiles; size_t row = blockIdx.x / ntiles;
  int col = threadIdx.x % 16, g = threadIdx.x / 16;
  v2f* base = Z + row * (size_t)N1 * N2 + tile * 16 + col;
  #pragma unroll 16
  for (int i = 0; i < 16; ++i) { int n1 = g + 16 * i; base[(size_t)n1 * N2] = (v2f){val + n1, val}; }
}
__global__ void rlin(const v2f* Z, size_t n, float* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t stride = (size_t)gridDim.x * blockDim.x;
  v2f acc = {0,0};
  for (; i < n; i += stride) acc += Z[i];
  if (acc.x == 123.456f) out[0] = acc.y;
}
int main() {
  const int N2 = 4096, N1 = 256; float* o; hipMalloc(&o, 4);
  hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
  for (size_t rows : {4, 8, 16, 32, 64, 128}) {   // 8 MiB per row
    size_t n = rows * N1 * N2; v2f* Z; hipMalloc(&Z, n * 8);
    const int reps = 20; float tw = 0, tr = 0;
    for (int r = 0; r < reps + 2; ++r) {
      hipEventRecord(e[0]); wk<<<rows * (N2/16), 256>>>(Z, N2, N2/16, 1.f); hipEventRecord(e[1]);
      rlin<<<256*8, 256>>>(Z, n, o); hipEventRecord(e[2]); hipEventSynchronize(e[2]);
      float a, b; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]);
      if (r >= 2) { tw += a; tr += b; }
    }
    printf("buffer %5zu MiB: write %.2f TB/s  read %.2f TB/s   (%.1f us / %.1f us per pass)\n", n * 8 >> 20, n * 8.0 * reps / tw / 1e9, n * 8.0 * reps / tr / 1e9, tw / reps * 1e3, tr / reps * 1e3);
    hipFree(Z);
  }
  return 0;
}
