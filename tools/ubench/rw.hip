// Can HBM reads and writes overlap? Run the pass-1-pattern writer and the linear reader concurrently on two streams.
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
  size_t rows = 512; size_t n = rows * N1 * N2; v2f *A, *B; hipMalloc(&A, n * 8); hipMalloc(&B, n * 8);
  hipMemset(A, 0, n*8); hipMemset(B, 0, n*8);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(e0, s1); wk<<<rows * (N2/16), 256, 0, s1>>>(A, N2, N2/16, 1.f); hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("write alone : %.3f ms %.2f TB/s\n", ms, n*8.0/ms/1e9);
    hipEventRecord(e0, s1); rlin<<<256*8, 256, 0, s1>>>(B, n, o); hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("read alone  : %.3f ms %.2f TB/s\n", ms, n*8.0/ms/1e9);
    hipDeviceSynchronize();
    hipEventRecord(e0, s1); hipStreamWaitEvent(s2, e0, 0);
    wk<<<rows * (N2/16), 256, 0, s1>>>(A, N2, N2/16, 1.f);
    rlin<<<256*8, 256, 0, s2>>>(B, n, o);
    hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("write||read : %.3f ms %.2f TB/s aggregate\n", ms, 2*n*8.0/ms/1e9);
  }
  return 0;
}
