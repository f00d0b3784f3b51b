// The reference's unfused formulation on MI355X with the vendor FFT, for comparison with the fused
// hand-written path: (1) multiply kernel writes the [Dc][M][N] cube, (2) hipFFT/rocFFT batched inverse
// C2C in place, (3) |.|^2 row sums.  Traffic 32*D*M*N bytes (write, read+write, read) vs 16*D*M*N fused.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/rocfft_ref tools/ubench/rocfft_ref.hip -lhipfft
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdio.h>
#include <vector>
typedef float2 cf;
__global__ void k_mul(cf* out, const cf* X, const cf* masks, const int* shifts, int N, int M, int j0) {
  int x = blockIdx.x * blockDim.x + threadIdx.x; int jl = blockIdx.y;
  int s = shifts[j0 + jl]; cf v = X[(x + s) & (N - 1)];
  for (int m = 0; m < M; ++m) { cf h = masks[(size_t)m * N + x];
    out[((size_t)jl * M + m) * N + x] = make_float2(v.x * h.x - v.y * h.y, v.x * h.y + v.y * h.x); }
}
__global__ void k_abs(float* sum, const cf* in, int N, int rows) {
  int row = blockIdx.y; const cf* p = in + (size_t)row * N; float a = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) a += p[i].x * p[i].x + p[i].y * p[i].y;
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sum[row], a * (1.f / 262144.f));
}
int main() {
  const int N = 1 << 20, M = 8, D = 256, Dc = 32;   // 2 GiB cube chunk
  cf *X, *masks, *cube; int* shifts; float* sum;
  hipMalloc(&X, N * 8); hipMalloc(&masks, (size_t)M * N * 8); hipMalloc(&cube, (size_t)Dc * M * N * 8);
  hipMalloc(&shifts, D * 4); hipMalloc(&sum, D * M * 4);
  std::vector<float> h((size_t)M * N * 2); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(masks, h.data(), (size_t)M * N * 8, hipMemcpyHostToDevice); hipMemcpy(X, h.data(), N * 8, hipMemcpyHostToDevice);
  std::vector<int> sh(D); for (int j = 0; j < D; ++j) sh[j] = (j * 4099) & (N - 1); hipMemcpy(shifts, sh.data(), D * 4, hipMemcpyHostToDevice);
  hipfftHandle plan; int n[1] = {N};
  if (hipfftPlanMany(&plan, 1, n, nullptr, 1, N, nullptr, 1, N, HIPFFT_C2C, Dc * M) != HIPFFT_SUCCESS) { printf("plan failed\n"); return 1; }
  hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  float tm = 0, tf = 0, ta = 0;
  for (int rep = 0; rep < 2; ++rep) { tm = tf = ta = 0;
    hipMemset(sum, 0, D * M * 4);
    for (int j0 = 0; j0 < D; j0 += Dc) {
      hipEventRecord(e0); k_mul<<<dim3(N / 256, Dc), 256>>>(cube, X, masks, shifts, N, M, j0);
      hipEventRecord(e1); hipfftExecC2C(plan, (hipfftComplex*)cube, (hipfftComplex*)cube, HIPFFT_BACKWARD);
      hipEventRecord(e2); k_abs<<<dim3(64, Dc * M), 256>>>(sum + j0 * M, cube, N, Dc * M);
      hipEventRecord(e3); hipEventSynchronize(e3);
      float a, b, c; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&c, e2, e3); tm += a; tf += b; ta += c; } }
  printf("C2 block (D=256, M=8, N=2^20) unfused with hipFFT: multiply %.2f ms + inverse FFTs %.2f ms + |.|^2 sums %.2f ms = %.2f ms  (%.1f Msamples/s)\n",
         tm, tf, ta, tm + tf + ta, (N - 1024) / (tm + tf + ta) / 1e3);
  return 0;
}
