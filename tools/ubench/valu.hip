// VALU issue-rate microbenchmark: scalar v_fma_f32 vs packed v_pk_fma_f32 / v_pk_add_f32 at 1,2,4 waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template<int MODE> __global__ void k(float* out, int iters, float a, float b) {
  float s[16]; v2f p[8];
  for (int i=0;i<16;++i) s[i] = threadIdx.x*0.001f + i;
  for (int i=0;i<8;++i) p[i] = (v2f){s[2*i], s[2*i+1]};
  v2f A = {a, a*0.5f}, B = {b, b*0.25f};
  for (int it=0; it<iters; ++it) {
    if (MODE==0) {
      #pragma unroll
      for (int i=0;i<16;++i) s[i] = __builtin_fmaf(s[i], a, b);
    } else if (MODE==1) {
      #pragma unroll
      for (int i=0;i<8;++i) p[i] = __builtin_elementwise_fma(p[i], A, B);
    } else if (MODE==2) {
      #pragma unroll
      for (int i=0;i<16;++i) s[i] = s[i] + a;
    } else {
      #pragma unroll
      for (int i=0;i<8;++i) p[i] = p[i] + A;
    }
  }
  float r=0; 
  if (MODE==0||MODE==2) for (int i=0;i<16;++i) r+=s[i]; else for (int i=0;i<8;++i) r+=p[i].x+p[i].y;
  out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}
template<int MODE> void run(const char* name, int wavesPerSimd) {
  int blocks = 256, threads = 256*wavesPerSimd; // 4 SIMDs x w waves
  float* d; hipMalloc(&d, blocks*threads*4);
  int iters = 20000;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks,threads>>>(d, 100, 1.0001f, 0.5f);
  hipEventRecord(e0); k<MODE><<<blocks,threads>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double lane_elems = (double)blocks*threads*iters*16; // 16 float results per iter per thread in all modes
  double instr_per_wave = (MODE==0||MODE==2)? 16.0*iters : 8.0*iters;
  double cyc = ms*1e-3*2.4e9;
  printf("%-12s waves/SIMD %d: %.3f ms  %.2f Tflop-elem/s  cycles/instr/wave %.2f  per-SIMD cycles/instr %.2f\n", name, wavesPerSimd, ms, lane_elems/ms/1e9, cyc/instr_per_wave, cyc/(instr_per_wave*wavesPerSimd));
  hipFree(d);
}
int main(){ for (int w: {1,2,4}) { run<0>("v_fma_f32",w); run<1>("v_pk_fma_f32",w); run<2>("v_add_f32",w); run<3>("v_pk_add_f32",w);} return 0; }
