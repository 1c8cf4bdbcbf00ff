// fp64 FMA: dependent-chain latency vs independent issue rate, 1 and 2 wavefronts per SIMD (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double* out, int n, double a, double b)
{
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int CHAINS>
void run(int waves_per_simd)
{
  double* d; hipMalloc(&d, 1 << 20);
  const int n = 20000;
  // one workgroup per CU is not controllable; use 256 threads (4 waves = 1 per SIMD) x waves_per_simd, grid = 1 WG
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS><<<1, 256 * waves_per_simd>>>(d, 100, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS><<<1, 256 * waves_per_simd>>>(d, n, 1.0000001, 1e-9);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("chains %d, %d wave(s)/SIMD: %.2f cycles per FMA instruction per wave (clock64), %.3f ms -> %.2f ns per FMA-instr per SIMD\n",
         CHAINS, waves_per_simd, h / ((double)n * CHAINS), ms, ms * 1e6 / ((double)n * CHAINS * waves_per_simd));
  hipFree(d);
}
int main()
{
  run<1>(1); run<1>(2); run<2>(1); run<2>(2); run<4>(1); run<4>(2); run<8>(1); run<8>(2);
  return 0;
}
