// fp64 FMA issue rate of a SIMD as a function of resident wavefronts and independent chains per wavefront (MI355X): what the
// "fp64 vector roof" of k_flat2 / k_flat1 is in practice, and how many wavefronts x chains it takes to reach it.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fp64_rate scripts/ubench/fp64_rate.hip && scripts/ubench/fp64_rate
// One workgroup of 256 x W threads on one CU = W wavefronts on each of its four SIMDs; time by wall clock around a long loop.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int KIND>
__global__ void k(double* out, int n, double a, double b)
{
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (KIND == 0) x[c] = __builtin_fma(x[c], a, b);
      else if (KIND == 1) {  // a DPP move pair + add (the prefix sums' step)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[c]), 0x111, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x[c]), 0x111, 0xF, 0xF, true);
        x[c] += __hiloint2double(hi, lo) * a;
      } else {               // permlane32_swap pair + add (pair_sum)
        const int xl = __double2loint(x[c]), xh = __double2hiint(x[c]);
        const auto rl = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
        x[c] = (__hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1])) * a;
      }
    }
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, int KIND>
void run(int waves_per_simd, int ncu)
{
  double* d; hipMalloc(&d, 64 << 20);
  const int n = 40000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS, KIND><<<ncu, 256 * waves_per_simd>>>(d, 100, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS, KIND><<<ncu, 256 * waves_per_simd>>>(d, n, 1.0000001, 1e-9);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd_ns = ms * 1e6 / ((double)n * CHAINS * waves_per_simd);
  printf("kind %d chains %2d waves/SIMD %d CUs %3d: %.3f ms, %.2f ns per step per SIMD", KIND, CHAINS, waves_per_simd, ncu, ms, per_simd_ns);
  if (KIND == 0) printf("  -> %.1f TFLOP/s on 256 CUs (2 x 64 flop per instruction)", 2.0 * 64 / per_simd_ns * 1e-3 * 1024);
  printf("\n");
  hipFree(d);
}
int main()
{
  for (int w = 1; w <= 4; ++w) { run<1, 0>(w, 1); run<2, 0>(w, 1); run<4, 0>(w, 1); run<8, 0>(w, 1); run<16, 0>(w, 1); }
  run<8, 0>(2, 256); run<8, 0>(4, 256); run<16, 0>(2, 256);
  for (int w = 1; w <= 3; ++w) { run<1, 1>(w, 1); run<4, 1>(w, 1); run<1, 2>(w, 1); run<4, 2>(w, 1); }
  return 0;
}
