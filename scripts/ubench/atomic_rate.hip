// throughput and latency of returning device-scope atomicAdd on ONE address (a work queue's head) from `grid` wavefronts (lane 0 each),
// interleaved with `work` dependent FMAs between two atomics; and the same with one counter per XCD (XCC_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(unsigned int* ctr, int reps, int work, int per_xcd, unsigned long long* out)
{
  unsigned int xcc = 0;
  if (per_xcd) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u; }
  unsigned int* c = ctr + xcc * 64;
  double x = threadIdx.x;
  unsigned int acc = 0;
  long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {
    if (threadIdx.x == 0) acc += atomicAdd(c, 1u);
    acc = __builtin_amdgcn_readfirstlane(acc);
    for (int w = 0; w < work; ++w) x = __builtin_fma(x, 1.0000001, 1e-9);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x] = (unsigned long long)(t1 - t0) + (acc & 1u) + (x > 1e300 ? 1 : 0); }
}
int main()
{
  unsigned int* ctr; unsigned long long* out;
  hipMalloc(&ctr, 4096); hipMalloc(&out, 8 * 4096);
  for (int per_xcd : {0, 1})
    for (int grid : {1, 256, 2048})
      for (int work : {0, 2000}) {
        const int reps = 2000;
        hipMemset(ctr, 0, 4096);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<grid, 64>>>(ctr, 10, work, per_xcd, out); hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 64>>>(ctr, reps, work, per_xcd, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("%s counter, %4d wavefronts, %4d FMAs between: %.1f M atomics/s, %.2f us per atomic + work per wavefront (%.0f cycles)\n",
               per_xcd ? "per-XCD" : "one    ", grid, work, (double)grid * reps / (ms * 1e-3) / 1e6, ms * 1e3 / reps, (double)h / reps);
      }
  return 0;
}
