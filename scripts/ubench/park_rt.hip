// What does a wavefront pay to dump ROWS rows of 64 doubles (lane-contiguous, 512 B each) to HBM and read them back -- the park /
// unpark of k_flat2<.., SLICED> -- with (a) plain stores / loads, (b) agent-scope relaxed atomic stores / loads (sc1: what instances
// that migrate between XCDs need), (c) plain accesses to fine-grained (uncached) device memory?  2048 wavefronts, each its own records.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ROWS = 40;
template <int MODE>
__global__ void __launch_bounds__(64) k(double* buf, int nrec, int reps, double* out)
{
  const int lane = threadIdx.x;
  double x[ROWS];
  for (int r = 0; r < ROWS; ++r) x[r] = lane + r * 0.5;
  double acc = 0;
  long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {
    double* rec = buf + (size_t)((blockIdx.x * 7 + i * 131) % nrec) * ROWS * 64;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (MODE == 1) __hip_atomic_store(rec + r * 64 + lane, x[r] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else rec[r * 64 + lane] = x[r] + i;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (MODE == 1) x[r] = __hip_atomic_load(rec + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else x[r] = rec[r * 64 + lane];
    }
    for (int r = 0; r < ROWS; ++r) acc += x[r];
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int MODE>
void run(const char* name, bool fine, int grid)
{
  const int nrec = 65536, reps = 50;
  double *buf, *out;
  const size_t bytes = (size_t)nrec * ROWS * 64 * 8;
  if (fine) { if (hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocFinegrained) != hipSuccess) { printf("%s: no fine-grained memory\n", name); return; } }
  else hipMalloc(&buf, bytes);
  hipMalloc(&out, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 64>>>(buf, nrec, 2, out); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 64>>>(buf, nrec, reps, out);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("%-44s grid %5d: %.2f us per dump + read-back per wavefront (clock64: %.0f cycles), %.1f GB/s\n", name, grid, ms * 1e3 / reps, h / reps,
         2.0 * ROWS * 512 * (double)grid * reps / (ms * 1e-3) / 1e9);
  hipFree(buf); hipFree(out);
}
int main()
{
  for (int grid : {1, 2048}) {
    run<0>("plain stores / loads", false, grid);
    run<1>("agent-scope relaxed atomic stores / loads", false, grid);
    run<0>("plain, fine-grained allocation", true, grid);
    run<1>("agent-scope atomics, fine-grained allocation", true, grid);
  }
  return 0;
}
