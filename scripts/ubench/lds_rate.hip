// LDS-pipe cost of the 64-bit LDS instructions the flat kernels' loops are made of (MI355X): cycles of a CU's LDS pipe per
// wave-instruction for ds_read_b64, ds_read2_b64, ds_read_b128, ds_write_b64, ds_write2_b64, ds_write_b128 -- lane-contiguous
// (conflict-free) addresses and a uniform (broadcast) address.  MI355X_MICROARCH.md's LDS table says ds_read2_b64 moves 128 B/clk
// where ds_read_b64 / ds_read_b128 move 256; the compiler forms ds_read2_b64 wherever two doubles sit behind one base register.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_rate scripts/ubench/lds_rate.hip && /tmp/lds_rate
// One workgroup of 512 threads on one CU (two wavefronts per SIMD: the pipe, not a wavefront's issue, is the limit), every wavefront
// issues N instructions of the kind back to back (eight in flight); cycles = clock64 of the whole loop / instructions of the CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, bool BCAST>
__global__ void k(double* out, int n, unsigned long long* cyc)
{
  __shared__ __attribute__((aligned(16))) double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // byte address: contiguous per lane (8 B apart for b64, 16 B for b128 / the read2 pairs), or one address for everybody
  const unsigned int a8 = BCAST ? 0u : (unsigned int)lane * 8u, a16 = BCAST ? 0u : (unsigned int)lane * 16u;
  const unsigned int base = (unsigned int)(size_t)lds + (unsigned int)(wave & 3) * 2048u;
  double acc = 0;
  double v0[2], v1[2], v2[2], v3[2];
  const unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {   // 4 x ds_read_b64
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0[0]), "=&v"(v1[0]), "=&v"(v2[0]), "=&v"(v3[0]) : "v"(base + a8) : "memory");
      acc += v0[0] + v1[0] + v2[0] + v3[0];
    } else if (KIND == 1) {   // 4 x ds_read2_b64 (two doubles 512 B apart each)
      asm volatile("ds_read2_b64 %0, %4 offset1:64\n\tds_read2_b64 %1, %4 offset0:1 offset1:65\n\tds_read2_b64 %2, %4 offset0:128 offset1:192\n\tds_read2_b64 %3, %4 offset0:129 offset1:193\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v0)), "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v1)),
                     "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v2)), "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v3))
                   : "v"(base + a8) : "memory");
      acc += v0[0] + v1[1] + v2[0] + v3[1];
    } else if (KIND == 2) {   // 4 x ds_read_b128
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4\n\tds_read_b128 %3, %4 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v0)), "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v1)),
                     "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v2)), "=&v"(*reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(v3))
                   : "v"(base + a16) : "memory");
      acc += v0[0] + v1[1] + v2[0] + v3[1];
    } else if (KIND == 3) {   // 4 x ds_write_b64
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                   :: "v"(base + a8), "v"(acc) : "memory");
    } else if (KIND == 4) {   // 4 x ds_write2_b64
      asm volatile("ds_write2_b64 %0, %1, %1 offset1:64\n\tds_write2_b64 %0, %1, %1 offset0:128 offset1:192\n\tds_write2_b64 %0, %1, %1 offset1:64\n\tds_write2_b64 %0, %1, %1 offset0:128 offset1:192\n\ts_waitcnt lgkmcnt(0)"
                   :: "v"(base + a8), "v"(acc) : "memory");
    } else if (KIND == 6) {   // 4 x ds_bpermute_b32 (the LDS crossbar, no memory)
      int r0, r1, r2, r3;
      asm volatile("ds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %5\n\tds_bpermute_b32 %2, %4, %5\n\tds_bpermute_b32 %3, %4, %5\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(BCAST ? 0u : (unsigned int)((lane * 7 + 3) & 63) * 4u), "v"(lane) : "memory");
      acc += r0 + r1 + r2 + r3;
    } else if (KIND == 7) {   // 4 x ds_write_b64 with four lanes active
      if (lane < 4)
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                   :: "v"(base + a8), "v"(acc) : "memory");
    } else if (KIND == 8) {   // 4 x ds_read_b64, a gather with a two-way bank conflict in every group of 32 lanes (lanes l and l + 16 on one bank pair, different addresses)
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0[0]), "=&v"(v1[0]), "=&v"(v2[0]), "=&v"(v3[0]) : "v"(base + (unsigned int)(lane & 15) * 8u + (unsigned int)((lane >> 4) & 1) * 256u) : "memory");
      acc += v0[0] + v1[0] + v2[0] + v3[0];
    } else {                  // 4 x ds_write_b128
      __attribute__((ext_vector_type(2))) double w2 = {acc, acc};
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                   :: "v"(base + a16), "v"(w2) : "memory");
    }
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, bool BCAST>
void run(const char* name, int bytes_per_lane)
{
  double* d; unsigned long long* c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 1024);
  const int n = 20000, threads = 1024;
  k<KIND, BCAST><<<1, threads>>>(d, 100, c);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<KIND, BCAST><<<1, threads>>>(d, n, c);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
  const double insts = (double)n * 4 * (threads / 64);
  // clock64 counts at 100 MHz on this part (wall clock): use the event time and an assumed 2.4 GHz beside it
  printf("%-20s %-10s: %.3f ms for %.0f wave-instructions of the CU: %.2f ns each = %.1f cycles at 2.4 GHz (%.0f B/clk)\n", name, BCAST ? "broadcast" : "contiguous", ms, insts,
         ms * 1e6 / insts, ms * 1e6 / insts * 2.4, 64.0 * bytes_per_lane / (ms * 1e6 / insts * 2.4));
  hipFree(d); hipFree(c);
}
int main()
{
  run<0, false>("ds_read_b64", 8); run<1, false>("ds_read2_b64", 16); run<2, false>("ds_read_b128", 16);
  run<0, true>("ds_read_b64", 8); run<1, true>("ds_read2_b64", 16); run<2, true>("ds_read_b128", 16);
  run<6, false>("ds_bpermute_b32", 4); run<6, true>("ds_bpermute_b32", 4); run<7, false>("ds_write_b64 4 lanes", 8); run<8, false>("ds_read_b64 2-way", 8);
  run<3, false>("ds_write_b64", 8); run<4, false>("ds_write2_b64", 16); run<5, false>("ds_write_b128", 16);
  return 0;
}
