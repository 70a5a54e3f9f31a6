// micro-benchmark: LDS read issue cost on one CU with 16 wavefronts streaming (the regime of k_roots' sequential pass):
// clk per wave-level read instruction for ds_read_b64 / ds_read_b128, with 11 or 64 active lanes.
// hipcc --offload-arch=gfx950 -O3 lds.hip -o lds && ./lds
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WIDTH, int ACTIVE>
__global__ void __launch_bounds__(1024) k_lds(long long *out, int iters, double *sink) {
  __shared__ double s[16 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 1024) s[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const long long c0 = clock64();
  if (lane < ACTIVE) {
    const double *p = s + w * 1024 + lane * 2;
    for (int i = 0; i < iters; ++i) {
      asm volatile("" ::: "memory");  // the LDS contents could have changed: every trip really reads
      const int o = (i & 3) * 128;
      if (WIDTH == 8) {
        a0 += p[o], a1 += p[o + 256], a2 += p[o + 512], a3 += p[o + 768];
      } else {
        const double2 v0 = *(const double2 *)(p + o), v1 = *(const double2 *)(p + o + 256), v2 = *(const double2 *)(p + o + 512),
                      v3 = *(const double2 *)(p + o + 768);
        a0 += v0.x + v0.y, a1 += v1.x + v1.y, a2 += v2.x + v2.y, a3 += v3.x + v3.y;
      }
    }
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0) out[0] = c1 - c0;
  sink[threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int WIDTH, int ACTIVE>
void run(const char *name, long long *d, double *sink) {
  long long h;
  const int iters = 4000;
  k_lds<WIDTH, ACTIVE><<<1, 1024>>>(d, iters, sink);
  k_lds<WIDTH, ACTIVE><<<1, 1024>>>(d, iters, sink);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  // 16 waves x 4 reads per iteration share the CU's LDS
  printf("%-28s %6.2f clk per wave-level read at 16 waves (%.2f clk per iteration of 4 reads per wave)\n", name, h / (double)iters / 4 / 16, h / (double)iters);
}
int main() {
  long long *d;
  double *sink;
  hipMalloc(&d, 16);
  hipMalloc(&sink, 8192);
  run<8, 64>("ds_read_b64, 64 lanes", d, sink);
  run<8, 11>("ds_read_b64, 11 lanes", d, sink);
  run<16, 64>("ds_read_b128, 64 lanes", d, sink);
  run<16, 11>("ds_read_b128, 11 lanes", d, sink);
  run<16, 33>("ds_read_b128, 33 lanes", d, sink);
  run<8, 33>("ds_read_b64, 33 lanes", d, sink);
  return 0;
}
