// micro-benchmark: effective shader clock seen by a one-wavefront kernel (is a mostly idle GPU clocked down?) and the
// cost of small dependent operations.  hipcc --offload-arch=gfx950 -O3 clk.hip -o clk && ./clk
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(long long *out, int iters, double *sink) {
  const long long c0 = clock64(), w0 = wall_clock64();
  double a = 1.0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;  // dependent fp64 fma chain
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  sink[threadIdx.x] = a;
}
__global__ void k_lds_chain(long long *out, int iters, double *sink) {
  __shared__ double s[64];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  int idx = threadIdx.x;
  double acc = 0;
  for (int i = 0; i < iters; ++i) { const double v = s[idx & 63]; acc += v; idx = (int)v + 1; }  // dependent LDS reads
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  sink[threadIdx.x] = acc;
}
__global__ void k_barrier(long long *out, int iters, double *sink) {
  __shared__ double s[256];
  const long long c0 = clock64(), w0 = wall_clock64();
  double acc = 0;
  for (int i = 0; i < iters; ++i) { s[threadIdx.x] = acc + i; __syncthreads(); acc += s[(threadIdx.x + 1) & 255]; __syncthreads(); }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  sink[threadIdx.x] = acc;
}
int main() {
  long long *d, h[2]; double *sink;
  hipMalloc(&d, 16); hipMalloc(&sink, 4096);
  int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", wall_khz);
  for (int rep = 0; rep < 3; ++rep) {
    k_spin<<<1, 64>>>(d, 20000, sink); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("fma chain  : %lld clk / %d iters = %.1f clk/iter; wall %.2f us -> sclk %.0f MHz\n", h[0], 20000, h[0] / 20000.0, h[1] * 1e3 / wall_khz, h[0] / (h[1] * 1e3 / wall_khz));
    k_lds_chain<<<1, 64>>>(d, 20000, sink); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("lds chain  : %.1f clk/iter; wall %.2f us\n", h[0] / 20000.0, h[1] * 1e3 / wall_khz);
    k_barrier<<<1, 256>>>(d, 2000, sink); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("2 barriers + lds rt: %.1f clk/iter; wall %.2f us\n", h[0] / 2000.0, h[1] * 1e3 / wall_khz);
  }
  return 0;
}
