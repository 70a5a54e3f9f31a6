// micro-benchmark (round 6): dependent-issue cost of the operations the 32 x 32 factor's pivot chain is made of, ONE wavefront
// on an otherwise idle CU (the situation of the lead workgroup of k_chol_step).  Shader clocks per operation.
//   hipcc --offload-arch=gfx950 -O3 dep64.hip -o dep64 && ./dep64
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define REP8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP32(...) REP8(__VA_ARGS__) REP8(__VA_ARGS__) REP8(__VA_ARGS__) REP8(__VA_ARGS__)

template <int WHAT>
__global__ void __launch_bounds__(256) k_chain(long long *out, int iters, double *sink) {
  __shared__ double s[1024];
  const int tid = threadIdx.x;
  s[tid] = 1.0 + tid * 1e-9;
  s[tid + 256] = s[tid + 512] = s[tid + 768] = 0.5;
  __syncthreads();
  double a = 1.0 + tid * 1e-9, b = 1.0000001, c = 1e-9;
  double a1 = a + 1, a2 = a + 2, a3 = a + 3;
  float fa = 1.0f + tid * 1e-6f, fb = 1.000001f, fc = 1e-6f;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  int ia = tid;
  const long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (WHAT == 0) {  // dependent v_fma_f64
      REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    } else if (WHAT == 1) {  // dependent v_mul_f64
      REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
    } else if (WHAT == 2) {  // dependent v_rsq_f64
      REP32(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));)
    } else if (WHAT == 3) {  // dependent v_rcp_f64
      REP32(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
    } else if (WHAT == 4) {  // four independent v_fma_f64 chains (issue rate)
      REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                        : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3)
                        : "v"(b), "v"(c));)
    } else if (WHAT == 5) {  // dependent v_fma_f32
      REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));)
    } else if (WHAT == 6) {  // MFMA f64 16x16x4, dependent through the accumulator
      REP32(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);)
    } else if (WHAT == 7) {  // MFMA f64 16x16x4, the result feeds the next one's A operand
      REP32(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], b, f64x4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);)
    } else if (WHAT == 8) {  // v_readlane x2 -> v_fma_f64 with the scalar (dependent)
      REP32({
        int lo, hi;
        asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(a)), "v"(__double2hiint(a)));
        const double sc = __hiloint2double(hi, lo);
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "s"(sc), "v"(c));
      })
    } else if (WHAT == 9) {  // ds_bpermute x2 (a 64-bit value from another lane) -> fma (dependent)
      REP32({
        const int lo = __builtin_amdgcn_ds_bpermute(((tid + 5) & 63) << 2, __double2loint(a));
        const int hi = __builtin_amdgcn_ds_bpermute(((tid + 5) & 63) << 2, __double2hiint(a));
        a = __hiloint2double(hi, lo);
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      })
    } else if (WHAT == 10) {  // ds_write_b64 -> ds_read_b64 of another lane's slot, same wavefront (dependent)
      REP32({
        s[tid & 63] = a;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        a = s[(tid + 7) & 63];
      })
    } else if (WHAT == 11) {  // DPP row_shr:1 move x2 -> fma (dependent)
      REP32({
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, false);
        a = __hiloint2double(hi, lo) + 1.0;
      })
    } else if (WHAT == 12) {  // LDS write, workgroup barrier, LDS read (4 wavefronts)
      REP8({
        s[tid] = a;
        __syncthreads();
        a = s[(tid + 64) & 255];
        __syncthreads();
      })
    } else if (WHAT == 13) {  // dependent ds_read_b64 (address from the value)
      REP32({
        const double v = s[ia & 255];
        ia = (int)v + 1;
      })
    } else if (WHAT == 14) {  // v_permlane32_swap (gfx950): upper and lower half of the wavefront exchange, dependent
      REP32({
        int lo = __double2loint(a), hi = __double2hiint(a), lo2 = lo, hi2 = hi;
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
        a = __hiloint2double(hi, lo) + 1.0;
      })
    }
  }
  const long long c1 = clock64();
  if (tid == 0) out[0] = c1 - c0;
  sink[tid] = a + a1 + a2 + a3 + fa + acc[0] + acc[1] + acc[2] + acc[3] + ia;
}

template <int WHAT>
static void run(const char *name, int threads, int ops_per_iter) {
  long long *d, h;
  double *sink;
  hipMalloc(&d, 16);
  hipMalloc(&sink, 4096);
  const int iters = 200;
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    k_chain<WHAT><<<1, threads>>>(d, iters, sink);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double v = (double)h / ((double)iters * ops_per_iter);
    best = v < best ? v : best;
  }
  printf("%-72s %7.1f clk/op\n", name, best);
  hipFree(d);
  hipFree(sink);
}

int main() {
  run<0>("dependent v_fma_f64, 1 wavefront", 64, 32);
  run<0>("dependent v_fma_f64, 4 wavefronts (one per SIMD)", 256, 32);
  run<1>("dependent v_mul_f64", 64, 32);
  run<2>("dependent v_rsq_f64", 64, 32);
  run<3>("dependent v_rcp_f64", 64, 32);
  run<4>("4 independent v_fma_f64 chains (per fma)", 64, 32);
  run<5>("dependent v_fma_f32", 64, 32);
  run<6>("v_mfma_f64_16x16x4, dependent accumulator", 64, 32);
  run<7>("v_mfma_f64_16x16x4, result -> next A operand", 64, 32);
  run<8>("2 x v_readlane_b32 -> v_fma_f64 (scalar operand), dependent", 64, 32);
  run<9>("2 x ds_bpermute_b32 -> v_fma_f64, dependent", 64, 32);
  run<10>("ds_write_b64 -> ds_read_b64 (other lane), same wavefront", 64, 32);
  run<11>("2 x DPP row_shr:1 -> v_add_f64, dependent", 64, 32);
  run<12>("LDS write + barrier + LDS read + barrier, 4 wavefronts", 256, 8);
  run<13>("dependent ds_read_b64 + cvt", 64, 32);
  run<14>("2 x v_permlane32_swap -> v_add_f64, dependent", 64, 32);
  return 0;
}
