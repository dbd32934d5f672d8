// Microbenchmark: issue cost of packed-f32 VALU (v_pk_mul_f32 / v_pk_add_f32) against the scalar forms
// on gfx950, with several waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
  const float t = threadIdx.x * 1e-3f + seed;
  if (MODE == 0) {           // scalar: 8 mul + 8 add per iteration
    float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; i++) {
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else {                   // packed: 8 pk_mul + 8 pk_add per iteration (twice the flops)
    f2 a0 = {t, t + 8}, a1 = {t + 1, t + 9}, a2 = {t + 2, t + 10}, a3 = {t + 3, t + 11}, a4 = {t + 4, t + 12}, a5 = {t + 5, t + 13},
       a6 = {t + 6, t + 14}, a7 = {t + 7, t + 15};
    const f2 m = {1.0000001f, 1.0000002f}, c = {1e-7f, 2e-7f};
    for (int i = 0; i < iters; i++) {
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    }
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
  }
}

int main()
{
  float *out; hipMalloc(&out, 256 * 256 * 8 * 8 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wavesPerSimd : {1, 2, 4, 8}) {
    const int blocks = 256 * wavesPerSimd;      // 256 CUs x 4 SIMDs, 4 waves per block
    for (int mode = 0; mode < 2; mode++) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
        else           hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double instr = double(iters) * 16;            // VALU instructions per wave
      const double cyc = ms * 1e-3 * 2.4e9;               // at 2.4 GHz
      printf("waves/SIMD %d %s: %.3f ms, %.2f cycles per wave-instruction per SIMD (%.2f per instr per wave)\n", wavesPerSimd,
             mode ? "packed" : "scalar", ms, cyc / (instr * wavesPerSimd), cyc / instr);
    }
  }
  return 0;
}
