// div_exact — is the rope walk's short division (exa_kernels.hip: refinedRcp + divByRcp) the correctly rounded quotient on
// its stated input range?  It is the hardware IEEE sequence without v_div_scale / v_div_fmas' scaling / v_div_fixup, so the
// answer must be "always"; this program checks it against the compiler's a / d on the GPU:
//   numerators  a = plane - o   with planes on the half-integer grid (|plane| <= 4096) and origins of any magnitude in range,
//               and raw random floats of 2^-60 <= |a| <= 2^60 (and 0)
//   denominators 2^-30 <= |d| <= 2, either sign (components of a direction, and a little beyond)
// usage: div_exact [million pairs, default 4000]          prints the number of mismatching quotients (expected 0)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/div_exact.hip -o tools/micro/div_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ float refinedRcp(float d)
{
  const float y = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, y, 1.f), y, y);
}
__device__ __forceinline__ float divByRcp(float a, float d, float y)
{
  float q = a * y;
  float r = __builtin_fmaf(-d, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-d, q, a);
  return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ uint32_t mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (uint32_t)x;
}
// a float of the given sign with exponent in [elo, ehi] (unbiased) and random mantissa
__device__ __forceinline__ float randomFloat(uint32_t bits, int elo, int ehi)
{
  const int e = elo + int((bits >> 23) % uint32_t(ehi - elo + 1));
  return __uint_as_float((bits & 0x80000000u) | (uint32_t(e + 127) << 23) | (bits & 0x007fffffu));
}
__global__ void check(unsigned long long pairsPerThread, unsigned long long seed, unsigned long long *bad, float *firstBad)
{
  const unsigned long long tid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  unsigned long long mine = 0;
  for (unsigned long long i = 0; i < pairsPerThread; i++) {
    const unsigned long long k = (tid * pairsPerThread + i) * 4ull + seed;
    const uint32_t r0 = mix(k), r1 = mix(k + 1), r2 = mix(k + 2), r3 = mix(k + 3);
    const float d = randomFloat(r0, -30, 0) * ((r3 & 1u) ? 2.f : 1.f);                 // 2^-30 <= |d| < 2 (and up to < 4 never: capped below)
    if (!(fabsf(d) <= 2.f)) continue;
    float a;
    switch (r3 >> 30) {
      case 0: {                                                   // plane - origin, origin near the grid
        const float plane = 0.5f * float(int(r1 % 16385u) - 8192);
        const float o = randomFloat(r2, -20, 13);
        a = plane - o;
        break;
      }
      case 1: {                                                   // plane - origin, origin far away
        const float plane = 0.5f * float(int(r1 % 16385u) - 8192);
        const float o = randomFloat(r2, 10, 40);
        a = plane - o;
        break;
      }
      case 2: a = randomFloat(r1, -60, 60); break;                // anything in range
      default: a = (r1 & 7u) ? 0.5f * float(int(r1 % 4097u) - 2048) : 0.f;   // exact grid values and zero
    }
    if (a != 0.f && !(fabsf(a) >= 8.6736173798840355e-19f /* 2^-60 */ && fabsf(a) <= 1.152921504606847e18f /* 2^60 */)) continue;
    const float ref = a / d;
    const float got = divByRcp(a, d, refinedRcp(d));
    if (__float_as_uint(ref) != __float_as_uint(got)) {
      if (atomicAdd(bad, 1ull) == 0ull) { firstBad[0] = a; firstBad[1] = d; firstBad[2] = ref; firstBad[3] = got; }
    }
    mine++;
  }
  atomicAdd(bad + 1, mine);
}

int main(int argc, char **argv)
{
  const unsigned long long millions = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000ull;
  const unsigned threads = 256, blocks = 256 * 32;
  const unsigned long long perThread = (millions * 1000000ull + threads * blocks - 1) / (threads * (unsigned long long)blocks);
  unsigned long long *bad; float *firstBad;
  if (hipMalloc((void **)&bad, 16) != hipSuccess || hipMalloc((void **)&firstBad, 16) != hipSuccess) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  (void)hipMemset(bad, 0, 16);
  hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, nullptr, perThread, 0x9E3779B97F4A7C15ull, bad, firstBad);
  unsigned long long h[2]; float fb[4];
  if (hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  (void)hipMemcpy(fb, firstBad, 16, hipMemcpyDeviceToHost);
  std::printf("div_exact: %llu quotients checked, %llu differ from a / d\n", h[1], h[0]);
  if (h[0]) std::printf("  first: a=%a d=%a  a/d=%a  short=%a\n", fb[0], fb[1], fb[2], fb[3]);
  return h[0] ? 1 : 0;
}
