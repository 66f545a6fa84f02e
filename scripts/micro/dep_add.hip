// micro-benchmark: how often one wave issues a DEPENDENT v_add_f32 (the Parseval chain of k_convolve_sums), alone on its
// SIMD and with 1 ... 3 independent chains interleaved.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dep_add scripts/micro/dep_add.hip && /tmp/dep_add
#include <hip/hip_runtime.h>
#include <cstdio>

template <int J>
__global__ void k_chain(const float *in, float *out, long long *cycles, int n)
{
  float s[J];
  float v[16];
  for (int u = 0; u < 16; u++)
    v[u] = in[u];
  for (int j = 0; j < J; j++)
    s[j] = in[16 + j];
  const long long t0 = clock64();
  for (int k = 0; k < n; k++)
  {
#pragma unroll
    for (int u = 0; u < 16; u++)
    {
#pragma unroll
      for (int j = 0; j < J; j++)
        s[j] += v[u];
    }
    asm volatile("" : "+v"(s[0]));
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int j = 0; j < J; j++)
    r += s[j];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0)
    *cycles = t1 - t0;
}

template <int J>
void run(const float *dIn, float *dOut, long long *dCyc)
{
  const int n = 4096;
  hipLaunchKernelGGL(k_chain<J>, dim3(1), dim3(64), 0, 0, dIn, dOut, dCyc, n);
  hipLaunchKernelGGL(k_chain<J>, dim3(1), dim3(64), 0, 0, dIn, dOut, dCyc, n);
  hipDeviceSynchronize();
  long long c = 0;
  hipMemcpy(&c, dCyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%d chain(s): %.2f clock64 ticks per addition of a chain (%lld ticks for %d x 16 x %d additions)\n", J,
         (double) c / (n * 16.0), c, n, J);
}

int main()
{
  float h[32];
  for (int i = 0; i < 32; i++)
    h[i] = 1e-3f * (i + 1);
  float *dIn, *dOut;
  long long *dCyc;
  hipMalloc(&dIn, sizeof(h));
  hipMalloc(&dOut, 64 * sizeof(float));
  hipMalloc(&dCyc, sizeof(long long));
  hipMemcpy(dIn, h, sizeof(h), hipMemcpyHostToDevice);
  run<1>(dIn, dOut, dCyc);
  run<2>(dIn, dOut, dCyc);
  run<3>(dIn, dOut, dCyc);
  run<4>(dIn, dOut, dCyc);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, shader clock %d kHz (clock64 = s_memtime)\n", rate, clk);
  return 0;
}
