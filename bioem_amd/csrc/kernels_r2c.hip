// kernels_r2c.hip -- the fast r2c of projections and particle maps (r2c_fft.hpp) and its launcher
#include "engine_types.hpp"
#include "r2c_fft.hpp"

namespace
{
// N = A * B, A the largest divisor up to sqrt(N) (the split of the exact-DFT kernels); false: no compiled length
bool r2c_fft_split(int N, int &A, int &B)
{
  A = 1;
  for (int d = 1; d * d <= N; d++)
    if (N % d == 0)
      A = d;
  B = N / A;
  return A >= 2 && B <= kR2cMaxLen;
}

bool r2c_fft_plan(int N, R2cArgs &a)
{
  if (!r2c_fft_split(N, a.A, a.B))
    return false;
  a.N = N;
  a.H = N / 2 + 1;
  const int raw = a.A * (a.B + 1);
  a.YG = raw + (((a.B - raw) % 32) + 32) % 32;
  for (int G = kR2cThreads / a.B; G >= 1; G--)
  {
    const int Gp = G | 1;
    const int plane = std::max(G * a.YG, N * Gp);
    if (sizeof(double) * (2 * (size_t) N + 2 * (size_t) plane) <= (size_t) kR2cLdsBudget || G == 1)
    {
      a.G = G;
      a.Gp = Gp;
      a.plane = plane;
      return true;
    }
  }
  return false;
}
size_t r2c_fft_lds(const R2cArgs &a) { return sizeof(double) * (2 * (size_t) a.N + 2 * (size_t) a.plane); }
} // namespace

bool bioem_r2c_fft_supported(int N)
{
  R2cArgs a{};
  return N >= 4 && r2c_fft_plan(N, a) && r2c_fft_lds(a) <= 64 * 1024;
}

hipError_t bioem_r2c_fft_launch(hipStream_t st, int nCU, const double *srcD, const float *srcF, const double *tempDen,
                                float NormDen, int N, int nImg, const double2 *tw, double2 *rowSpec, float2 *out, int lo,
                                int side)
{
  R2cArgs a{};
  if (!r2c_fft_plan(N, a))
    return hipErrorNotSupported;
  a.srcD = srcD;
  a.srcF = srcF;
  a.tempden = tempDen;
  a.NormDen = NormDen;
  a.specIn = rowSpec;
  a.specOut = rowSpec;
  a.out = out;
  a.twD = tw;
  a.nImg = nImg;
  a.lo = lo;
  a.side = side;
  const size_t lds = r2c_fft_lds(a);
  const int perCU = std::max(1, std::min(4, (int) (160 * 1024 / (lds + 512))));
  const long rowItems = (long) nImg * ((side + 1) / 2), colItems = (long) nImg * a.H;
  const int rowUnits = (int) ((rowItems + a.G - 1) / a.G), colUnits = (int) ((colItems + a.G - 1) / a.G);
  // three blocks per CU, i.e. at most 168 registers: beside a comparison kernel (three 168-register blocks per CU, the
  // preparation stream at a lower priority) a block must fit the slot ONE retiring comparison block leaves, or the
  // kernel waits for the whole comparison launch to drain (measured with a 191-register variant that fetched its
  // inputs one unit ahead: faster alone, 3 % off the whole job at 1 000 particles)
  // (four blocks of 39 KiB per CU instead of three of 53: the same alone, 4 % off a 20-particle pass at 224^2)
  if (a.B <= 16)
  {
    hipLaunchKernelGGL((k_r2c_fft<true, 16, 3>), dim3(std::min(rowUnits, perCU * nCU)), dim3(kR2cThreads), lds, st, a);
    hipLaunchKernelGGL((k_r2c_fft<false, 16, 3>), dim3(std::min(colUnits, perCU * nCU)), dim3(kR2cThreads), lds, st, a);
  }
  else
  {
    hipLaunchKernelGGL((k_r2c_fft<true, 20, 3>), dim3(std::min(rowUnits, perCU * nCU)), dim3(kR2cThreads), lds, st, a);
    hipLaunchKernelGGL((k_r2c_fft<false, 20, 3>), dim3(std::min(colUnits, perCU * nCU)), dim3(kR2cThreads), lds, st, a);
  }
  return hipGetLastError();
}
