// bioem_hip.hip -- MI355X (gfx950 / CDNA4) implementation of the BioEM compare path behind the
// C ABI of include/bioem_hip.h.  Hand-written HIP, wave64, no vendor FFT/BLAS on the hot path.
//
// Hot path (replaces bioem_cuda::compareRefMaps + cuFFT, /root/reference/bioem_cuda.cu:527-684, and the
// host-side createProjection / createConvolutedProjectionMap, /root/reference/bioem.cpp:1604-1923):
//
//   prep_kernels.hpp     k_project        model points -> real-space projection (double atomics), one launch per batch
//                        k_dft_rows/cols  r2c of the projections (exact DFT, double accumulation; off the critical path)
//                        k_convolve       proj * conj(CTF) -> conv spectra in the comparison layout, sumC, sumsquareC
//                        k_reorder, k_map_sums: particle-side precompute
//   compare_fast.hpp     k_compare_fast   one WAVE per (particle, orientation*CTF) comparison:
//                          spectrum product -> pruned inverse 2-D transform -> displacement-window log posterior
//                          -> wave log-sum-exp/arg-max partial.  The length-N inverse along kx is split as
//                          N = N1*R (R = 32, 16, 8, 4, 2): N1 register-resident R-point FFTs per frequency column
//                          (lane = column, fft_registers.hpp), recombined only for the window rows that are
//                          consumed (output pruning), so no radix-7 butterfly is ever needed for N = 224.  The
//                          transform along ky is a pruned real DFT evaluated from LDS for the window only.
//                        k_nyquist_rows   the Nyquist column of 128^2 / 256^2 by direct summation
//   compare_wide.hpp     k_compare_wide   wide windows: 2 or 4 waves per comparison share the column transforms, one
//                          y-tile of the window per wave (tiles: window_tiles.hpp)
//   compare_rows.hpp     k_compare_oddfft / k_compare_rows  odd image sizes: register FFT of odd length (3..25) over the
//                          reference layout, or direct column sums when N has no factor 3 or 5
//   compare_generic.hpp  k_compare_generic  same maths for odd N / very wide windows (direct pruned DFT)
//   posterior.hpp        calc_logpro / calProb semantics (bioem_algorithm.h:18-142)
//   fold_kernels.hpp     k_fold_wave, k_fold_angles (k_fold: serial variant): fold the per-comparison partials into the probability block in the
//                          reference's (orientation, CTF) order (bioem_algorithm.h:96-123, bioem.cpp:1527-1600)
//   this file            device context, launch logic, the C ABI
//
// Numerics: float expressions that the reference evaluates in float are written in the same order and the
// file is compiled with -ffp-contract=off (FMAs only where fmaf() is spelled out).  Sums that the reference
// accumulates sequentially in float (sumsquareC, particle sums) are accumulated in the same order.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: the library itself is loaded on the first bioem_hip_merge (dlopen)

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "bioem_hip.h"

#define MIN_PROB (-999999.)

namespace
{

struct Partial
{
  double sumExp;
  float best;
  int id;
  float value;
  int pad;
};

typedef bioem_hip_param_device PD;

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
struct HipErr
{
  std::string msg;
};

#define HIP_CHECK(h, expr)                                                                                         \
  do                                                                                                               \
  {                                                                                                                \
    hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess)                                                                                          \
    {                                                                                                              \
      char buf_[512];                                                                                              \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
      (h)->err = buf_;                                                                                             \
      return 1;                                                                                                    \
    }                                                                                                              \
  } while (0)

} // namespace

struct bioem_hip_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  PD pd;
  int nMaps = 0, nAngles = 0, nCTF = 0, algo = 1;
  int N = 0, H = 0, M = 0;
  int fast = 0, N1 = 0, winD = 0; // winD = template window half width used by the fast kernel
  int pchunk = 128;               // particle chunk of the fast kernel's block order (0 = all particles); measured:
                                  // 1 000 particles 6.73 -> 6.56 ms, 10 000 particles (2 GB, beyond the Infinity
                                  // Cache) 78.5 -> 63.2 ms per launch
  int gs = 1;                     // pixels per window row of the fast kernel (gcd of the displacement offsets)
  // wide windows (more than 31 offsets per axis): tilesPerAxis^2 launches of a tileT-row window (window_tiles.hpp)
  int genericWaves = 4; // waves per block of the generic kernel
  bool rowsK = false;   // k_compare_rows / k_compare_oddfft (odd N) instead of the generic kernel
  int oddR = 0;         // k_compare_oddfft: register-FFT length (3, 5, 9, 15, 25) dividing an odd N; 0 = direct sums
  int tileT = 0, tilesPerAxis = 1;
  std::vector<int> tileCenter, tileValid; // per axis tile: centre (in window rows) and number of rows inside the window
  int *dDispLocal = nullptr, *dTileCenter = nullptr, *dTileValid = nullptr, *dRankOfRow = nullptr;
  int wideWPC = 0; // k_compare_wide: waves per comparison (= y-tiles per launch), 0 = one launch per tile
  // k_compare_wide2 (compare_wide2.hpp): wide window in ONE launch per batch -- column transforms shared by the four
  // waves of a comparison, row FFT
  bool wide2 = false;
  bool fastm = false; // 23..31-row windows: k_compare_fastm (window pass on the matrix cores)
  int w2NRW = 0, w2NBLK = 0, w2TS = 0, w2Rows2 = 0, nyqWD = 0, w2Halves = 1, w2NW = 4;
  float2 *dTwk2 = nullptr; // [N1][nd] recombination twiddles exp(2 pi i dx k1 / N), rows in sorted order
  float2 *dConvShift = nullptr;
  Partial *dPartTiles = nullptr;
  bool nyq = false;               // Nyquist column handled outside the 64-column blocks (N/2 a multiple of 64)
  int nd = 0;                     // displacements per axis
  std::vector<int> disp;
  int OB = 0;     // orientations per batch of the native path
  int maxOC = 0;  // capacity of conv/param/partial buffers in (orientation*CTF) units
  int chunkB = 0; // images per DFT chunk

  float2 *dRef = nullptr;
  float *dSumRef = nullptr, *dSumsqRef = nullptr;
  float2 *dCTF = nullptr;
  float *dCtfParam = nullptr;
  bioem_hip_model_point *dPts = nullptr;
  int nPts = 0;
  float NormDen = 0, pixelSize = 0;
  int shiftX = 0, shiftY = 0;
  float4 *dAngles = nullptr;
  int nAnglesUp = 0, isQuat = 1;
  float2 *dTw = nullptr;   // N+1 entries exp(+2 pi i k/N), float
  double2 *dTwD = nullptr; // N entries, double
  int *dDisp = nullptr;
  double2 *dLtab = nullptr;
  float2 *dTwk = nullptr;
  float2 *dTwNyq = nullptr; // [N/2 row pairs][2*winD+1][2] twiddles of the Nyquist pre-kernel (nyq only)
  float *dTnyq = nullptr; // [nMaps][maxOC][2*winD+1] Nyquist-column rows of the current launch (nyq only)

  double *dProjReal = nullptr; // [chunkB][N*N]
  double *dTempDen = nullptr;  // [chunkB]
  double2 *dRowSpec = nullptr; // [chunkB][N][H]
  float2 *dSpecRef = nullptr;  // [chunkB][M] reference layout
  float *dScratch = nullptr;   // [maxOC][M] ordered |X|^2 terms for sumsquareC
  float2 *dConv = nullptr;     // [maxOC][M] comparison layout
  bioem_hip_param5 *dParams = nullptr;
  double2 *dPostC = nullptr;    // [maxOC] {t2, prior} of the log posterior per (orientation, CTF) row (k_posterior_consts)
  Partial *dPartials = nullptr; // [nMaps][maxOC]
  unsigned char *dProb = nullptr;
  size_t probBytes = 0;    // bytes start_run / finish_run move (shard handles: the map entries only)
  size_t devProbBytes = 0; // device block: map entries + the angle table of the owned orientations
  bool shard = false;      // created by bioem_hip_create_shard
  int angO0 = 0, angO1 = 0; // orientations whose angle entries this handle holds ([0, nAngles) unless a shard)
  bioem_hip_angle_candidate *dCand = nullptr; // [nMaps][candK] result of the last top-K selection
  int candK = 0;
  unsigned char *dSend = nullptr, *dRecv = nullptr; // RCCL merge buffers
  size_t sendBytes = 0, recvBytes = 0;
  bioem_hip_prob_map *dMerged = nullptr;

  // second buffer set + stream: projection/convolution of batch k+1 overlap the comparison of batch k
  hipStream_t prepStream = nullptr;
  double *dProjReal2 = nullptr;
  double *dTempDen2 = nullptr;
  double2 *dRowSpec2 = nullptr;
  float2 *dSpecRef2 = nullptr;
  float *dScratch2 = nullptr;
  float2 *dConv2 = nullptr;
  bioem_hip_param5 *dParams2 = nullptr;
  double2 *dPostC2 = nullptr;
  hipEvent_t prepDone[2] = {nullptr, nullptr};
  hipEvent_t cmpDone[2] = {nullptr, nullptr};
  bool cmpPending[2] = {false, false};

  // reference-compatible entry (bioem_hip_compare): the caller hands over a few conv spectra per call (ONE per call
  // in the reference's default ALGO-1 loop, bioem.cpp:534,811-853).  They are staged into a two-half ring -- pinned
  // host rows, H2D copies on their own stream -- and compared with ONE kernel launch per filled half (flush at
  // finish_run), folding in call order through a per-row (orientation, CTF) table.  Half k uses buffer set k of
  // the native pipeline (conv, params, cmpDone[k]).
  struct CompatHalf
  {
    float2 *hConv = nullptr;          // pinned [ringCap][M], reference layout
    bioem_hip_param5 *hPar = nullptr; // pinned [ringCap]
    int2 *hIds = nullptr;             // pinned [ringCap] {orientation, CTF}
    int4 *hSeg = nullptr;             // pinned [ringCap] runs of equal orientation {first row, end row, orientation, 0}
    float2 *dStage = nullptr;         // device [ringCap][M], reference layout
    int2 *dIds = nullptr;
    int4 *dSeg = nullptr;
    hipEvent_t staged = nullptr;      // H2D copies of this half done (copyStream)
  };
  CompatHalf ring[2];
  int ringCap = 0, ringHalf = 0, ringCount = 0;
  std::vector<int> ringOrients; // orientations with rows in the current half, in first-appearance order
  hipStream_t copyStream = nullptr;

  // timing
  std::vector<hipEvent_t> evPool;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evPending;
  double compareMs = 0;
  long long launches = 0, comparisons = 0;

  std::string err;
};

#include "prep_kernels.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#include "compare_wide.hpp"
#include "compare_wide2.hpp"
#include "compare_fastm.hpp"
#include "compare_generic.hpp"
#include "compare_rows.hpp"
#include "fold_kernels.hpp"
#include "window_tiles.hpp"

#ifndef BIOEM_NYQUIST_SPLIT
#define BIOEM_NYQUIST_SPLIT 1
#endif

namespace
{

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
size_t compare_lds_bytes(int N, int H, int NW, int waves)
{ // generic kernel: tables + per-wave T [nd rows][Hs]
  const int Hs = (H + 1) & ~1;
  const size_t dispBytes = ((size_t) NW * 4 + 255) & ~(size_t) 255;
  return (size_t) ((N + 2) & ~1) * 8 + dispBytes + (size_t) waves * NW * Hs * 8;
}

size_t fast_lds_bytes(int N, int NW, int waves, bool half)
{ // fast kernel: twiddles + displacement list + log table + per-wave T block [NW][66] ([NW][34] with half exchange)
  return (size_t) ((N + 2) & ~1) * 8 + 256 + 1024 + (size_t) waves * NW * (half ? 34 : 66) * 8;
}

hipEvent_t get_event(bioem_hip_ctx *h)
{
  if (!h->evPool.empty())
  {
    hipEvent_t e = h->evPool.back();
    h->evPool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess)
    return nullptr;
  return e;
}

void drain_events(bioem_hip_ctx *h)
{
  for (auto &pr : h->evPending)
  {
    float ms = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
      h->compareMs += (double) ms;
    h->evPool.push_back(pr.first);
    h->evPool.push_back(pr.second);
  }
  h->evPending.clear();
}

struct BatchBuf
{
  double *projReal;
  double *tempDen;
  double2 *rowSpec;
  float2 *specRef;
  float *scratch;
  float2 *conv;
  bioem_hip_param5 *params;
  double2 *postc;
};

BatchBuf batch_buf(bioem_hip_ctx *h, int which)
{
  BatchBuf b;
  if (which == 0)
    b = {h->dProjReal, h->dTempDen, h->dRowSpec, h->dSpecRef, h->dScratch, h->dConv, h->dParams, h->dPostC};
  else
    b = {h->dProjReal2, h->dTempDen2, h->dRowSpec2, h->dSpecRef2, h->dScratch2, h->dConv2, h->dParams2, h->dPostC2};
  return b;
}

// the fast-kernel instantiation for a window half width (10 or 15) and register-FFT length (32, 16, 8)
typedef void (*fast_kernel_t)(const CompareArgs);
#ifndef BIOEM_SLIM
template <int WD, int GS>
fast_kernel_t fast_kernel_r(int R, bool nyq)
{
  if (nyq) // N/2 a multiple of 64 implies R = 32
    return k_compare_fast<WD, 32, true, GS>;
  if constexpr (GS == 1)
  { // mixed-radix register FFTs (unit window stride only)
    switch (R)
    {
    case 6: return k_compare_fast<WD, 6, false, 1>;
    case 10: return k_compare_fast<WD, 10, false, 1>;
    case 12: return k_compare_fast<WD, 12, false, 1>;
    case 18: return k_compare_fast<WD, 18, false, 1>;
    case 20: return k_compare_fast<WD, 20, false, 1>;
    case 30: return k_compare_fast<WD, 30, false, 1>;
    default: break;
    }
  }
  return R == 32   ? k_compare_fast<WD, 32, false, GS>
         : R == 16 ? k_compare_fast<WD, 16, false, GS>
         : R == 8  ? k_compare_fast<WD, 8, false, GS>
         : R == 4  ? k_compare_fast<WD, 4, false, GS>
                   : k_compare_fast<WD, 2, false, GS>;
}

template <int WD>
fast_kernel_t fast_kernel_g(int R, bool nyq, int gs)
{
  return gs == 1 ? fast_kernel_r<WD, 1>(R, nyq) : gs == 2 ? fast_kernel_r<WD, 2>(R, nyq)
         : gs == 3 ? fast_kernel_r<WD, 3>(R, nyq) : fast_kernel_r<WD, 4>(R, nyq);
}

// k_compare_wide instantiations: every register-FFT length of the fast kernel, row stride 1/2 (mixed radix: 1), 2 or 4
// waves per comparison
template <int R, bool NYQ>
fast_kernel_t wide_kernel_r(int gs, int wpc)
{
  if (gs == 1)
    return wpc == 2 ? k_compare_wide<R, 1, 2, NYQ> : k_compare_wide<R, 1, 4, NYQ>;
  return wpc == 2 ? k_compare_wide<R, 2, 2, NYQ> : k_compare_wide<R, 2, 4, NYQ>;
}
#endif // BIOEM_SLIM
fast_kernel_t wide_kernel(int R, int gs, int wpc, bool nyq)
{
#ifdef BIOEM_SLIM
  return nullptr;
#else
  if (nyq) // N/2 a multiple of 64 implies R = 32
    return wide_kernel_r<32, true>(gs, wpc);
  switch (R)
  {
  case 32: return wide_kernel_r<32, false>(gs, wpc);
  case 16: return wide_kernel_r<16, false>(gs, wpc);
  case 8: return wide_kernel_r<8, false>(gs, wpc);
  case 4: return wide_kernel_r<4, false>(gs, wpc);
  case 2: return wide_kernel_r<2, false>(gs, wpc);
  // mixed-radix lengths are only chosen with a unit row stride
  case 30: return wpc == 2 ? k_compare_wide<30, 1, 2, false> : k_compare_wide<30, 1, 4, false>;
  case 20: return wpc == 2 ? k_compare_wide<20, 1, 2, false> : k_compare_wide<20, 1, 4, false>;
  case 18: return wpc == 2 ? k_compare_wide<18, 1, 2, false> : k_compare_wide<18, 1, 4, false>;
  case 12: return wpc == 2 ? k_compare_wide<12, 1, 2, false> : k_compare_wide<12, 1, 4, false>;
  case 10: return wpc == 2 ? k_compare_wide<10, 1, 2, false> : k_compare_wide<10, 1, 4, false>;
  default: return wpc == 2 ? k_compare_wide<6, 1, 2, false> : k_compare_wide<6, 1, 4, false>;
  }
#endif
}
size_t wide_lds_bytes(int N, int H, int wpc, bool nyq)
{ // tables + per comparison one T block [21][66] per 64-column block
  const int nblk = nyq ? (H - 1) / 64 : (H + 63) / 64;
  return (size_t) ((N + 2) & ~1) * 8 + 256 + 1024 + (size_t) (4 / wpc) * nblk * 21 * 66 * 8;
}

#ifndef BIOEM_SLIM
// k_compare_wide2 instantiations: every register-FFT length; (rows per wave, column blocks) = (32, 1) or (21, 2)
template <int R>
fast_kernel_t wide2_kernel_r(int nblk, bool nyq, int halves)
{
  if constexpr (R == 32)
  {
    if (nyq && nblk == 3) // 384
      return halves == 2 ? k_compare_wide2<32, 21, 3, true, 2> : k_compare_wide2<32, 21, 3, true>;
    if (nyq)
      return nblk == 1     ? k_compare_wide2<32, 32, 1, true>
             : halves == 2 ? k_compare_wide2<32, 21, 2, true, 2>
                           : k_compare_wide2<32, 21, 2, true>;
  }
  if constexpr (R == 32 || R == 16 || R == 8 || R == 30 || R == 20 || R == 12 || R == 10)
  {
    if (nblk == 3) // 256 < N <= 382
      return halves == 2 ? k_compare_wide2<R, 21, 3, false, 2> : k_compare_wide2<R, 21, 3, false>;
  }
  return nblk == 1 ? k_compare_wide2<R, 32, 1, false> : halves == 2 ? k_compare_wide2<R, 21, 2, false, 2>
                                                                    : k_compare_wide2<R, 21, 2, false>;
}
// windows of 32..52 rows (at most 11 / 13 per wave) over two column blocks with 16- or 8-point register FFTs: 44 / 52 T
// accumulators + a short FFT fit three waves per SIMD, the T block (<= 48 KiB) three blocks per CU
// one column block (N <= 126, or 128 with the Nyquist split), at most 21 rows per wave, 16- / 8-point FFTs: 42 T
// accumulators -> three waves per SIMD as well
fast_kernel_t wide2_kernel_small1(int R, bool nyq)
{
  if (nyq)
    return k_compare_wide2<16, 21, 1, true>;
  switch (R)
  {
  case 8: return k_compare_wide2<8, 21, 1, false>;
  case 12: return k_compare_wide2<12, 21, 1, false>;
  case 10: return k_compare_wide2<10, 21, 1, false>;
  default: return k_compare_wide2<16, 21, 1, false>;
  }
}
template <int NRW>
fast_kernel_t wide2_kernel_small_n(int R, bool nyq)
{
  if (nyq) // N/2 a multiple of 64 (256^2): R = 16 by choice
    return k_compare_wide2<16, NRW, 2, true>;
  switch (R)
  {
  case 8: return k_compare_wide2<8, NRW, 2, false>;
  case 12: return k_compare_wide2<12, NRW, 2, false>;
  case 10: return k_compare_wide2<10, NRW, 2, false>;
  default: return k_compare_wide2<16, NRW, 2, false>;
  }
}
fast_kernel_t wide2_kernel_small(int R, int nrw, bool nyq)
{
  return nrw == 13 ? wide2_kernel_small_n<13>(R, nyq) : wide2_kernel_small_n<11>(R, nyq);
}
// 22..24 rows per wave (windows of 85..96 rows: +-42 ... +-47 px) over two column blocks, 32- / 16-point FFTs
fast_kernel_t wide2_kernel_24(int R, bool nyq, int halves)
{
  if (R == 32 && nyq)
    return halves == 2 ? k_compare_wide2<32, 24, 2, true, 2> : k_compare_wide2<32, 24, 2, true>;
  if (R == 32)
    return halves == 2 ? k_compare_wide2<32, 24, 2, false, 2> : k_compare_wide2<32, 24, 2, false>;
  return halves == 2 ? k_compare_wide2<16, 24, 2, false, 2> : k_compare_wide2<16, 24, 2, false>;
}
// four column blocks (384 < N <= 512) with at most 11 rows per wave (windows of 32..44 rows), 32- / 16-point FFTs
fast_kernel_t wide2_kernel_4(int R, bool nyq, int halves)
{
  if (R == 32 && nyq)
    return halves == 2 ? k_compare_wide2<32, 11, 4, true, 2> : k_compare_wide2<32, 11, 4, true>;
  if (R == 32)
    return halves == 2 ? k_compare_wide2<32, 11, 4, false, 2> : k_compare_wide2<32, 11, 4, false>;
  return halves == 2 ? k_compare_wide2<16, 11, 4, false, 2> : k_compare_wide2<16, 11, 4, false>;
}
fast_kernel_t wide2_kernel(int R, int nblk, bool nyq, int halves = 1, int nrw = 21)
{
  if (nblk == 4)
    return wide2_kernel_4(R, nyq, halves);
  if (nrw == 24)
    return wide2_kernel_24(R, nyq, halves);
  switch (R)
  {
  case 32: return wide2_kernel_r<32>(nblk, nyq, halves);
  case 16: return wide2_kernel_r<16>(nblk, nyq, halves);
  case 8: return wide2_kernel_r<8>(nblk, nyq, halves);
  case 4: return wide2_kernel_r<4>(nblk, nyq, halves);
  case 2: return wide2_kernel_r<2>(nblk, nyq, halves);
  case 30: return wide2_kernel_r<30>(nblk, nyq, halves);
  case 20: return wide2_kernel_r<20>(nblk, nyq, halves);
  case 18: return wide2_kernel_r<18>(nblk, nyq, halves);
  case 12: return wide2_kernel_r<12>(nblk, nyq, halves);
  case 10: return wide2_kernel_r<10>(nblk, nyq, halves);
  default: return wide2_kernel_r<6>(nblk, nyq, halves);
  }
}
// the instantiation for a selection (create sets the attributes of the SAME function the launch uses)
// eight waves per comparison: 16- / 12- / 10- / 8-point FFTs, 11 rows per wave, two column blocks, whole T block
fast_kernel_t wide2_kernel_w8(int R, bool nyq)
{
  if (nyq)
    return k_compare_wide2<16, 11, 2, true, 1, 8>;
  switch (R)
  {
  case 12: return k_compare_wide2<12, 11, 2, false, 1, 8>;
  case 10: return k_compare_wide2<10, 11, 2, false, 1, 8>;
  case 8: return k_compare_wide2<8, 11, 2, false, 1, 8>;
  default: return k_compare_wide2<16, 11, 2, false, 1, 8>;
  }
}
// ... over four column blocks (384 < N <= 512), windows of 45..88 rows: one 512-thread block per CU
fast_kernel_t wide2_kernel_w8_4(int R, bool nyq, int halves)
{
  if (R == 32 && nyq)
    return halves == 2 ? k_compare_wide2<32, 11, 4, true, 2, 8> : k_compare_wide2<32, 11, 4, true, 1, 8>;
  if (R == 32)
    return halves == 2 ? k_compare_wide2<32, 11, 4, false, 2, 8> : k_compare_wide2<32, 11, 4, false, 1, 8>;
  return halves == 2 ? k_compare_wide2<16, 11, 4, false, 2, 8> : k_compare_wide2<16, 11, 4, false, 1, 8>;
}
#endif // BIOEM_SLIM
fast_kernel_t wide2_pick(int R, int nrw, int nblk, bool nyq, int halves, int nw = 4)
{
#ifdef BIOEM_SLIM
#ifdef BIOEM_SLIM_W2
  return k_compare_wide2<BIOEM_SLIM_W2>;
#else
  return nullptr;
#endif
#else
  if (nw == 8 && nblk == 4)
    return wide2_kernel_w8_4(R, nyq, halves);
  if (nw == 8)
    return wide2_kernel_w8(R, nyq);
  if (nblk == 1 && nrw == 21)
    return wide2_kernel_small1(R, nyq);
  if (nblk == 2 && nrw <= 13)
    return wide2_kernel_small(R, nrw, nyq);
  return wide2_kernel(R, nblk, nyq, halves, nrw);
#endif
}
size_t wide2_lds_bytes(int N, int R, int rows2, int ts, int nw = 4)
{ // tables (twiddles, visiting ranks, log table, wave results, posterior constants) + max(one FFT-output slot per wave, T block)
  const size_t slots = (size_t) nw * R * 64 * 8, tblock = (size_t) rows2 * ts * 8;
  return (size_t) ((N + 2) & ~1) * 8 + 512 + 1024 + 256 + std::max(slots, tblock);
}

// k_compare_fastm instantiations: 27- and 31-row windows, register-FFT lengths up to 16 (three waves per SIMD), row strides
// 1..4 for the power-of-two lengths, Nyquist split with 16 points
size_t fastm_lds_bytes(int N)
{ // cos / sin planes of the twiddle table (padded), window ranks, log table, per wave two 32 x 33 float planes and the
  // resting place of the 16 tile accumulators
  return (size_t) 2 * fastm_table_floats(N) * 4 + 128 + 1024 + (size_t) 4 * 2 * 32 * 33 * 4 + (size_t) 4 * 16 * 64 * 4;
}
#ifndef BIOEM_SLIM
template <int WD, int GS>
fast_kernel_t fastm_kernel_r(int R, bool nyq)
{
  if (nyq)
    return k_compare_fastm<WD, 16, true, GS>;
  if constexpr (GS == 1)
  {
    switch (R)
    {
    case 6: return k_compare_fastm<WD, 6, false, 1>;
    case 10: return k_compare_fastm<WD, 10, false, 1>;
    case 12: return k_compare_fastm<WD, 12, false, 1>;
    default: break;
    }
  }
  return R == 16 ? k_compare_fastm<WD, 16, false, GS>
         : R == 8 ? k_compare_fastm<WD, 8, false, GS>
         : R == 4 ? k_compare_fastm<WD, 4, false, GS>
                  : k_compare_fastm<WD, 2, false, GS>;
}
template <int WD>
fast_kernel_t fastm_kernel_g(int R, bool nyq, int gs)
{
  return gs == 1 ? fastm_kernel_r<WD, 1>(R, nyq) : gs == 2 ? fastm_kernel_r<WD, 2>(R, nyq)
         : gs == 3 ? fastm_kernel_r<WD, 3>(R, nyq) : fastm_kernel_r<WD, 4>(R, nyq);
}
#endif // BIOEM_SLIM
fast_kernel_t fastm_kernel(int winD, int R, bool nyq, int gs)
{
#ifdef BIOEM_SLIM
#ifdef BIOEM_SLIM_FASTM
  return k_compare_fastm<BIOEM_SLIM_FASTM>;
#else
  return nullptr;
#endif
#else
  return winD == 13 ? fastm_kernel_g<13>(R, nyq, gs) : fastm_kernel_g<15>(R, nyq, gs);
#endif
}

#ifndef BIOEM_SLIM
template <int WD>
fast_kernel_t rows_kernel_g(int gs)
{
  return gs == 1 ? k_compare_rows<WD, 1> : gs == 2 ? k_compare_rows<WD, 2> : gs == 3 ? k_compare_rows<WD, 3>
                                                                                      : k_compare_rows<WD, 4>;
}
template <int WD>
fast_kernel_t oddfft_kernel_r(int R)
{
  return R == 25 ? k_compare_oddfft<WD, 25> : R == 15 ? k_compare_oddfft<WD, 15> : R == 9 ? k_compare_oddfft<WD, 9>
         : R == 5 ? k_compare_oddfft<WD, 5> : k_compare_oddfft<WD, 3>;
}
#endif // BIOEM_SLIM
// odd N: register FFT of odd length R over the reference layout if R > 0 (unit row stride), else direct column sums
fast_kernel_t rows_kernel(int winD, int gs, int oddR = 0)
{
#ifdef BIOEM_SLIM
  return nullptr;
#else
  if (oddR)
    return winD == 5 ? oddfft_kernel_r<5>(oddR) : winD == 10 ? oddfft_kernel_r<10>(oddR)
           : winD == 13 ? oddfft_kernel_r<13>(oddR) : oddfft_kernel_r<15>(oddR);
  return winD == 5 ? rows_kernel_g<5>(gs) : winD == 10 ? rows_kernel_g<10>(gs) : winD == 13 ? rows_kernel_g<13>(gs)
                                                                                             : rows_kernel_g<15>(gs);
#endif
}

fast_kernel_t fast_kernel(int winD, int R, bool nyq, int gs)
{
#ifdef BIOEM_SLIM
  // experiment builds (scripts/slim_build.sh, never shipped): only the instantiations named on the command line are
  // compiled -- seconds instead of minutes; a shape that selects anything else gets a null kernel and fails the launch
#ifdef BIOEM_SLIM_FAST
  return k_compare_fast<BIOEM_SLIM_FAST>;
#else
  return nullptr;
#endif
#else
  // (27- and 31-row windows run k_compare_fastm)
  return winD == 5 ? fast_kernel_g<5>(R, nyq, gs) : fast_kernel_g<10>(R, nyq, gs);
#endif
}

// ids == nullptr: row oc of the launch is (orient0 + oc / convPerOrient, conv0 + oc % convPerOrient) (native path);
// otherwise ids[oc] = {orientation, CTF} and segs[0..nSeg) = runs of equal orientation (compat ring)
int launch_compare_fold(bioem_hip_ctx *h, const BatchBuf &bb, int nOC, int orient0, int conv0, int convPerOrient,
                        const int2 *ids = nullptr, const int4 *segs = nullptr, int nSeg = 0)
{
  CompareArgs a;
  a.ref = h->dRef;
  a.conv = bb.conv;
  a.params = bb.params;
  a.postc = bb.postc;
  a.sumRef = h->dSumRef;
  a.sumsqRef = h->dSumsqRef;
  a.tw = h->dTw;
  a.disp = h->dDisp;
  a.ltab = h->dLtab;
  a.twk = h->dTwk;
  a.tnyq = h->dTnyq;
  a.twnyq = h->dTwNyq;
  a.partials = h->dPartials;
  a.ldPart = h->maxOC;
  a.N = h->N;
  a.H = h->H;
  a.N1 = h->N1;
  a.nd = h->nd;
  a.maxD = h->pd.maxDisplaceCenter;
  a.nOC = nOC;
  a.nMaps = h->nMaps;
  a.algo = h->algo;
  a.pd = h->pd;
  a.gs = h->gs;
  a.ndx = a.ndy = h->nd;
  a.pchunk = h->pchunk > 0 ? std::min(h->pchunk, h->nMaps) : h->nMaps;
  const int ocGroups = (nOC + 3) / 4;
  // few particles: whole groups per XCD (fast_block_pair); the grid is padded to a multiple of 8 groups
  const bool groupPerXcd = h->nMaps <= 64 && h->fast && !h->wide2 && !(h->tileT && h->wideWPC) &&
                           !getenv("BIOEM_NO_GROUP_XCD");
  if (groupPerXcd)
    a.pchunk = -1;
  const dim3 grid((unsigned) ((size_t) (groupPerXcd ? (ocGroups + 7) / 8 * 8 : ocGroups) * h->nMaps));
  hipEvent_t e0 = get_event(h), e1 = get_event(h);
  if (!e0 || !e1)
  {
    if (e0)
      h->evPool.push_back(e0);
    if (e1)
      h->evPool.push_back(e1);
    h->err = "hipEventCreate failed";
    return 1;
  }
  // {t2, prior} of every row of the launch, once per row instead of once per comparison and lane
  hipLaunchKernelGGL(k_posterior_consts, dim3((nOC + 63) / 64), dim3(64), 0, h->stream, bb.params, h->pd, bb.postc, nOC);
  HIP_CHECK(h, hipEventRecord(e0, h->stream));
  if (h->wide2)
  {
    CompareArgs aw = a;
    aw.twk = h->dTwk2;
    aw.ts = h->w2TS;
    aw.nyqWD = h->nyqWD;
    if (h->nyq)
    {
      const dim3 gridq((unsigned) (((size_t) (h->nMaps + 15) / 16) * ((nOC + 15) / 16)));
      if (h->nyqWD == 20)
        hipLaunchKernelGGL(k_nyquist_rows<20>, gridq, dim3(256), 0, h->stream, aw);
      else if (h->nyqWD == 31)
        hipLaunchKernelGGL(k_nyquist_rows<31>, gridq, dim3(256), 0, h->stream, aw);
      else
        hipLaunchKernelGGL(k_nyquist_rows<42>, gridq, dim3(256), 0, h->stream, aw);
    }
    const size_t lds = wide2_lds_bytes(h->N, 2 * h->fast, h->w2Rows2, h->w2TS, h->w2NW);
    hipLaunchKernelGGL(wide2_pick(2 * h->fast, h->w2NRW, h->w2NBLK, h->nyq, h->w2Halves, h->w2NW),
                       dim3((unsigned) ((size_t) nOC * h->nMaps)), dim3(64 * h->w2NW), lds, h->stream, aw);
  }
  else if (h->fast || h->rowsK)
  {
    const int NW = 2 * h->winD + 1;
    const size_t lds = h->fastm ? fastm_lds_bytes(h->N)
                                : fast_lds_bytes(h->N, NW, 4, h->fast ? fast_half_t(h->winD, 2 * h->fast) : false);
    auto launch_window = [&](const CompareArgs &aw) {
      if (h->rowsK)
      {
        hipLaunchKernelGGL(rows_kernel(h->winD, h->gs, h->oddR), grid, dim3(256), lds, h->stream, aw);
        return;
      }
      if (h->nyq)
      {
        const dim3 gridq((unsigned) (((size_t) (h->nMaps + 15) / 16) * ((nOC + 15) / 16)));
        if (h->winD == 5)
          hipLaunchKernelGGL(k_nyquist_rows<5>, gridq, dim3(256), 0, h->stream, aw);
        else if (h->winD == 10)
          hipLaunchKernelGGL(k_nyquist_rows<10>, gridq, dim3(256), 0, h->stream, aw);
        else if (h->winD == 13)
          hipLaunchKernelGGL(k_nyquist_rows<13>, gridq, dim3(256), 0, h->stream, aw);
        else
          hipLaunchKernelGGL(k_nyquist_rows<15>, gridq, dim3(256), 0, h->stream, aw);
      }
      hipLaunchKernelGGL(h->fastm ? fastm_kernel(h->winD, 2 * h->fast, h->nyq, h->gs)
                                  : fast_kernel(h->winD, 2 * h->fast, h->nyq, h->gs),
                         grid, dim3(256), lds, h->stream, aw);
    };
    if (!h->tileT)
      launch_window(a);
    else
    { // wide window: one launch per tile on the phase-shifted conv spectra, then merge (window_tiles.hpp)
      const int nT = h->tilesPerAxis;
      const size_t tileStride = (size_t) h->nMaps * h->maxOC;
      const size_t total = (size_t) nOC * h->M;
      CompareArgs at = a;
      at.disp = h->dDispLocal;
      at.nd = h->tileT;
      at.maxD = h->winD * h->gs;
      if (h->wideWPC)
      { // per x-tile: conv shifted in x only, then all y-tiles in groups of wideWPC inside k_compare_wide
        const int wpc = h->wideWPC, cpb = 4 / wpc;
        const dim3 gridw((unsigned) ((size_t) ((nOC + cpb - 1) / cpb) * h->nMaps));
        const size_t ldsw = wide_lds_bytes(h->N, h->H, wpc, h->nyq);
        at.nTiles = nT;
        at.tileCenter = h->dTileCenter;
        at.tileValid = h->dTileValid;
        at.tileStride = tileStride;
        for (int tx = 0; tx < nT; tx++)
        {
          const int sx = h->gs * h->tileCenter[tx];
          if (sx == 0)
            at.conv = bb.conv;
          else
          {
            hipLaunchKernelGGL(k_phase_shift, dim3(2048), dim3(256), 0, h->stream, bb.conv, h->dConvShift, total, h->N,
                               h->H, h->fast, h->N1, sx, 0, h->dTw);
            at.conv = h->dConvShift;
          }
          at.ndx = h->tileValid[tx];
          at.partials = h->dPartTiles + (size_t) (tx * nT) * tileStride;
          if (h->nyq)
          { // Nyquist-column rows of this x-tile (they do not depend on the y-tile)
            const dim3 gridq((unsigned) (((size_t) (h->nMaps + 15) / 16) * ((nOC + 15) / 16)));
            hipLaunchKernelGGL(k_nyquist_rows<10>, gridq, dim3(256), 0, h->stream, at);
          }
          for (int y0 = 0; y0 < nT; y0 += wpc)
          {
            at.yTile0 = y0;
            hipLaunchKernelGGL(wide_kernel(2 * h->fast, h->gs, wpc, h->nyq), gridw, dim3(256), ldsw, h->stream, at);
          }
        }
      }
      else
      for (int tx = 0; tx < nT; tx++)
        for (int ty = 0; ty < nT; ty++)
        {
          const int sx = h->gs * h->tileCenter[tx], sy = h->gs * h->tileCenter[ty];
          if (sx == 0 && sy == 0)
            at.conv = bb.conv;
          else
          {
            hipLaunchKernelGGL(k_phase_shift, dim3(2048), dim3(256), 0, h->stream, bb.conv, h->dConvShift, total, h->N,
                               h->H, h->fast, h->N1, sx, sy, h->dTw);
            at.conv = h->dConvShift;
          }
          at.ndx = h->tileValid[tx];
          at.ndy = h->tileValid[ty];
          at.partials = h->dPartTiles + (size_t) (tx * nT + ty) * tileStride;
          launch_window(at);
        }
      const long long nt = (long long) nOC * h->nMaps;
      hipLaunchKernelGGL(k_merge_tiles, dim3((unsigned) ((nt + 255) / 256)), dim3(256), 0, h->stream, h->dPartTiles,
                         nT * nT, tileStride, h->maxOC, nOC, h->nMaps, h->tileT, nT, h->dTileCenter,
                         h->pd.maxDisplaceCenter / h->gs, h->nd, h->dRankOfRow, h->dPartials);
    }
  }
  else
  {
    const int gw = h->genericWaves;
    const size_t lds = compare_lds_bytes(h->N, h->H, h->nd, gw);
    const dim3 gridg((unsigned) ((size_t) ((nOC + gw - 1) / gw) * h->nMaps));
    hipLaunchKernelGGL(k_compare_generic, gridg, dim3(64 * gw), lds, h->stream, a);
  }
  HIP_CHECK(h, hipGetLastError());
  HIP_CHECK(h, hipEventRecord(e1, h->stream));
  h->evPending.push_back({e0, e1});
  h->launches++;
  h->comparisons += (long long) nOC * h->nMaps;
  bioem_hip_prob_map *pmap = reinterpret_cast<bioem_hip_prob_map *>(h->dProb);
  bioem_hip_prob_angle *pang = reinterpret_cast<bioem_hip_prob_angle *>(h->dProb + sizeof(bioem_hip_prob_map) * h->nMaps);
  static const bool serialFold = getenv("BIOEM_SERIAL_FOLD") != nullptr; // debugging: the one-thread-per-particle fold
  if (serialFold)
    hipLaunchKernelGGL(k_fold, dim3((h->nMaps + 127) / 128), dim3(128), 0, h->stream, h->dPartials, h->maxOC, nOC,
                       h->nMaps, bb.params, h->dSumRef, h->dDisp, h->nd, h->pd, orient0, conv0, convPerOrient, ids,
                       pmap, pang, h->angO0);
  else
  {
    if (h->pd.writeAngles)
    {
      const int nRuns = ids ? nSeg : (nOC + convPerOrient - 1) / convPerOrient;
      const long long nt = (long long) nRuns * h->nMaps;
      hipLaunchKernelGGL(k_fold_angles, dim3((unsigned) ((nt + 255) / 256)), dim3(256), 0, h->stream, h->dPartials,
                         h->maxOC, nOC, h->nMaps, orient0, convPerOrient, segs, nRuns, pang, h->angO0);
    }
    hipLaunchKernelGGL(k_fold_wave, dim3((h->nMaps + 3) / 4), dim3(256), 0, h->stream, h->dPartials, h->maxOC, nOC,
                       h->nMaps, bb.params, h->dSumRef, h->dDisp, h->nd, h->pd, orient0, conv0, convPerOrient, ids,
                       pmap);
  }
  HIP_CHECK(h, hipGetLastError());
  if (h->evPending.size() > 512)
    drain_events(h);
  return 0;
}

// the exact-DFT r2c kernels keep one image row / column (and its twiddles) in dynamic LDS: 24 N + 16 and 32 N bytes.
// Above 64 KiB a launch needs the explicit opt-in; 32 N <= 160 KiB bounds the image size at 5120 pixels (the
// reference's MRC reader stops at 5000, mrc.h:128-133).
const int kMaxPixels = 5120;
hipError_t dft_allow_lds(int N)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dft_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int) (sizeof(double) * (3 * (size_t) N + 2)));
  if (e != hipSuccess)
    return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k_dft_cols), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int) (sizeof(double2) * 2 * (size_t) N));
}

void dft_split(int N, int &A, int &B)
{
  A = 1;
  for (int d = 1; d * d <= N; d++)
    if (N % d == 0)
      A = d;
  B = N / A;
}

// r2c of nImg images (either double projection maps with tempden scaling, or float maps) into bb.specRef
int run_r2c(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, const double *srcD, const float *srcF, int nImg)
{
  const int N = h->N, H = h->H;
  int A, B;
  dft_split(N, A, B);
  hipLaunchKernelGGL(k_dft_rows, dim3(N, nImg), dim3(128), sizeof(double) * (3 * N + 2), st, srcD, srcF, bb.tempDen,
                     h->NormDen, N, H, A, B, h->dTwD, bb.rowSpec);
  HIP_CHECK(h, hipGetLastError());
  hipLaunchKernelGGL(k_dft_cols, dim3(H, nImg), dim3(256), sizeof(double2) * 2 * N, st, bb.rowSpec, N, H, A, B,
                     h->dTwD, bb.specRef);
  HIP_CHECK(h, hipGetLastError());
  return 0;
}

int project_batch(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, int o0, int nO)
{
  const int N = h->N;
  HIP_CHECK(h, hipMemsetAsync(bb.projReal, 0, sizeof(double) * (size_t) nO * N * N, st));
  HIP_CHECK(h, hipMemsetAsync(bb.tempDen, 0, sizeof(double) * nO, st));
  hipLaunchKernelGGL(k_project, dim3((h->nPts + 255) / 256, nO), dim3(256), 0, st, h->dPts, h->nPts, h->dAngles, o0,
                     h->isQuat, N, h->pixelSize, h->shiftX, h->shiftY, bb.projReal, bb.tempDen);
  HIP_CHECK(h, hipGetLastError());
  return run_r2c(h, bb, st, bb.projReal, nullptr, nO);
}

// ---- compat ring (bioem_hip_compare) ----
int compat_alloc(bioem_hip_ctx *h)
{
  if (h->ringCap)
    return 0;
  int cap = 128; // rows per half: 128 x 1 000 particles = 2.6 ms of comparison kernel at 224^2
  if (getenv("BIOEM_COMPAT_RING"))
    cap = atoi(getenv("BIOEM_COMPAT_RING"));
  cap = std::max(1, std::min(cap, h->maxOC));
  const size_t M = (size_t) h->M;
  for (int k = 0; k < 2; k++)
  {
    bioem_hip_ctx::CompatHalf &r = h->ring[k];
    HIP_CHECK(h, hipHostMalloc(&r.hConv, sizeof(float2) * M * cap, hipHostMallocDefault));
    HIP_CHECK(h, hipHostMalloc(&r.hPar, sizeof(bioem_hip_param5) * cap, hipHostMallocDefault));
    HIP_CHECK(h, hipHostMalloc(&r.hIds, sizeof(int2) * cap, hipHostMallocDefault));
    HIP_CHECK(h, hipHostMalloc(&r.hSeg, sizeof(int4) * cap, hipHostMallocDefault));
    HIP_CHECK(h, hipMalloc(&r.dStage, sizeof(float2) * M * cap));
    HIP_CHECK(h, hipMalloc(&r.dIds, sizeof(int2) * cap));
    HIP_CHECK(h, hipMalloc(&r.dSeg, sizeof(int4) * cap));
    HIP_CHECK(h, hipEventCreateWithFlags(&r.staged, hipEventDisableTiming));
  }
  HIP_CHECK(h, hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking));
  h->ringCap = cap;
  h->ringHalf = 0;
  h->ringCount = 0;
  return 0;
}

void compat_free(bioem_hip_ctx *h)
{
  for (int k = 0; k < 2; k++)
  {
    bioem_hip_ctx::CompatHalf &r = h->ring[k];
    if (r.hConv)
      hipHostFree(r.hConv);
    if (r.hPar)
      hipHostFree(r.hPar);
    if (r.hIds)
      hipHostFree(r.hIds);
    if (r.hSeg)
      hipHostFree(r.hSeg);
    if (r.dStage)
      hipFree(r.dStage);
    if (r.dIds)
      hipFree(r.dIds);
    if (r.dSeg)
      hipFree(r.dSeg);
    if (r.staged)
      hipEventDestroy(r.staged);
    r = bioem_hip_ctx::CompatHalf();
  }
  if (h->copyStream)
    hipStreamDestroy(h->copyStream);
  h->copyStream = nullptr;
  h->ringCap = h->ringCount = 0;
}

// launch the comparison of the rows staged in the current half, then switch halves
int compat_flush(bioem_hip_ctx *h)
{
  const int n = h->ringCount;
  if (n == 0)
    return 0;
  const int half = h->ringHalf;
  bioem_hip_ctx::CompatHalf &r = h->ring[half];
  const BatchBuf bb = batch_buf(h, half);
  // runs of equal orientation (rows arrive in call order; an orientation never returns within a half, see compare)
  int nSeg = 0;
  for (int i = 0; i < n; i++)
  {
    if (nSeg && r.hSeg[nSeg - 1].z == r.hIds[i].x)
      r.hSeg[nSeg - 1].y = i + 1;
    else
      r.hSeg[nSeg++] = make_int4(i, i + 1, r.hIds[i].x, 0);
  }
  HIP_CHECK(h, hipEventRecord(r.staged, h->copyStream));
  HIP_CHECK(h, hipStreamWaitEvent(h->stream, r.staged, 0));
  HIP_CHECK(h, hipMemcpyAsync(bb.params, r.hPar, sizeof(bioem_hip_param5) * n, hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipMemcpyAsync(r.dIds, r.hIds, sizeof(int2) * n, hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipMemcpyAsync(r.dSeg, r.hSeg, sizeof(int4) * nSeg, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_reorder, dim3(std::min(2048, 8 * n)), dim3(256), 0, h->stream, r.dStage, bb.conv, n, h->N, h->H,
                     h->fast, h->N1);
  HIP_CHECK(h, hipGetLastError());
  if (launch_compare_fold(h, bb, n, 0, 0, 1, r.dIds, r.dSeg, nSeg))
    return 1;
  HIP_CHECK(h, hipEventRecord(h->cmpDone[half], h->stream));
  h->cmpPending[half] = true;
  h->ringHalf ^= 1;
  h->ringCount = 0;
  h->ringOrients.clear();
  return 0;
}

// conv spectra of CTFs [c0, c0 + nC) of the nO projected orientations, row ob * nC + (c - c0)
int convolve_batch(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, int nO, int c0, int nC)
{
  const int M4 = (int) ((h->M + 3) & ~(size_t) 3);
  hipLaunchKernelGGL(k_convolve, dim3(nC, nO), dim3(256), 0, st, bb.specRef, h->dCTF, h->dCtfParam, h->N, h->H,
                     h->fast, h->N1, c0, bb.conv, bb.scratch, M4, bb.params);
  HIP_CHECK(h, hipGetLastError());
  // the ordered Parseval sums of all nC x nO spectra side by side: four sequential chains per wave
  hipLaunchKernelGGL(k_parseval_ordered, dim3((nC * nO + 3) / 4), dim3(64), 0, st, bb.scratch, (int) h->M, M4, nC * nO,
                     (float) (h->N * h->N), bb.params);
  HIP_CHECK(h, hipGetLastError());
  return 0;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int bioem_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

size_t bioem_hip_prob_size(int nMaps, int nAngles, int writeAngles)
{
  size_t size = sizeof(bioem_hip_prob_map);
  if (writeAngles)
    size += (size_t) nAngles * sizeof(bioem_hip_prob_angle);
  return (size_t) nMaps * size;
}

const char *bioem_hip_last_error(bioem_hip_handle h) { return h ? h->err.c_str() : "null handle"; }

static int create_impl(bioem_hip_handle *out, int device, const bioem_hip_param_device *pd, int nMaps, int nAngles,
                       int nCTF, int algo, bool shard, int angO0, int angO1)
{
  if (!out || !pd)
    return 2;
  bioem_hip_ctx *h = new bioem_hip_ctx;
  *out = h;
  h->shard = shard;
  h->angO0 = angO0;
  h->angO1 = angO1;
  h->device = device;
  h->pd = *pd;
  h->nMaps = nMaps;
  h->nAngles = nAngles;
  h->nCTF = nCTF;
  h->algo = algo;
  const int N = pd->NumberPixels;
  h->N = N;
  h->H = N / 2 + 1;
  h->M = N * h->H;
  if (N < 2 || nMaps < 1 || nAngles < 1 || nCTF < 1 || pd->maxDisplaceCenter < 0 || pd->GridSpaceCenter < 1 ||
      pd->maxDisplaceCenter >= N / 2)
  {
    h->err = "invalid configuration (need N>=2, nMaps,nAngles,nCTF>=1, 0<=maxD<N/2, grid>=1)";
    return 2;
  }
  if (angO0 < 0 || angO1 > nAngles || angO0 >= angO1)
  {
    h->err = "invalid configuration: orientation range of the shard must be a non-empty part of [0, nAngles)";
    return 2;
  }
  if (N > kMaxPixels)
  {
    h->err = "invalid configuration: images larger than 5120 x 5120 pixels are not supported";
    return 2;
  }
  HIP_CHECK(h, hipSetDevice(device));
  HIP_CHECK(h, dft_allow_lds(N));
  {
    int prLow = 0, prHigh = 0;
    HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    HIP_CHECK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prHigh));
  }

  // displacement list per axis in the reference's visiting order
  const int maxD = pd->maxDisplaceCenter, g = pd->GridSpaceCenter;
  if (algo == 1)
  { // bioem_algorithm.h:156-197
    for (int c = 0; c <= maxD; c += g)
      h->disp.push_back(c);
    for (int c = N - maxD; c < N; c += g)
      h->disp.push_back(c - N);
  }
  else
  { // bioem.cpp:1477-1485
    const int NxDisp = 2 * (maxD / g) + 1;
    for (int m = 0; m < NxDisp; m++)
      h->disp.push_back(m * g - maxD);
  }
  h->nd = (int) h->disp.size();

  // fast path: N = N1 * R, R = register-FFT length; h->fast holds R/2 (rows per k1 step)
  // window rows: row m holds displacement m * gs, gs = gcd of all offsets (1..4 are instantiated), so a coarse grid
  // with maxD a multiple of the spacing reaches +-15*gs pixels
  {
    int gg = 0;
    for (int d : h->disp)
    {
      int x = d < 0 ? -d : d, y = gg;
      while (y)
      {
        const int t = x % y;
        x = y;
        y = t;
      }
      gg = x;
    }
    h->gs = (gg >= 1 && gg <= 4) ? gg : 1;
  }
  // window template: 2*winD+1 rows, nd <= rows (ALGO 1 with maxD % grid != 0 visits up to 2*(maxD/grid)+2 offsets)
  const int mD = maxD / h->gs;
  h->winD = (mD <= 5 && h->nd <= 11) ? 5 : (mD <= 10 && h->nd <= 21) ? 10 : (mD <= 13 && h->nd <= 27) ? 13 : 15;
  // wide windows: more rows than the 31-row template -> tiles of 21 or 31 rows (window_tiles.hpp); needs the plain
  // symmetric set {gs*m, |m| <= mD}
  h->tileT = 0;
  h->tilesPerAxis = 1;
  // wide windows, first choice: k_compare_wide2 (one launch per batch, shared column transforms, row FFT).  Needs an
  // even image size with at most two 64-column blocks, at most 32 (one block) / 21 (two blocks) window rows per wave,
  // and its T block [rows][H] in LDS
  // ... and for the 23..31-row windows where the 27/31-row templates of k_compare_fast are weak: the sizes that keep
  // 32-point FFTs there (Nyquist split: 128^2 +-15 px 47.7 -> 54.8 M/s, 256^2 20.8 -> 24.0) and images whose second
  // column block is mostly empty (160^2 +-15 px 36.1 -> 43.9, 192^2 33.8 -> 38.1); at 200^2 / 224^2 / 240^2 the two tie
  // (31.5 / 33.6 / 29.8 vs 29.8 / 32.6 / 29.7) and at 96^2 the template wins (73.1 vs 67.8): those keep it
  const bool nyqSize0 = BIOEM_NYQUIST_SPLIT && N % 2 == 0 && (N / 2) % 64 == 0;
  // (k_compare_fastm took these windows over; BIOEM_MID_WIDE2 brings the old choice back for A/B runs)
  const bool midWindow = h->nd > 21 && h->nd <= 31 && (nyqSize0 || (h->H > 64 && h->H <= 100)) &&
                         getenv("BIOEM_MID_WIDE2");
  if (N % 2 == 0 && N >= 8 && (mD > 15 || h->nd > 31 || midWindow || (getenv("BIOEM_FORCE_WIDE2") && h->nd >= 21)) &&
      h->nd == 2 * mD + 1 && !getenv("BIOEM_NO_WIDE2"))
  {
    int R = (N % 32 == 0) ? 32 : (N % 16 == 0) ? 16 : (N % 8 == 0) ? 8 : (N % 4 == 0) ? 4 : 2;
    if (R < 8 && !getenv("BIOEM_POW2_FFT"))
    {
      static const int mixed[] = {30, 20, 18, 12, 10, 6};
      for (int r : mixed)
        if (N % r == 0 && r > R)
        {
          R = r;
          break;
        }
    }
    if (const char *fr = getenv("BIOEM_W2_R"))
    { // experiments: a given register-FFT length where it divides N
      const int r = atoi(fr);
      static const int lens[] = {32, 16, 8, 4, 2, 30, 20, 18, 12, 10, 6};
      for (int l : lens)
        if (l == r && N % r == 0)
          R = r;
    }
    const bool nyq = BIOEM_NYQUIST_SPLIT && (N / 2) % 64 == 0;
    const int nblk = nyq ? (h->H - 1) / 64 : (h->H + 63) / 64;
    const int rpw = (h->nd + 3) / 4;
    // the longest length with a three-waves-per-SIMD instantiation (16, 12, 10, 8) that divides N
    int r3 = 0;
    if (!getenv("BIOEM_W2_R"))
    {
      static const int lens3[] = {16, 12, 10, 8};
      for (int l : lens3)
        if (!r3 && N % l == 0 && (l == 16 || !nyq))
          r3 = l;
    }
    else if (R == 16 || ((R == 12 || R == 10 || R == 8) && !nyq))
      r3 = R;
    const bool mixedLen = R != 32 && R != 16 && R != 8 && R != 4 && R != 2;
    const int rows2 = 2 * ((h->nd + 1) / 2);
    int ts = h->H; // row stride = 4 mod 16 float2: the (row pair, k1) lanes of the row pass spread over the banks
    while (ts % 16 != 4)
      ts++;
    // small variant (see wide2_kernel_small): measured at 224^2 against the tiled kernel / the 31-row template / the
    // two-wave instantiation: +-20 px 24.3 vs 20.7 M/s, +-24 px 22.7 vs 17.7, +-15 px 32.8 vs 31.7, +-12 px 33.5 vs 34.6
    // -> used from 32 rows on
    // (three blocks per CU must fit: at 256^2 the 13-row variant does not -- +-24 px 11.8 vs 14.9 M/s for the two-wave
    // instantiation -- the 11-row one does: +-16 px 17.4 -> 23.2, +-20 px 17.4 -> 17.7)
    // Mixed-radix sizes take it with a 12- or 10-point FFT (`scripts/w2_length_sweep.sh`: 180^2 +-20 px 21.0 on the
    // tiled kernel -> 29.5, 150^2 25.9 -> 32.4, 200^2 19.9 with 8 points -> 22.3 with 10)
    const bool small = nblk == 2 && rpw <= 13 && r3 && wide2_lds_bytes(N, r3, rows2, ts) <= 160 * 1024 / 3 &&
                       !getenv("BIOEM_NO_WIDE2_SMALL");
    // the same for one column block (measured: 128^2 +-40 px 21.1 -> 23.3, +-30 px 27.8 -> 38.6, 120^2 +-25 px 29.2 -> 38.6;
    // 90^2 +-30 px 30.2 with 30 points at two waves -> 43.1 with 10, 120^2 +-30 px 34.9 with 8 -> 39.1 with 12)
    const bool small1 = nblk == 1 && rpw <= 21 && r3 && !getenv("BIOEM_NO_WIDE2_SMALL");
    if (small || small1)
      R = r3;
    // two column blocks of up to 21 rows per wave: with 16-point FFTs the kernel needs 146 registers (three waves per
    // SIMD) and half the slot space -- worth it exactly where three blocks per CU then fit (224^2 +-26 px: 17.6 -> 18.7
    // M/s; one row more and only two fit: 14.3)
    if (R == 32 && nblk == 2 && !small && wide2_lds_bytes(N, 16, rows2, ts) <= 160 * 1024 / 3 &&
        !getenv("BIOEM_NO_WIDE2_SMALL"))
      R = 16;
    // sizes whose power-of-two part is 8 (200, 120, 280): the two-wave instantiations run faster on the longest
    // mixed length (200^2 +-30 px 11.2 -> 15.3 M/s, +-40 px 9.2 -> 12.7 with 20 points)
    if (R == 8 && !small && !small1 && !getenv("BIOEM_W2_R") && !getenv("BIOEM_POW2_FFT"))
    {
      static const int mixed[] = {30, 20, 18, 12, 10};
      for (int r : mixed)
        if (N % r == 0)
        {
          R = r;
          break;
        }
    }
    // measured against the tiled k_compare_wide (224^2): +-20 px (two 21-row tiles per axis) 15.9 vs 20.7 M/s, +-30 px
    // (three tiles) 14.8 vs 9.6, +-40 px 12.5 vs 7.2; with a T block beyond 80 KiB only one block fits a CU (256^2
    // +-40 px: 5.5 vs 6.2) -> this kernel from three tiles per axis on, while two blocks per CU fit
    // a T block that leaves one block per CU goes through LDS in two halves of the window rows where that brings the
    // second block back (k_compare_wide2<.., HALVES = 2>: 256^2 +-40 px 6.1 on the tiled kernel, 6.7 at one block per CU,
    // 11.1 in halves).  Halves + 16-point FFTs for a THIRD block per CU at 224^2 lose: +-40 px 12.1 vs 13.9, +-30 px 14.8 vs 17.1
    const int hrows = (rows2 / 2 + 1) & ~1;
    // three column blocks (256 < N <= 384): 63 rows of T accumulators per wave at two waves per SIMD, against the tiled
    // kernel 320^2 +-30 px 5.5 -> 7.4 M/s, +-40 px 4.1 -> 6.8, 288^2 +-30 px 6.0 -> 9.4, 272^2 +-40 px 4.8 -> 7.2, 384^2 +-40 px
    // 3.1 -> 5.7; from 32 window rows on (320^2 +-20 px 7.8 -> 9.7, 288^2 8.6 -> 10.0, 300^2 7.8 -> 9.7)
    const bool blocks3 = nblk == 3 && (R == 32 || ((R == 16 || R == 8 || R == 30 || R == 20 || R == 12 || R == 10) && !nyq)) &&
                         !getenv("BIOEM_NO_WIDE2_BLOCKS3");
    const bool blocks4 = nblk == 4 && (R == 32 || (R == 16 && !nyq)) && rpw <= 11 && h->nd > 31 &&
                         !getenv("BIOEM_NO_WIDE2_BLOCKS4");
    const bool halves2 = (nblk == 2 || blocks3 || blocks4) && !small && wide2_lds_bytes(N, R, rows2, ts) > 80 * 1024 &&
                         wide2_lds_bytes(N, R, hrows, ts) <= 80 * 1024 && !getenv("BIOEM_NO_WIDE2_HALVES");
    const int N1 = N / R;
    const int ldsRows = halves2 ? hrows : rows2;
    const bool pays = ((h->nd > 42 || ((blocks3 || blocks4) && h->nd > 31) || ((small || small1) && (h->nd > 31 || (midWindow && !mixedLen)))) &&
                       wide2_lds_bytes(N, R, ldsRows, ts) <= 80 * 1024) ||
                      getenv("BIOEM_FORCE_WIDE2");
    const bool rows24 = nblk == 2 && (R == 32 || (R == 16 && !nyq)) && rpw > 21 && rpw <= 24; // 208^2 +-42 px: 3.2 M/s tiled
    if (pays && (nblk <= 2 || blocks3 || blocks4) && rpw <= (nblk == 1 ? 32 : rows24 ? 24 : 21) && N1 <= 32 && h->nd <= 128 &&
        (!nyq || mD <= 42) &&
        wide2_lds_bytes(N, R, ldsRows, ts) <= 160 * 1024)
    {
      h->w2Halves = halves2 ? 2 : 1;
      h->wide2 = true;
      h->fast = R / 2;
      h->N1 = N1;
      h->nyq = nyq;
      h->w2NBLK = nblk;
      h->w2NRW = nblk == 1 ? 32 : rows24 ? 24 : blocks4 ? 11 : 21;
      if (small)
        h->w2NRW = rpw <= 11 ? 11 : 13;
      if (small1)
        h->w2NRW = 21;
      h->w2TS = ts;
      h->w2Rows2 = ldsRows; // rows of the T block in LDS
      h->nyqWD = mD <= 20 ? 20 : mD <= 31 ? 31 : 42;
      if (nyq)
        h->winD = h->nyqWD; // sizes the Nyquist pre-kernel's tables
      int ldsFinal = (int) wide2_lds_bytes(N, R, ldsRows, ts);
      // eight waves per comparison (512-thread blocks, `k_compare_wide2<.., NW = 8>`): 11 rows per wave over two column
      // blocks with a 16- / 12- / 10- / 8-point FFT -- 110 registers, four waves per SIMD at two blocks per CU.  It wins
      // where the four-wave kernel of that length is held to two blocks per CU by its T block (208^2 +-30 px 14.5 ->
      // 17.7 M/s, +-40 px 11.8 -> 14.1, 240^2 +-30 px 13.4 -> 16.0, 176^2 +-40 px 13.4 -> 15.2, 144^2 +-40 px 14.7 -> 17.4) and
      // loses against three blocks per CU (176^2 +-30 px 22.9 vs 19.1), against the 32-point two-wave kernel (224^2 +-40 px
      // 13.8 vs 13.4) and with the T block in halves (240^2 +-40 px 11.8 vs 9.5: spills under the 128-register cap)
      {
        int r8 = 0;
        if (const char *f8 = getenv("BIOEM_W2_W8"))
          r8 = atoi(f8); // experiments: force a length (16, 12, 10, 8)
        else if ((R == 16 || R == 10) && !nyq && !getenv("BIOEM_NO_WIDE2_W8"))
          r8 = R; // (250^2 +-30 px 10.2 -> 12.1 with 10 points; 200^2 keeps its 20-point two-wave kernel: 15.3 vs 14.4)
        const bool ok8 = (r8 == 16 || ((r8 == 12 || r8 == 10 || r8 == 8) && !nyq)) && N % r8 == 0 && N / r8 <= 32;
        if (ok8 && nblk == 2 && !small && h->w2Halves == 1 && (h->nd + 7) / 8 <= 11 &&
            wide2_lds_bytes(N, r8, rows2, ts, 4) > 160 * 1024 / 3 && wide2_lds_bytes(N, r8, rows2, ts, 8) <= 80 * 1024)
        {
          R = r8;
          h->w2NW = 8;
          h->fast = r8 / 2;
          h->N1 = N / r8;
          h->w2NRW = 11;
          h->w2Rows2 = rows2;
          ldsFinal = (int) wide2_lds_bytes(N, r8, rows2, ts, 8);
        }
      }
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(
                                           wide2_pick(R, h->w2NRW, nblk, nyq, h->w2Halves, h->w2NW)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ldsFinal));
    }
  }
  // four column blocks with 45..88 window rows: eight waves per comparison, one 512-thread block per CU (T block whole
  // or in halves within 160 KiB) -- the tiled kernel runs 512^2 +-40 px at 1.7 M/s
  if (!h->wide2 && N % 2 == 0 && h->nd == 2 * mD + 1 && h->nd > 44 && (h->nd + 7) / 8 <= 11 && !getenv("BIOEM_NO_WIDE2") &&
      !getenv("BIOEM_NO_WIDE2_BLOCKS4"))
  {
    const bool nyq = BIOEM_NYQUIST_SPLIT && (N / 2) % 64 == 0;
    const int nblk = nyq ? (h->H - 1) / 64 : (h->H + 63) / 64;
    const int R = (N % 32 == 0) ? 32 : (N % 16 == 0 && !nyq) ? 16 : 0;
    const int rows2 = 2 * ((h->nd + 1) / 2), hrows = (rows2 / 2 + 1) & ~1;
    int ts = h->H;
    while (ts % 16 != 4)
      ts++;
    // (measured: 512^2 +-40 px 1.67 -> 2.28 M/s, +-30 px 2.11 -> 2.96, 448^2 +-40 px 2.48 -> 3.71; with 16 points 400^2 +-40 px
    // 2.80 -> 3.11 but +-30 px 3.74 -> 3.42: from 71 rows on there)
    if (nblk == 4 && R && (R == 32 || h->nd > 70) && N / R <= 32 && (!nyq || mD <= 42))
    {
      const bool full = wide2_lds_bytes(N, R, rows2, ts, 8) <= 160 * 1024;
      const bool half = !full && wide2_lds_bytes(N, R, hrows, ts, 8) <= 160 * 1024;
      if (full || half)
      {
        h->wide2 = true;
        h->w2NW = 8;
        h->w2Halves = half ? 2 : 1;
        h->fast = R / 2;
        h->N1 = N / R;
        h->nyq = nyq;
        h->w2NBLK = 4;
        h->w2NRW = 11;
        h->w2TS = ts;
        h->w2Rows2 = half ? hrows : rows2;
        h->nyqWD = mD <= 20 ? 20 : mD <= 31 ? 31 : 42;
        if (nyq)
          h->winD = h->nyqWD;
        HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(wide2_pick(R, 11, 4, nyq, h->w2Halves, 8)),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int) wide2_lds_bytes(N, R, h->w2Rows2, ts, 8)));
      }
    }
  }
  if (!h->wide2 && N >= 8 && (mD > 15 || h->nd > 31) && h->nd == 2 * mD + 1 && !getenv("BIOEM_NO_TILES"))
  {
    const int W = h->nd;
    // launches^2 x the measured cost of one launch of the 21- / 27- / 31-row kernel (ms at 224^2)
    static const int tileRows[3] = {21, 27, 31};
    static const double tileCost[3] = {6.5, 9.25, 9.9};
    double best = 1e300;
    for (int k = 0; k < 3; k++)
    {
      const int nt = (W + tileRows[k] - 1) / tileRows[k];
      if (nt * nt * tileCost[k] < best)
      {
        best = nt * nt * tileCost[k];
        h->tileT = tileRows[k];
      }
    }
    if (getenv("BIOEM_TILE_ROWS"))
    {
      const int t = atoi(getenv("BIOEM_TILE_ROWS"));
      h->tileT = (t == 31 || t == 27) ? t : 21;
    }
    // k_compare_wide shares the column transforms between the y-tiles of an x-tile: 21-row tiles, power-of-two
    // at most four 64-column blocks
    const bool nyqSize = BIOEM_NYQUIST_SPLIT && (N / 2) % 64 == 0; // Nyquist column outside the 64-column blocks
    const int wideBlocks = nyqSize ? (h->H - 1) / 64 : (h->H + 63) / 64;
    if (N % 2 == 0 && wideBlocks <= 4 && h->gs <= 2 && !getenv("BIOEM_NO_WIDE") && !getenv("BIOEM_TILE_ROWS"))
    {
      h->tileT = 21;
      // waves per comparison: one per y-tile, and at least one per column block
      h->wideWPC = ((W + 20) / 21 == 2 && wideBlocks <= 2) ? 2 : 4;
    }
    h->tilesPerAxis = (W + h->tileT - 1) / h->tileT;
    h->winD = (h->tileT - 1) / 2;
    for (int k = 0; k < h->tilesPerAxis; k++)
    {
      h->tileCenter.push_back(-mD + k * h->tileT + h->winD);               // centre row of tile k
      h->tileValid.push_back(std::min(h->tileT, W - k * h->tileT));        // rows of tile k inside the window
    }
  }
  if (!h->wide2)
    h->fast = 0;
  if (!h->wide2 && N % 2 == 0 && N >= 8 && ((maxD / h->gs <= 15 && h->nd <= 31) || h->tileT))
  {
    int R = (N % 32 == 0) ? 32 : (N % 16 == 0) ? 16 : (N % 8 == 0) ? 8 : (N % 4 == 0) ? 4 : 2;
    // 31-row window: a 16-point register FFT keeps the kernel at 3 waves per SIMD (see fast_half_t); sizes that
    // take the Nyquist split (N/2 a multiple of 64) keep R = 32
    if (h->winD > 10 && R == 32 && fast_half_t(15, 16) && (N / 2) % 64 != 0 && !getenv("BIOEM_WIDE_R32"))
      R = 16;
    if (R < 8 && h->gs == 1 && !getenv("BIOEM_POW2_FFT"))
    { // power-of-two part 2 or 4: the largest 2/3/5-smooth even divisor <= 30 (mixed-radix register FFT) wins
      // (measured: 250^2 20 -> 40 M/s, 180^2 47 -> 53, 100^2 129 -> 143; with a part of 8 it does not: 200^2, 120^2)
      // 27/31-row windows: lengths up to 16 keep three waves per SIMD (fast_half_t), as for the powers of two
      static const int mixed[] = {30, 20, 18, 12, 10, 6};
      const bool small16 = h->winD > 10 && fast_half_t(15, 16) && !getenv("BIOEM_WIDE_R32");
      for (int r : mixed)
        if (N % r == 0 && r > R && !(small16 && r > 16))
        {
          R = r;
          break;
        }
    }
    h->fast = R / 2;
  }
  if (!h->wide2)
  {
    h->N1 = h->fast ? N / (2 * h->fast) : 0;
    h->nyq = BIOEM_NYQUIST_SPLIT && h->fast && (N / 2) % 64 == 0;
    // 11-row windows: with four column blocks and a length <= 16 (or 40+ column steps) the 21-row template is the faster
    // one (+-5 px: 432^2 11.0 -> 13.0 M/s, 360^2 16.7 -> 19.2, 400^2 13.5 -> 14.7; with 32 points 448^2 / 512^2 tie; up to
    // 336^2 and at 384^2 the 11-row template wins by 7-25 %)
    const int nblkF = h->nyq ? (h->H - 1) / 64 : (h->H + 63) / 64;
    if (h->fast && h->winD == 5 && !h->tileT && ((nblkF >= 4 && h->fast <= 8) || h->N1 >= 40) && !getenv("BIOEM_KEEP_WD5"))
      h->winD = 10;
  }
  // 23..31-row windows on even sizes: k_compare_fastm -- window pass on the matrix cores, register FFT of at most 16
  // points (27/31 rows of T accumulators + a longer FFT do not fit three waves per SIMD)
  if (!h->wide2 && h->fast && (h->winD > 10 || getenv("BIOEM_FASTM_ALL")) && !(h->tileT && h->wideWPC))
  {
    int R = 2 * h->fast;
    if (R > 16 && !getenv("BIOEM_FASTM_R32"))
    {
      static const int lens[] = {16, 12, 10, 8, 6, 4, 2};
      for (int l : lens)
        if (N % l == 0 && (h->gs == 1 || (l & (l - 1)) == 0))
        {
          R = l;
          break;
        }
    }
    if (h->nyq && !getenv("BIOEM_FASTM_R32"))
      R = 16; // (N / 2 a multiple of 64)
    h->fast = R / 2;
    h->N1 = N / R;
    h->fastm = true;
  }
  // no even factor (odd N) but a window of at most 31 rows: k_compare_rows (reference layout, direct column sums,
  // the fast kernel's T exchange / window / posterior) instead of the generic kernel
  h->rowsK = !h->fast && N >= 8 && (h->tileT || (mD <= 15 && h->nd <= 31)) && !getenv("BIOEM_NO_ROWS_KERNEL");
  if (h->rowsK && h->gs == 1 && !getenv("BIOEM_NO_ODD_FFT"))
  {
    static const int oddLens[] = {25, 15, 9, 5, 3};
    for (int r : oddLens)
      if (N % r == 0)
      {
        h->oddR = r;
        h->N1 = N / r;
        break;
      }
  }
  if (h->tileT && !h->fast && !h->rowsK)
  { // no tiled kernel available after all: plain generic kernel on the whole window
    h->tileT = 0;
    h->tilesPerAxis = 1;
    h->wideWPC = 0;
  }
  // LDS budget check
  if (!h->wide2)
  {
    // generic kernel: as many waves per block (4, 2, 1) as its per-wave T block [nd][H] lets fit
    h->genericWaves = 4;
    while (!h->fast && h->genericWaves > 1 && compare_lds_bytes(N, h->H, h->nd, h->genericWaves) > 160 * 1024)
      h->genericWaves >>= 1;
    const size_t lds = h->fastm   ? fastm_lds_bytes(N)
                       : h->fast  ? fast_lds_bytes(N, 2 * h->winD + 1, 4, fast_half_t(h->winD, 2 * h->fast))
                       : h->rowsK ? fast_lds_bytes(N, 2 * h->winD + 1, 4, false)
                                  : compare_lds_bytes(N, h->H, h->nd, h->genericWaves);
    if (lds > 160 * 1024)
    {
      h->err = "configuration exceeds the 160 KiB LDS budget of the comparison kernel";
      return 2;
    }
    if (h->fastm)
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(fastm_kernel(h->winD, 2 * h->fast, h->nyq, h->gs)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    else if (h->fast)
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(fast_kernel(h->winD, 2 * h->fast, h->nyq, h->gs)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    else if (h->rowsK)
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(rows_kernel(h->winD, h->gs, h->oddR)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) fast_lds_bytes(N, 2 * h->winD + 1, 4, false)));
    else
    {
      const size_t ldsg = compare_lds_bytes(N, h->H, h->nd, h->genericWaves);
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_compare_generic),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsg));
    }
  }

  // batch sizing: conv buffer <= ~96 MiB, partial buffer <= ~128 MiB
  const size_t M = (size_t) h->M;
  size_t ocCap = (96u << 20) / (M * sizeof(float2));
  // (tiled wide windows keep one partial per tile and comparison besides the merged one)
  const size_t partBuffers = 1 + (h->tileT ? (size_t) h->tilesPerAxis * h->tilesPerAxis : 0);
  size_t partCap = (128u << 20) / ((size_t) nMaps * sizeof(Partial));
  if (partBuffers > 1)
    partCap = std::max<size_t>((size_t) nCTF, (1024u << 20) / ((size_t) nMaps * sizeof(Partial) * partBuffers));
  if (ocCap > partCap)
    ocCap = partCap;
  // Few particles (BASELINE config 1: 10): a batch of 64 orientations is a comparison launch of a few thousand pairs
  // behind a preparation whose duration is set by latencies, not work (the ordered Parseval sum of k_convolve is one
  // sequential float chain per conv spectrum, bit-pinned to bioem.cpp:1896-1914) -- all chains of a batch run side by
  // side, so the batch grows until one launch compares ~320 000 pairs (what 64 orientations are at 1 000 particles x 5
  // CTFs); the conv buffer may then take 1 GiB per pipeline slot instead of 96 MiB.
  int obMax = 64;
  {
    const long long perOrient = (long long) nCTF * nMaps;
    if (perOrient * 64 < 320000 && !getenv("BIOEM_FIXED_BATCH"))
    {
      obMax = (int) std::min<long long>(2048, ((320000 + perOrient - 1) / perOrient + 63) / 64 * 64);
      ocCap = std::min(partCap, (size_t) (1024u << 20) / (M * sizeof(float2)));
    }
  }
  int OB = (int) (ocCap / (size_t) nCTF);
  if (OB < 1)
    OB = 1;
  if (OB > obMax)
    OB = obMax;
  if (getenv("BIOEM_PCHUNK")) // tuning knob: particle chunk of the comparison kernel's block order
    h->pchunk = atoi(getenv("BIOEM_PCHUNK"));
  if (getenv("BIOEM_BATCH_ORIENTATIONS")) // tuning knob: orientations per batch (conv buffer = OB*nCTF spectra)
    OB = std::max(1, std::min(OB, atoi(getenv("BIOEM_BATCH_ORIENTATIONS"))));
  if (OB > angO1 - angO0)
    OB = angO1 - angO0;
  h->OB = OB;
  h->maxOC = OB * nCTF;
  h->chunkB = OB > 32 ? OB : 32;

  HIP_CHECK(h, hipMalloc(&h->dRef, sizeof(float2) * M * nMaps));
  HIP_CHECK(h, hipMalloc(&h->dSumRef, sizeof(float) * nMaps));
  HIP_CHECK(h, hipMalloc(&h->dSumsqRef, sizeof(float) * nMaps));
  HIP_CHECK(h, hipMalloc(&h->dCTF, sizeof(float2) * M * nCTF));
  HIP_CHECK(h, hipMalloc(&h->dCtfParam, sizeof(float) * 3 * nCTF));
  HIP_CHECK(h, hipMalloc(&h->dAngles, sizeof(float4) * nAngles));
  HIP_CHECK(h, hipMalloc(&h->dTw, sizeof(float2) * (N + 1)));
  HIP_CHECK(h, hipMalloc(&h->dTwD, sizeof(double2) * N));
  HIP_CHECK(h, hipMalloc(&h->dDisp, sizeof(int) * h->nd));
  HIP_CHECK(h, hipMalloc(&h->dLtab, sizeof(double2) * 64));
  HIP_CHECK(h, hipMalloc(&h->dProjReal, sizeof(double) * (size_t) h->chunkB * N * N));
  HIP_CHECK(h, hipMalloc(&h->dTempDen, sizeof(double) * h->chunkB));
  HIP_CHECK(h, hipMalloc(&h->dRowSpec, sizeof(double2) * (size_t) h->chunkB * M));
  HIP_CHECK(h, hipMalloc(&h->dSpecRef, sizeof(float2) * (size_t) h->chunkB * M));
  HIP_CHECK(h, hipMalloc(&h->dScratch, sizeof(float) * (size_t) ((h->maxOC + 31) & ~31) * ((M + 3) & ~(size_t) 3)));
  HIP_CHECK(h, hipMalloc(&h->dConv, sizeof(float2) * (size_t) h->maxOC * M));
  HIP_CHECK(h, hipMalloc(&h->dParams, sizeof(bioem_hip_param5) * h->maxOC));
  HIP_CHECK(h, hipMalloc(&h->dPostC, sizeof(double2) * h->maxOC));
  HIP_CHECK(h, hipMalloc(&h->dPartials, sizeof(Partial) * (size_t) nMaps * h->maxOC));
  if (h->nyq)
    HIP_CHECK(h, hipMalloc(&h->dTnyq, sizeof(float) * (size_t) nMaps * h->maxOC * (2 * h->winD + 1)));
  if (h->tileT)
  {
    const int nT = h->tilesPerAxis;
    HIP_CHECK(h, hipMalloc(&h->dPartTiles, sizeof(Partial) * (size_t) nT * nT * nMaps * h->maxOC));
    HIP_CHECK(h, hipMalloc(&h->dConvShift, sizeof(float2) * (size_t) h->maxOC * M));
    std::vector<int> local(h->tileT), rank(2 * mD + 1, 0);
    for (int j = 0; j < h->tileT; j++)
      local[j] = h->gs * (j - h->winD); // sorted local list: the kernel's rows and lanes in the same order
    for (int v = 0; v < h->nd; v++)
      rank[h->disp[v] / h->gs + mD] = v; // visiting rank of window row m in the reference's order
    HIP_CHECK(h, hipMalloc(&h->dDispLocal, sizeof(int) * local.size()));
    HIP_CHECK(h, hipMemcpy(h->dDispLocal, local.data(), sizeof(int) * local.size(), hipMemcpyHostToDevice));
    HIP_CHECK(h, hipMalloc(&h->dRankOfRow, sizeof(int) * rank.size()));
    HIP_CHECK(h, hipMemcpy(h->dRankOfRow, rank.data(), sizeof(int) * rank.size(), hipMemcpyHostToDevice));
    HIP_CHECK(h, hipMalloc(&h->dTileCenter, sizeof(int) * nT));
    HIP_CHECK(h, hipMemcpy(h->dTileCenter, h->tileCenter.data(), sizeof(int) * nT, hipMemcpyHostToDevice));
    HIP_CHECK(h, hipMalloc(&h->dTileValid, sizeof(int) * nT));
    HIP_CHECK(h, hipMemcpy(h->dTileValid, h->tileValid.data(), sizeof(int) * nT, hipMemcpyHostToDevice));
    if (h->wideWPC)
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(wide_kernel(2 * h->fast, h->gs, h->wideWPC, h->nyq)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) wide_lds_bytes(N, h->H, h->wideWPC, h->nyq)));
  }
  h->devProbBytes = bioem_hip_prob_size(nMaps, angO1 - angO0, pd->writeAngles);
  h->probBytes = shard ? bioem_hip_prob_size(nMaps, 0, 0) : h->devProbBytes;
  HIP_CHECK(h, hipMalloc(&h->dProb, h->devProbBytes));
  {
    // projection/convolution are filler work: lowest priority so that comparison blocks win the CUs
    int prLow = 0, prHigh = 0;
    HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    HIP_CHECK(h, hipStreamCreateWithPriority(&h->prepStream, hipStreamNonBlocking, prLow));
  }
  HIP_CHECK(h, hipMalloc(&h->dProjReal2, sizeof(double) * (size_t) h->OB * N * N));
  HIP_CHECK(h, hipMalloc(&h->dTempDen2, sizeof(double) * h->OB));
  HIP_CHECK(h, hipMalloc(&h->dRowSpec2, sizeof(double2) * (size_t) h->OB * M));
  HIP_CHECK(h, hipMalloc(&h->dSpecRef2, sizeof(float2) * (size_t) h->OB * M));
  HIP_CHECK(h, hipMalloc(&h->dScratch2, sizeof(float) * (size_t) ((h->maxOC + 31) & ~31) * ((M + 3) & ~(size_t) 3)));
  HIP_CHECK(h, hipMalloc(&h->dConv2, sizeof(float2) * (size_t) h->maxOC * M));
  HIP_CHECK(h, hipMalloc(&h->dParams2, sizeof(bioem_hip_param5) * h->maxOC));
  HIP_CHECK(h, hipMalloc(&h->dPostC2, sizeof(double2) * h->maxOC));
  for (int i = 0; i < 2; i++)
  {
    HIP_CHECK(h, hipEventCreateWithFlags(&h->prepDone[i], hipEventDisableTiming));
    HIP_CHECK(h, hipEventCreateWithFlags(&h->cmpDone[i], hipEventDisableTiming));
  }

  if (h->wide2)
  { // recombination twiddles exp(2 pi i dx k1 / N), dx = (m - mD) gs, laid out per (k1, wave): the NRW window rows a
    // wave folds are contiguous (rows beyond the window: zero)
    const int NRW = h->w2NRW, NWV = h->w2NW, rpw = (h->nd + NWV - 1) / NWV;
    std::vector<float2> t2((size_t) h->N1 * NWV * NRW, make_float2(0.f, 0.f));
    for (int k1 = 0; k1 < h->N1; k1++)
      for (int w = 0; w < NWV; w++)
        for (int d = 0; d < NRW; d++)
        {
          const int m = w * rpw + d;
          if (m >= h->nd)
            continue;
          const long long dx = (long long) (m - mD) * h->gs;
          const double ang = 2.0 * M_PI * (double) (((dx * k1) % N + N) % N) / (double) N;
          t2[((size_t) k1 * NWV + w) * NRW + d] = make_float2((float) cos(ang), (float) sin(ang));
        }
    HIP_CHECK(h, hipMalloc(&h->dTwk2, sizeof(float2) * t2.size()));
    HIP_CHECK(h, hipMemcpy(h->dTwk2, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice));
  }
  std::vector<float2> tw(N + 1);
  std::vector<double2> twd(N);
  for (int k = 0; k <= N; k++)
  {
    const double ang = 2.0 * M_PI * (double) (k % N) / (double) N;
    tw[k] = make_float2((float) cos(ang), (float) sin(ang));
    if (k < N)
      twd[k] = make_double2(cos(ang), sin(ang));
  }
  HIP_CHECK(h, hipMemcpy(h->dTw, tw.data(), sizeof(float2) * (N + 1), hipMemcpyHostToDevice));
  HIP_CHECK(h, hipMemcpy(h->dTwD, twd.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
  HIP_CHECK(h, hipMemcpy(h->dDisp, h->disp.data(), sizeof(int) * h->nd, hipMemcpyHostToDevice));
  {
    // log table: bin i of the mantissa interval [1,2): c = 1/centre, entry {c, -log(c)}
    std::vector<double2> lt(64);
    for (int i = 0; i < 64; i++)
    {
      const double c = 1.0 / (1.0 + ((double) i + 0.5) / 64.0);
      lt[i] = make_double2(c, -log(c));
    }
    HIP_CHECK(h, hipMemcpy(h->dLtab, lt.data(), sizeof(double2) * 64, hipMemcpyHostToDevice));
  }
  if (h->fast || h->rowsK)
  {
    const int NW = 2 * h->winD + 1;
    // k_compare_rows: a "register FFT" of length 1, one table row per kx; k_compare_oddfft: N1 = N / oddR
    const int nK1 = (h->rowsK && !h->oddR) ? N : h->N1;
    std::vector<float2> twk((size_t) nK1 * NW);
    for (int k1 = 0; k1 < nK1; k1++)
      for (int d = -h->winD; d <= h->winD; d++)
      {
        const double ang = 2.0 * M_PI * (double) ((((long long) d * h->gs * k1) % N + N) % N) / (double) N;
        twk[(size_t) k1 * NW + d + h->winD] = make_float2((float) cos(ang), (float) sin(ang));
      }
    HIP_CHECK(h, hipMalloc(&h->dTwk, sizeof(float2) * twk.size()));
    HIP_CHECK(h, hipMemcpy(h->dTwk, twk.data(), sizeof(float2) * twk.size(), hipMemcpyHostToDevice));
    if (h->nyq)
    { // Nyquist pre-kernel: per (k1, k2 pair) the 2*winD+1 twiddle pairs w^(kx0 m gs), w^(kx1 m gs), contiguous
      const int R2 = h->fast;
      std::vector<float2> twn((size_t) (N / 2) * NW * 2);
      for (int k1 = 0; k1 < h->N1; k1++)
        for (int k2p = 0; k2p < R2; k2p++)
          for (int m = -h->winD; m <= h->winD; m++)
            for (int e = 0; e < 2; e++)
            {
              const long long kx = (long long) h->N1 * (2 * k2p + e) + k1;
              const int idx = (int) (((kx * m * h->gs) % N + N) % N);
              twn[(((size_t) (k1 * R2 + k2p) * NW) + (m + h->winD)) * 2 + e] = tw[idx];
            }
      HIP_CHECK(h, hipMalloc(&h->dTwNyq, sizeof(float2) * twn.size()));
      HIP_CHECK(h, hipMemcpy(h->dTwNyq, twn.data(), sizeof(float2) * twn.size(), hipMemcpyHostToDevice));
    }
  }
  // BIOEM_SIGNATURE_LOG=<file>: one line per handle with the comparison-kernel instantiations its launches will use
  // (scripts/check_kernel_coverage.py holds the lines of a test run against the kernels in the code object)
  if (const char *lg = getenv("BIOEM_SIGNATURE_LOG"))
    if (FILE *f = fopen(lg, "a"))
    {
      fprintf(f, "%s\n", bioem_hip_kernel_signature(h));
      if (h->nyq)
        fprintf(f, "k_nyquist_rows<%d>\n", h->wide2 ? h->nyqWD : (h->tileT && h->wideWPC) ? 10 : h->winD);
      fclose(f);
    }
  return 0;
}

int bioem_hip_create(bioem_hip_handle *out, int device, const bioem_hip_param_device *pd, int nMaps, int nAngles,
                     int nCTF, int algo)
{
  return create_impl(out, device, pd, nMaps, nAngles, nCTF, algo, false, 0, nAngles);
}

int bioem_hip_create_shard(bioem_hip_handle *out, int device, const bioem_hip_param_device *pd, int nMaps, int nAngles,
                           int nCTF, int algo, int iOrientBegin, int iOrientEnd)
{
  return create_impl(out, device, pd, nMaps, nAngles, nCTF, algo, true, iOrientBegin, iOrientEnd);
}

int bioem_hip_destroy(bioem_hip_handle h)
{
  if (!h)
    return 0;
  hipSetDevice(h->device);
  if (h->stream)
    hipStreamSynchronize(h->stream);
  drain_events(h);
  for (hipEvent_t e : h->evPool)
    hipEventDestroy(e);
  void *ptrs[] = {h->dRef,     h->dSumRef,  h->dSumsqRef, h->dCTF,     h->dCtfParam, h->dPts,   h->dAngles,
                  h->dTw,      h->dTwD,     h->dDisp,     h->dLtab,    h->dTwk,     h->dProjReal, h->dTempDen,  h->dRowSpec, h->dSpecRef,
                  h->dScratch, h->dConv,    h->dParams,   h->dPartials, h->dProb,
                  h->dProjReal2, h->dTempDen2, h->dRowSpec2, h->dSpecRef2, h->dScratch2, h->dConv2, h->dParams2,
                  h->dTnyq, h->dTwNyq, h->dPartTiles, h->dConvShift, h->dDispLocal, h->dRankOfRow, h->dTileCenter, h->dTileValid,
                  h->dCand, h->dSend, h->dRecv, h->dMerged, h->dTwk2, h->dPostC, h->dPostC2};
  for (void *p : ptrs)
    if (p)
      hipFree(p);
  compat_free(h);
  for (int i = 0; i < 2; i++)
  {
    if (h->prepDone[i])
      hipEventDestroy(h->prepDone[i]);
    if (h->cmpDone[i])
      hipEventDestroy(h->cmpDone[i]);
  }
  if (h->prepStream)
  {
    hipStreamSynchronize(h->prepStream);
    hipStreamDestroy(h->prepStream);
  }
  if (h->stream)
    hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int bioem_hip_upload_particles(bioem_hip_handle h, const float *refFFT, const float *sum, const float *sumsq)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  HIP_CHECK(h, hipMemcpyAsync(h->dSumRef, sum, sizeof(float) * h->nMaps, hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipMemcpyAsync(h->dSumsqRef, sumsq, sizeof(float) * h->nMaps, hipMemcpyHostToDevice, h->stream));
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    HIP_CHECK(h, hipMemcpyAsync(h->dSpecRef, refFFT + 2 * M * (size_t) b, sizeof(float2) * M * n,
                                hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_reorder, dim3(1024), dim3(256), 0, h->stream, h->dSpecRef, h->dRef + M * (size_t) b, n, h->N,
                       h->H, h->fast, h->N1);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

int bioem_hip_upload_particle_maps(bioem_hip_handle h, const float *maps)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  const int N = h->N;
  float *dMaps = nullptr;
  HIP_CHECK(h, hipMalloc(&dMaps, sizeof(float) * (size_t) h->chunkB * N * N));
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    HIP_CHECK(h, hipMemcpyAsync(dMaps, maps + (size_t) b * N * N, sizeof(float) * (size_t) n * N * N,
                                hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_map_sums, dim3(n), dim3(256), 0, h->stream, dMaps, N * N, h->dSumRef + b, h->dSumsqRef + b);
    HIP_CHECK(h, hipGetLastError());
    if (run_r2c(h, batch_buf(h, 0), h->stream, nullptr, dMaps, n))
    {
      hipFree(dMaps);
      return 1;
    }
    hipLaunchKernelGGL(k_reorder, dim3(1024), dim3(256), 0, h->stream, h->dSpecRef, h->dRef + M * (size_t) b, n, N,
                       h->H, h->fast, h->N1);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  hipFree(dMaps);
  return 0;
}

int bioem_hip_upload_ctf(bioem_hip_handle h, const float *refCTF, const float *ctfParam3)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipMemcpy(h->dCTF, refCTF, sizeof(float2) * (size_t) h->M * h->nCTF, hipMemcpyHostToDevice));
  HIP_CHECK(h, hipMemcpy(h->dCtfParam, ctfParam3, sizeof(float) * 3 * h->nCTF, hipMemcpyHostToDevice));
  return 0;
}

int bioem_hip_upload_model(bioem_hip_handle h, const bioem_hip_model_point *pts, int nPts, float NormDen,
                           float pixelSize, int shiftX, int shiftY)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (h->dPts)
    hipFree(h->dPts);
  h->dPts = nullptr;
  HIP_CHECK(h, hipMalloc(&h->dPts, sizeof(bioem_hip_model_point) * (size_t) nPts));
  HIP_CHECK(h, hipMemcpy(h->dPts, pts, sizeof(bioem_hip_model_point) * (size_t) nPts, hipMemcpyHostToDevice));
  h->nPts = nPts;
  h->NormDen = NormDen;
  h->pixelSize = pixelSize;
  h->shiftX = shiftX;
  h->shiftY = shiftY;
  return 0;
}

int bioem_hip_upload_orientations(bioem_hip_handle h, const float *angles4, int n, int isQuat)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (n > h->nAngles)
  {
    h->err = "more orientations than the handle was created for";
    return 2;
  }
  HIP_CHECK(h, hipMemcpy(h->dAngles, angles4, sizeof(float4) * (size_t) n, hipMemcpyHostToDevice));
  h->nAnglesUp = n;
  h->isQuat = isQuat;
  return 0;
}

void *bioem_hip_host_alloc(size_t size)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess)
    return nullptr;
  return p;
}

void bioem_hip_host_free(void *ptr)
{
  if (ptr)
    hipHostFree(ptr);
}

int bioem_hip_start_run(bioem_hip_handle h, const void *pProb_host)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipMemcpyAsync(h->dProb, pProb_host, h->probBytes, hipMemcpyHostToDevice, h->stream));
  if (h->shard && h->pd.writeAngles)
  { // the shard's angle table is initialised where it lives (bioem.cpp:688-697)
    bioem_hip_prob_angle *pang = reinterpret_cast<bioem_hip_prob_angle *>(h->dProb + sizeof(bioem_hip_prob_map) * h->nMaps);
    hipLaunchKernelGGL(k_init_angles, dim3(1024), dim3(256), 0, h->stream, pang, (size_t) (h->angO1 - h->angO0) * h->nMaps);
    HIP_CHECK(h, hipGetLastError());
  }
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_compare(bioem_hip_handle h, int iPipeline, int iOrient, int iConvStart, int maxParallelConv,
                      int nTotParallelConv, const float *conv_mapsFFT, const bioem_hip_param5 *comp_params)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  if (maxParallelConv < 1 || maxParallelConv > nTotParallelConv || iOrient < h->angO0 || iOrient >= h->angO1 ||
      iConvStart < 0 || iConvStart + maxParallelConv > h->nCTF)
  {
    h->err = "bioem_hip_compare: orientation / convolution range out of bounds";
    return 2;
  }
  if (compat_alloc(h))
    return 1;
  const int k = (iPipeline & 1) * nTotParallelConv; // bioem.cpp:1388
  // with WRITE_PROB_ANGLES every orientation of a launch must be one run of rows: an orientation that returns after
  // another one was staged starts a new launch
  bool seen = false, last = !h->ringOrients.empty() && h->ringOrients.back() == iOrient;
  for (int o : h->ringOrients)
    seen = seen || o == iOrient;
  if (seen && !last && compat_flush(h))
    return 1;
  int done = 0;
  while (done < maxParallelConv)
  {
    if (h->ringCount == h->ringCap && compat_flush(h))
      return 1;
    const int half = h->ringHalf;
    bioem_hip_ctx::CompatHalf &r = h->ring[half];
    if (h->ringCount == 0 && h->cmpPending[half])
    { // the launch that last used this half (two flushes ago) must have consumed its staged rows
      HIP_CHECK(h, hipEventSynchronize(h->cmpDone[half]));
      h->cmpPending[half] = false;
    }
    const int row0 = h->ringCount;
    const int n = std::min(maxParallelConv - done, h->ringCap - row0);
    // the caller's slot is free again when this returns (bioem_cuda.cu:539-561 makes the caller wait instead)
    memcpy(r.hConv + M * row0, conv_mapsFFT + 2 * M * (size_t) (k + done), sizeof(float2) * M * n);
    for (int i = 0; i < n; i++)
    {
      r.hPar[row0 + i] = comp_params[k + done + i];
      r.hIds[row0 + i] = make_int2(iOrient, iConvStart + done + i);
    }
    HIP_CHECK(h, hipMemcpyAsync(r.dStage + M * row0, r.hConv + M * row0, sizeof(float2) * M * n, hipMemcpyHostToDevice,
                                h->copyStream));
    if (h->ringOrients.empty() || h->ringOrients.back() != iOrient)
      h->ringOrients.push_back(iOrient);
    h->ringCount += n;
    done += n;
  }
  // launch when the half is full -- or earlier when the device has nothing left to do and a launch's worth of rows
  // (32 x nMaps comparisons) is waiting: keeps the GPU busy while a short run ramps up
  bool launch = h->ringCount == h->ringCap;
  if (!launch && h->ringCount >= 32)
  {
    const int other = h->ringHalf ^ 1;
    launch = !h->cmpPending[other] || hipEventQuery(h->cmpDone[other]) == hipSuccess;
  }
  if (launch && compat_flush(h))
    return 1;
  return 0;
}

int bioem_hip_project_convolve_compare(bioem_hip_handle h, int iOrientBegin, int iOrientEnd)
{
  return bioem_hip_project_convolve_compare_ctf(h, iOrientBegin, iOrientEnd, 0, h ? h->nCTF : 0);
}

int bioem_hip_project_convolve_compare_ctf(bioem_hip_handle h, int iOrientBegin, int iOrientEnd, int iConvBegin,
                                           int iConvEnd)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (!h->dPts || iOrientBegin < 0 || iOrientEnd > h->nAnglesUp || iOrientBegin > iOrientEnd || iConvBegin < 0 ||
      iConvEnd > h->nCTF || iConvBegin >= iConvEnd)
  {
    h->err = "project_convolve_compare: model/orientations not uploaded or range invalid";
    return 2;
  }
  if (iOrientBegin < iOrientEnd && (iOrientBegin < h->angO0 || iOrientEnd > h->angO1))
  {
    h->err = "project_convolve_compare: orientations outside the range this shard handle was created for";
    return 2;
  }
  const int nC = iConvEnd - iConvBegin;
  if (compat_flush(h)) // rows staged through the reference-compatible entry go first (call order)
    return 1;
  // two-slot pipeline: projection + convolution of batch b+1 run on prepStream while batch b is compared
  const int nb = (iOrientEnd - iOrientBegin + h->OB - 1) / h->OB;
  auto prep = [&](int b) -> int {
    const int slot = b & 1;
    const int o0 = iOrientBegin + b * h->OB;
    const int nO = std::min(h->OB, iOrientEnd - o0);
    const BatchBuf bb = batch_buf(h, slot);
    if (h->cmpPending[slot])
    {
      HIP_CHECK(h, hipStreamWaitEvent(h->prepStream, h->cmpDone[slot], 0));
      h->cmpPending[slot] = false;
    }
    if (project_batch(h, bb, h->prepStream, o0, nO))
      return 1;
    if (convolve_batch(h, bb, h->prepStream, nO, iConvBegin, nC))
      return 1;
    HIP_CHECK(h, hipEventRecord(h->prepDone[slot], h->prepStream));
    return 0;
  };
  // anything still queued on the main stream that uses slot 0/1 buffers (debug hooks, compat entry) goes first
  HIP_CHECK(h, hipEventRecord(h->cmpDone[0], h->stream));
  HIP_CHECK(h, hipEventRecord(h->cmpDone[1], h->stream));
  h->cmpPending[0] = h->cmpPending[1] = true;
  if (nb > 0 && prep(0))
    return 1;
  for (int b = 0; b < nb; b++)
  {
    const int slot = b & 1;
    const int o0 = iOrientBegin + b * h->OB;
    const int nO = std::min(h->OB, iOrientEnd - o0);
    if (b + 1 < nb && prep(b + 1))
      return 1;
    HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->prepDone[slot], 0));
    if (launch_compare_fold(h, batch_buf(h, slot), nO * nC, o0, iConvBegin, nC))
      return 1;
    HIP_CHECK(h, hipEventRecord(h->cmpDone[slot], h->stream));
    h->cmpPending[slot] = true;
  }
  // later main-stream work (finish_run, debug hooks) must also see prepStream drained: it is, through prepDone
  return 0;
}

int bioem_hip_finish_run(bioem_hip_handle h, void *pProb_host)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (compat_flush(h))
    return 1;
  HIP_CHECK(h, hipMemcpyAsync(pProb_host, h->dProb, h->probBytes, hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  drain_events(h);
  return 0;
}

int bioem_hip_synchronize(bioem_hip_handle h)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (compat_flush(h))
    return 1;
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_merge_host(int nShards, int nMaps, int nAngles, int writeAngles, const void *const *shards, void *out)
{
  if (nShards < 1)
    return 2;
  bioem_hip_prob_map *om = reinterpret_cast<bioem_hip_prob_map *>(out);
  bioem_hip_prob_angle *oa = reinterpret_cast<bioem_hip_prob_angle *>(om + nMaps);
  for (int i = 0; i < nMaps; i++)
  {
    int who = 0;
    double cmax = reinterpret_cast<const bioem_hip_prob_map *>(shards[0])[i].Constoadd;
    for (int s = 1; s < nShards; s++)
    {
      const double c = reinterpret_cast<const bioem_hip_prob_map *>(shards[s])[i].Constoadd;
      if (c > cmax)
      {
        cmax = c;
        who = s;
      }
    }
    double tot = 0.;
    for (int s = 0; s < nShards; s++)
    {
      const bioem_hip_prob_map &m = reinterpret_cast<const bioem_hip_prob_map *>(shards[s])[i];
      tot += m.Total * exp(m.Constoadd - cmax);
    }
    om[i] = reinterpret_cast<const bioem_hip_prob_map *>(shards[who])[i];
    om[i].Total = tot;
    om[i].Constoadd = cmax;
  }
  if (writeAngles)
  {
    const size_t cnt = (size_t) nMaps * nAngles;
    for (size_t e = 0; e < cnt; e++)
    {
      double cmax = MIN_PROB;
      for (int s = 0; s < nShards; s++)
      {
        const bioem_hip_prob_angle *a =
            reinterpret_cast<const bioem_hip_prob_angle *>(reinterpret_cast<const bioem_hip_prob_map *>(shards[s]) + nMaps);
        if (a[e].ConstAngle > cmax)
          cmax = a[e].ConstAngle;
      }
      double tot = 0.;
      for (int s = 0; s < nShards; s++)
      {
        const bioem_hip_prob_angle *a =
            reinterpret_cast<const bioem_hip_prob_angle *>(reinterpret_cast<const bioem_hip_prob_map *>(shards[s]) + nMaps);
        tot += a[e].forAngles * exp(a[e].ConstAngle - cmax);
      }
      oa[e].forAngles = tot;
      oa[e].ConstAngle = cmax;
    }
  }
  return 0;
}


// ---- WRITE_PROB_ANGLES: K best orientations per particle, selected where the table lives ----
static int topk_device(bioem_hip_ctx *h, int K, double numconst)
{
  if (!h->pd.writeAngles || K < 1)
  {
    h->err = "topk_angles: handle was created without WRITE_PROB_ANGLES or K < 1";
    return 2;
  }
  if (h->candK != K)
  {
    if (h->dCand)
      hipFree(h->dCand);
    h->dCand = nullptr;
    h->candK = 0;
    HIP_CHECK(h, hipMalloc(&h->dCand, sizeof(bioem_hip_angle_candidate) * (size_t) h->nMaps * K));
    h->candK = K;
  }
  const bioem_hip_prob_angle *pang =
      reinterpret_cast<const bioem_hip_prob_angle *>(h->dProb + sizeof(bioem_hip_prob_map) * h->nMaps);
  hipLaunchKernelGGL(k_topk_angles, dim3((h->nMaps + 63) / 64), dim3(64), 0, h->stream, pang, h->angO1 - h->angO0, h->nMaps,
                     h->angO0, K, numconst, h->dCand);
  HIP_CHECK(h, hipGetLastError());
  return 0;
}

int bioem_hip_topk_angles(bioem_hip_handle h, int K, double numconst, bioem_hip_angle_candidate *out)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (compat_flush(h))
    return 1;
  if (const int rc = topk_device(h, K, numconst))
    return rc;
  HIP_CHECK(h, hipMemcpyAsync(out, h->dCand, sizeof(bioem_hip_angle_candidate) * (size_t) h->nMaps * K, hipMemcpyDeviceToHost,
                              h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_merge_topk_host(int nShards, int nMaps, int K, const bioem_hip_angle_candidate *const *cands,
                              bioem_hip_angle_candidate *out)
{
  if (nShards < 1 || K < 1)
    return 2;
  typedef std::pair<double, int> Item; // (logp, index into `all`): the reference's heap item with the orientation
                                       // replaced by a position that is monotone in it
  std::vector<bioem_hip_angle_candidate> all;
  for (int i = 0; i < nMaps; i++)
  {
    all.clear();
    for (int s = 0; s < nShards; s++)
      for (int k = 0; k < K; k++)
        if (cands[s][(size_t) i * K + k].orient >= 0)
          all.push_back(cands[s][(size_t) i * K + k]);
    // the writer walks the orientations in ascending order (bioem.cpp:1257)
    std::sort(all.begin(), all.end(),
              [](const bioem_hip_angle_candidate &a, const bioem_hip_angle_candidate &b) { return a.orient < b.orient; });
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> q;
    for (int j = 0; j < (int) all.size(); j++)
    {
      if ((int) q.size() < K)
        q.push(Item(all[j].logp, j));
      else if (q.top().first < all[j].logp)
      {
        q.pop();
        q.push(Item(all[j].logp, j));
      }
    }
    bioem_hip_angle_candidate *o = out + (size_t) i * K;
    const int cnt = (int) q.size();
    for (int r = cnt - 1; r >= 0; r--)
    {
      o[r] = all[q.top().second];
      q.pop();
    }
    for (int r = cnt; r < K; r++)
    {
      o[r].forAngles = 0.;
      o[r].ConstAngle = MIN_PROB;
      o[r].logp = -INFINITY;
      o[r].orient = -1;
      o[r].pad = 0;
    }
  }
  return 0;
}

// ---- RCCL merge (one process, n GPUs): librccl is large, so it is loaded when the first merge asks for it ----
namespace
{
struct Rccl
{
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::map<std::vector<int>, std::vector<ncclComm_t>> comms; // by device list; kept for the life of the process
  std::mutex mu;
};
Rccl g_rccl;

const char *rccl_load()
{
  if (g_rccl.lib)
    return nullptr;
  // a process must hold ONE copy of RCCL (a second one -- e.g. PyTorch's bundled librccl.so next to ROCm's -- ends in
  // a double free at exit): take the copy that is already loaded, if any, before loading one by the name other
  // libraries ask for
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void *lib = nullptr;
  for (const char *n : {"librccl.so.1", "librccl.so"})
    if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD)))
      break;
  for (const char *n : names)
    if (!lib)
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!lib)
    return "librccl.so not found (needed for the multi-GPU merge)";
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll)) dlsym(lib, "ncclCommInitAll");
  g_rccl.AllGather = (decltype(g_rccl.AllGather)) dlsym(lib, "ncclAllGather");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart)) dlsym(lib, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd)) dlsym(lib, "ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString)) dlsym(lib, "ncclGetErrorString");
  if (!g_rccl.CommInitAll || !g_rccl.AllGather || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.GetErrorString)
    return "librccl.so lacks ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd";
  g_rccl.lib = lib;
  return nullptr;
}
} // namespace

#define RCCL_CHECK(h, expr)                                                                                        \
  do                                                                                                               \
  {                                                                                                                \
    ncclResult_t r_ = (expr);                                                                                      \
    if (r_ != ncclSuccess)                                                                                         \
    {                                                                                                              \
      char buf_[512];                                                                                              \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      (h)->err = buf_;                                                                                             \
      return 1;                                                                                                    \
    }                                                                                                              \
  } while (0)

int bioem_hip_merge(bioem_hip_handle *handles, int n, void *pProbMaps_host, int K, double numconst,
                    bioem_hip_angle_candidate *cand_host)
{
  if (!handles || n < 1 || !handles[0])
    return 2;
  bioem_hip_ctx *h0 = handles[0];
  const int nMaps = h0->nMaps;
  std::vector<int> devs(n);
  for (int i = 0; i < n; i++)
  {
    if (!handles[i] || handles[i]->nMaps != nMaps)
    {
      h0->err = "bioem_hip_merge: handles of different shape";
      return 2;
    }
    devs[i] = handles[i]->device;
    for (int j = 0; j < i; j++)
      if (devs[j] == devs[i])
      {
        h0->err = "bioem_hip_merge: RCCL needs one GPU per shard (two handles share a device; use bioem_hip_merge_host)";
        return 2;
      }
  }
  if (K > 0 && !cand_host)
    return 2;
  std::lock_guard<std::mutex> lock(g_rccl.mu);
  if (const char *e = rccl_load())
  {
    h0->err = e;
    return 1;
  }
  auto it = g_rccl.comms.find(devs);
  if (it == g_rccl.comms.end())
  {
    std::vector<ncclComm_t> c(n);
    RCCL_CHECK(h0, g_rccl.CommInitAll(c.data(), n, devs.data()));
    it = g_rccl.comms.emplace(devs, c).first;
  }
  const std::vector<ncclComm_t> &comm = it->second;
  const size_t mapBytes = sizeof(bioem_hip_prob_map) * (size_t) nMaps;
  const size_t payload = mapBytes + (K > 0 ? sizeof(bioem_hip_angle_candidate) * (size_t) nMaps * K : 0);
  // stage every shard's contribution: its map entries (already on the device) and its K best orientations
  for (int i = 0; i < n; i++)
  {
    bioem_hip_ctx *h = handles[i];
    HIP_CHECK(h, hipSetDevice(h->device));
    if (compat_flush(h))
      return 1;
    if (h->sendBytes < payload)
    {
      if (h->dSend)
        hipFree(h->dSend);
      h->dSend = nullptr;
      h->sendBytes = 0;
      HIP_CHECK(h, hipMalloc(&h->dSend, payload));
      h->sendBytes = payload;
    }
    if (h->recvBytes < payload * n)
    {
      if (h->dRecv)
        hipFree(h->dRecv);
      h->dRecv = nullptr;
      h->recvBytes = 0;
      HIP_CHECK(h, hipMalloc(&h->dRecv, payload * n));
      h->recvBytes = payload * n;
    }
    HIP_CHECK(h, hipMemcpyAsync(h->dSend, h->dProb, mapBytes, hipMemcpyDeviceToDevice, h->stream));
    if (K > 0)
    {
      if (const int rc = topk_device(h, K, numconst))
        return rc;
      HIP_CHECK(h, hipMemcpyAsync(h->dSend + mapBytes, h->dCand, payload - mapBytes, hipMemcpyDeviceToDevice, h->stream));
    }
  }
  // the exchange: one all-gather over xGMI
  RCCL_CHECK(h0, g_rccl.GroupStart());
  for (int i = 0; i < n; i++)
  {
    bioem_hip_ctx *h = handles[i];
    HIP_CHECK(h, hipSetDevice(h->device));
    RCCL_CHECK(h, g_rccl.AllGather(h->dSend, h->dRecv, payload, ncclChar, comm[i], h->stream));
  }
  RCCL_CHECK(h0, g_rccl.GroupEnd());
  // fold on the first device, result to the host
  HIP_CHECK(h0, hipSetDevice(h0->device));
  if (!h0->dMerged)
    HIP_CHECK(h0, hipMalloc(&h0->dMerged, mapBytes));
  hipLaunchKernelGGL(k_merge_shards, dim3((nMaps + 127) / 128), dim3(128), 0, h0->stream, h0->dRecv, n, payload, nMaps,
                     h0->dMerged);
  HIP_CHECK(h0, hipGetLastError());
  HIP_CHECK(h0, hipMemcpyAsync(pProbMaps_host, h0->dMerged, mapBytes, hipMemcpyDeviceToHost, h0->stream));
  std::vector<std::vector<bioem_hip_angle_candidate>> gathered;
  if (K > 0)
  {
    gathered.resize(n);
    for (int s = 0; s < n; s++)
    {
      gathered[s].resize((size_t) nMaps * K);
      HIP_CHECK(h0, hipMemcpyAsync(gathered[s].data(), h0->dRecv + (size_t) s * payload + mapBytes, payload - mapBytes,
                                   hipMemcpyDeviceToHost, h0->stream));
    }
  }
  for (int i = 0; i < n; i++)
  {
    HIP_CHECK(handles[i], hipSetDevice(handles[i]->device));
    HIP_CHECK(handles[i], hipStreamSynchronize(handles[i]->stream));
  }
  if (K > 0)
  {
    std::vector<const bioem_hip_angle_candidate *> ptrs(n);
    for (int s = 0; s < n; s++)
      ptrs[s] = gathered[s].data();
    return bioem_hip_merge_topk_host(n, nMaps, K, ptrs.data(), cand_host);
  }
  return 0;
}

int bioem_hip_debug_projection(bioem_hip_handle h, int iOrient, float *spec_out)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (project_batch(h, batch_buf(h, 0), h->stream, iOrient, 1))
    return 1;
  HIP_CHECK(h, hipMemcpyAsync(spec_out, h->dSpecRef, sizeof(float2) * (size_t) h->M, hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_debug_convolution(bioem_hip_handle h, int iOrient, int iConv, float *spec_out, float *sumC,
                                float *sumsquareC)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (project_batch(h, batch_buf(h, 0), h->stream, iOrient, 1))
    return 1;
  if (convolve_batch(h, batch_buf(h, 0), h->stream, 1, 0, h->nCTF))
    return 1;
  const size_t M = (size_t) h->M;
  float2 *tmp = h->dSpecRef + M; // chunkB >= 32 slots; slot 0 holds the projection spectrum
  hipLaunchKernelGGL(k_unreorder, dim3(256), dim3(256), 0, h->stream, h->dConv + M * (size_t) iConv, tmp, 1, h->N,
                     h->H, h->fast, h->N1);
  HIP_CHECK(h, hipGetLastError());
  HIP_CHECK(h, hipMemcpyAsync(spec_out, tmp, sizeof(float2) * M, hipMemcpyDeviceToHost, h->stream));
  bioem_hip_param5 q;
  HIP_CHECK(h, hipMemcpyAsync(&q, h->dParams + iConv, sizeof(q), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  *sumC = q.sumC;
  *sumsquareC = q.sumsquareC;
  return 0;
}

int bioem_hip_debug_particles(bioem_hip_handle h, float *refFFT_out, float *sum_out, float *sumsq_out)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    hipLaunchKernelGGL(k_unreorder, dim3(1024), dim3(256), 0, h->stream, h->dRef + M * (size_t) b, h->dSpecRef, n, h->N,
                       h->H, h->fast, h->N1);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipMemcpyAsync(refFFT_out + 2 * M * (size_t) b, h->dSpecRef, sizeof(float2) * M * n,
                                hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  HIP_CHECK(h, hipMemcpy(sum_out, h->dSumRef, sizeof(float) * h->nMaps, hipMemcpyDeviceToHost));
  HIP_CHECK(h, hipMemcpy(sumsq_out, h->dSumsqRef, sizeof(float) * h->nMaps, hipMemcpyDeviceToHost));
  return 0;
}

int bioem_hip_kernel_stats(bioem_hip_handle h, double *compare_ms, long long *launches, long long *comparisons)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (compat_flush(h))
    return 1;
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  drain_events(h);
  if (compare_ms)
    *compare_ms = h->compareMs;
  if (launches)
    *launches = h->launches;
  if (comparisons)
    *comparisons = h->comparisons;
  return 0;
}

int bioem_hip_reset_kernel_stats(bioem_hip_handle h)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  drain_events(h);
  h->compareMs = 0;
  h->launches = 0;
  h->comparisons = 0;
  return 0;
}

#ifdef BIOEM_W2_STAMPS
// diagnostic build only: summed shader cycles per phase of k_compare_wide2 (and reset)
int bioem_hip_debug_w2_stamps(unsigned long long *out8)
{
  unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_w2_stamps), sizeof(z)) != hipSuccess)
    return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w2_stamps), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

int bioem_hip_uses_fast_path(bioem_hip_handle h) { return h && h->fast ? 1 : 0; }

const char *bioem_hip_kernel_name(bioem_hip_handle h)
{
  if (!h)
    return "";
  if (h->wide2)
    return "k_compare_wide2";
  if (h->fastm)
    return "k_compare_fastm";
  if (h->fast)
    return (h->tileT && h->wideWPC) ? "k_compare_wide" : "k_compare_fast";
  return h->rowsK ? (h->oddR ? "k_compare_oddfft" : "k_compare_rows") : "k_compare_generic";
}

const char *bioem_hip_kernel_signature(bioem_hip_handle h)
{
  if (!h)
    return "";
  static thread_local char buf[96];
  const char *nq = h->nyq ? "true" : "false";
  if (h->wide2)
    if (h->w2Halves == 2)
      snprintf(buf, sizeof(buf), h->w2NW == 8 ? "k_compare_wide2<%d, %d, %d, %s, 2, 8>" : "k_compare_wide2<%d, %d, %d, %s, 2>",
               2 * h->fast, h->w2NRW, h->w2NBLK, nq);
    else
      snprintf(buf, sizeof(buf), h->w2NW == 8 ? "k_compare_wide2<%d, %d, %d, %s, 1, 8>" : "k_compare_wide2<%d, %d, %d, %s>",
               2 * h->fast, h->w2NRW, h->w2NBLK, nq);
  else if (h->fast && h->tileT && h->wideWPC)
    snprintf(buf, sizeof(buf), "k_compare_wide<%d, %d, %d, %s>", 2 * h->fast, h->gs, h->wideWPC, nq);
  else if (h->fastm)
    snprintf(buf, sizeof(buf), "k_compare_fastm<%d, %d, %s, %d>", h->winD, 2 * h->fast, nq, h->gs);
  else if (h->fast)
    snprintf(buf, sizeof(buf), "k_compare_fast<%d, %d, %s, %d>", h->winD, 2 * h->fast, nq, h->gs);
  else if (h->rowsK && h->oddR)
    snprintf(buf, sizeof(buf), "k_compare_oddfft<%d, %d>", h->winD, h->oddR);
  else if (h->rowsK)
    snprintf(buf, sizeof(buf), "k_compare_rows<%d, %d>", h->winD, h->gs);
  else
    snprintf(buf, sizeof(buf), "k_compare_generic");
  return buf;
}

int bioem_hip_r2c(int device, int N, int nImg, const float *in, float *out)
{
  if (N < 1 || N > kMaxPixels || nImg < 1 || hipSetDevice(device) != hipSuccess || dft_allow_lds(N) != hipSuccess)
    return 1;
  const int H = N / 2 + 1;
  const size_t M = (size_t) N * H;
  std::vector<double2> twd(N);
  for (int k = 0; k < N; k++)
  {
    const double ang = 2.0 * M_PI * (double) k / (double) N;
    twd[k] = make_double2(cos(ang), sin(ang));
  }
  double2 *dTw = nullptr, *dRow = nullptr;
  float *dIn = nullptr;
  float2 *dOut = nullptr;
  int rc = 1;
  if (hipMalloc(&dTw, sizeof(double2) * N) == hipSuccess && hipMalloc(&dRow, sizeof(double2) * M * nImg) == hipSuccess &&
      hipMalloc(&dIn, sizeof(float) * (size_t) N * N * nImg) == hipSuccess &&
      hipMalloc(&dOut, sizeof(float2) * M * nImg) == hipSuccess &&
      hipMemcpy(dTw, twd.data(), sizeof(double2) * N, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dIn, in, sizeof(float) * (size_t) N * N * nImg, hipMemcpyHostToDevice) == hipSuccess)
  {
    int A, B;
    dft_split(N, A, B);
    hipLaunchKernelGGL(k_dft_rows, dim3(N, nImg), dim3(128), sizeof(double) * (3 * N + 2), 0, nullptr, dIn, nullptr,
                       1.f, N, H, A, B, dTw, dRow);
    hipLaunchKernelGGL(k_dft_cols, dim3(H, nImg), dim3(256), sizeof(double2) * 2 * N, 0, dRow, N, H, A, B, dTw, dOut);
    if (hipGetLastError() == hipSuccess &&
        hipMemcpy(out, dOut, sizeof(float2) * M * nImg, hipMemcpyDeviceToHost) == hipSuccess)
      rc = 0;
  }
  hipFree(dTw);
  hipFree(dRow);
  hipFree(dIn);
  hipFree(dOut);
  return rc;
}

} // extern "C"

