// bioem_hip.hip -- MI355X (gfx950 / CDNA4) implementation of the BioEM compare path behind the
// C ABI of include/bioem_hip.h.  Hand-written HIP, wave64, no vendor FFT/BLAS on the hot path.
//
// Hot path (replaces bioem_cuda::compareRefMaps + cuFFT, /root/reference/bioem_cuda.cu:527-684, and the
// host-side createProjection / createConvolutedProjectionMap, /root/reference/bioem.cpp:1604-1923):
//
//   prep_kernels.hpp     k_project_stamps, k_project_box  model points -> real-space projection: the spheres' footprints
//                          tabulated once per model, one block per orientation with the box of pixels the model can reach
//                          in LDS (ds_add_f64), the box stored; k_project_coords + k_project_bands (records, bands of map
//                          rows in LDS) for models beyond the box, k_project (global double atomics) beyond 426 pixels
//                        k_convolve       proj * conj(CTF) -> conv spectra in the comparison layout, sumC, Parseval terms
//                        k_parseval_ordered  sumsquareC: the reference's sequential float sum, four chains per wave
//                        k_convolve_sums  the two in one kernel (256 particles or fewer), 3...4 orientations per block
//                        k_dft_rows/cols  r2c by exact DFT on the vector units (images beyond 304 pixels)
//   r2c_fft.hpp          k_r2c_fft        r2c of the projections and particle maps (kernels_r2c.hip): one Cooley-Tukey split,
//                          register FFTs of 2...20 points in double; the rows of the projection's box only
//   dft_mfma.hpp         k_dft_rows_mfma / k_dft_cols_mfma  the r2c of image sizes with a prime factor above 19: exact DFT
//                          (double accumulation) as v_mfma_f64_16x16x4_f64 products
//                        k_reorder, k_map_sums: particle-side precompute
//   compare_fast.hpp     k_compare_fast   windows of at most 21 rows; one WAVE per (particle, orientation*CTF) comparison:
//                          spectrum product -> pruned inverse 2-D transform -> displacement-window log posterior
//                          -> wave log-sum-exp/arg-max partial.  The length-N inverse along kx is split as
//                          N = N1*R (R = 32, 16, 8, 4, 2 or a mixed length): N1 register-resident R-point FFTs per
//                          frequency column (lane = column, fft_registers.hpp), recombined only for the window rows that
//                          are consumed (output pruning), so no radix-7 butterfly is ever needed for N = 224.  The
//                          transform along ky is a pruned real DFT evaluated from LDS for the window only.
//                        k_nyquist_rows   the Nyquist column of 128^2 / 256^2 by direct summation
//                        posterior_batch  the log posterior of a batch of displacements (all comparison kernels)
//   compare_fastm.hpp    k_compare_fastm  27- / 31-row windows: the same column pass, the window pass as one 32 x 32 tile of
//                          v_mfma_f32_32x32x2_f32 (exact f32) per comparison on the matrix cores
//   compare_wide2.hpp    k_compare_wide2  wide windows (32..128 rows): four or eight waves per comparison share the
//                          column transforms through LDS, row FFT over row pairs
//   compare_rows.hpp     k_compare_oddfft / k_compare_rows  odd image sizes: register FFT of odd length (3..25) over the
//                          reference layout, or direct column sums when N has no factor 3 or 5
//   compare_generic.hpp  k_compare_generic  same maths by direct pruned DFT (irregular wide displacement sets, N < 8)
//   compare_direct.hpp   k_c2r_cols/rows, k_compare_direct  BIOEM_CC_DIRECT=1 (BASELINE config 4): the cross-correlation as a
//                          sliding window in real space on the f32 matrix cores, no transform of the product
//   window_tiles.hpp     k_phase_shift, k_merge_tiles  wide windows no kernel covers: tiles of a window kernel
//   kernel_select.hpp    kernel table, k_compare_wide2 rules, plan_kernels: which kernel runs which shape
//   posterior.hpp        calc_logpro / calProb semantics (bioem_algorithm.h:18-142)
//   fold_kernels.hpp     k_fold_wave, k_fold_angles (k_fold: serial variant): fold the per-comparison partials into the probability block in the
//                          reference's (orientation, CTF) order (bioem_algorithm.h:96-123, bioem.cpp:1527-1600)
//   this file            device context, launch logic, the C ABI
//   kernels_*.hip        one translation unit per comparison-kernel family (the instantiations of kernel_table.inc),
//                          linked into the same library: kernels_fast, kernels_fastm, kernels_wide2_{short,16,long},
//                          kernels_odd; kernels_r2c: the fast r2c; this file keeps the generic kernel, the preparation,
//                          fold and merge kernels
//
// Numerics: float expressions that the reference evaluates in float are written in the same order and the
// file is compiled with -ffp-contract=off (FMAs only where fmaf() is spelled out).  Sums that the reference
// accumulates sequentially in float (sumsquareC, particle sums) are accumulated in the same order.
#define BIOEM_MAIN_TU 1 // the non-template kernels shared headers hold (k_posterior_consts) are compiled here only
#include "engine_types.hpp"

#include <rccl/rccl.h> // types only: the library itself is loaded on the first bioem_hip_merge (dlopen)

#include <dlfcn.h>

#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

namespace
{

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
  int Hp = 0;    // row-pair pitch of the comparison layout in 16-byte words (H, or H + 15: comparison_pitch)
  size_t Mc = 0; // float2 per image of the comparison layout, N * Hp
  int fast = 0, N1 = 0, winD = 0; // winD = template window half width used by the fast kernel
  int pchunk = 128;               // particle chunk of the fast kernel's block order (0 = all particles); measured:
                                  // 1 000 particles 6.73 -> 6.56 ms, 10 000 particles (2 GB, beyond the Infinity
                                  // Cache) 78.5 -> 63.2 ms per launch
  int gs = 1;                     // pixels per window row of the fast kernel (gcd of the displacement offsets)
  // wide windows (more than 31 offsets per axis): tilesPerAxis^2 launches of a tileT-row window (window_tiles.hpp)
  int genericWaves = 4; // waves per block of the generic kernel
  int genericRows = 0;  // ... and its window rows per pass through the LDS (0 = all)
  bool rowsK = false;   // k_compare_rows / k_compare_oddfft (odd N) instead of the generic kernel
  int oddR = 0;         // k_compare_oddfft: register-FFT length (3, 5, 9, 15, 25) dividing an odd N; 0 = direct sums
  int tileT = 0, tilesPerAxis = 1;
  std::vector<int> tileCenter, tileValid; // per axis tile: centre (in window rows) and number of rows inside the window
  int *dDispLocal = nullptr, *dTileCenter = nullptr, *dTileValid = nullptr, *dRankOfRow = nullptr;
  const void *fn = nullptr; // the comparison kernel of this handle (fast_kernel_t, kernel_select.hpp)
  size_t ldsBytes = 0;      // its dynamic LDS per block
  int family = 0;           // KernelFamily
  // k_compare_wide2 (compare_wide2.hpp): wide window in ONE launch per batch -- column transforms shared by the four
  // waves of a comparison, row FFT
  bool wide2 = false;
  bool fastm = false; // 23..31-row windows: k_compare_fastm (window pass on the matrix cores)
  bool fastm2 = false; // 33..47-row windows: k_compare_fastm2 (rows split over the half-waves, 3 x 3 matrix tiles)
  int w2NRW = 0, w2NBLK = 0, w2TS = 0, w2Rows2 = 0, nyqWD = 0, w2Halves = 1, w2NW = 4;
  float *dBtab = nullptr;  // k_compare_fastm2: tabulated B operand of the matrix pass
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
  double *dStamp = nullptr; // sphere footprints of the model (k_project_stamps), (2 iradMax + 1)^2 doubles per point
  double modelRadius = 0.;  // max |point| of the model, Angstrom (k_project_box)
  double quatNormDev = 0.;  // max | |q|^2 - 1 | over the uploaded quaternions (0 for Euler angles: always rotations)
  int nPts = 0;
  float NormDen = 0, pixelSize = 0;
  int shiftX = 0, shiftY = 0;
  int iradMax = 0; // widest sphere footprint of the model, pixels (k_project_bands)
  // BIOEM_CC_DIRECT=1 (BASELINE config 4): real-space particles, real-space conv maps of a launch, column pass of the c2r
  bool direct = false;
  float *dMapsReal = nullptr, *dConvReal = nullptr;
  double2 *dDirectZ = nullptr;
  int nCU = 256;   // compute units of the device: size of the resident grids of the preparation kernels
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
  // staged device entries (bioem_hip_project / _convolve / _compare_device): what each pipeline slot holds
  int stageO0[2] = {0, 0}, stageNO[2] = {0, 0}, stageC0[2] = {0, 0}, stageNC[2] = {0, 0};
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
  // per-batch phase records (bioem_hip_set_phase_timing): the reference's BIOEM_DEBUG_OUTPUT report (timer.cpp:138-165,
  // bioem.cpp:769-889) from HIP events on the streams the phases run on
  struct PhaseEvents
  {
    int phase, o0, o1, c0, c1;
    hipEvent_t a, b;
  };
  bool phaseTiming = false;
  std::vector<PhaseEvents> phasePending;
  std::vector<bioem_hip_phase_record> phaseDone;
  std::vector<hipEvent_t> evPool;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evPending;
  double compareMs = 0;
  long long launches = 0, comparisons = 0;

  std::string err;
};

#include "prep_kernels.hpp"
#include "dft_mfma.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#include "compare_wide2.hpp"
#include "compare_fastm.hpp"
#include "compare_fastm2.hpp"
#include "compare_generic.hpp"
#include "compare_rows.hpp"
#include "compare_direct.hpp"
#include "kernel_select.hpp"
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

// phase timing: an event pair around a phase of a batch on the stream it runs on (no-ops unless enabled)
int phase_begin(bioem_hip_ctx *h, hipStream_t st, int phase, int o0, int o1, int c0, int c1)
{
  if (!h->phaseTiming)
    return 0;
  bioem_hip_ctx::PhaseEvents e = {phase, o0, o1, c0, c1, get_event(h), get_event(h)};
  if (!e.a || !e.b)
  {
    h->err = "hipEventCreate failed";
    return 1;
  }
  HIP_CHECK(h, hipEventRecord(e.a, st));
  h->phasePending.push_back(e);
  return 0;
}

int phase_end(bioem_hip_ctx *h, hipStream_t st)
{
  if (!h->phaseTiming || h->phasePending.empty())
    return 0;
  HIP_CHECK(h, hipEventRecord(h->phasePending.back().b, st));
  return 0;
}

void drain_phases(bioem_hip_ctx *h)
{
  for (auto &e : h->phasePending)
  {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess)
    {
      bioem_hip_phase_record r = {e.phase, e.o0, e.o1, e.c0, e.c1, (double) ms * 1e-3};
      h->phaseDone.push_back(r);
    }
    h->evPool.push_back(e.a);
    h->evPool.push_back(e.b);
  }
  h->phasePending.clear();
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

// ids == nullptr: row oc of the launch is (orient0 + oc / convPerOrient, conv0 + oc % convPerOrient) (native path);
// otherwise ids[oc] = {orientation, CTF} and segs[0..nSeg) = runs of equal orientation (compat ring)
// the Nyquist column of the 64-column blocks (N / 2 a multiple of 64): its window rows by direct summation.  With 64
// particles or fewer four threads share a (particle, spectrum) pair (a block per 4 x 16 pairs), else one (16 x 16).
template <int Q>
void launch_nyquist_rows(bioem_hip_ctx *h, const CompareArgs &aw, int WD, int nOC)
{
  const int PB = 16 / Q;
  const dim3 gridq((unsigned) (((size_t) (h->nMaps + PB - 1) / PB) * ((nOC + 15) / 16)));
  switch (WD)
  {
#define X(W)                                                                                                            \
  case W:                                                                                                               \
    hipLaunchKernelGGL((k_nyquist_rows<W, Q>), gridq, dim3(256), 0, h->stream, aw);                                     \
    break;
    X(5) X(10) X(13) X(15) X(20) X(31) X(42)
#undef X
  }
}
void launch_nyquist(bioem_hip_ctx *h, const CompareArgs &aw, int WD, int nOC)
{
  if (h->nMaps <= 64)
    launch_nyquist_rows<4>(h, aw, WD, nOC);
  else
    launch_nyquist_rows<1>(h, aw, WD, nOC);
}

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
  a.btab = h->dBtab;
  a.tnyq = h->dTnyq;
  a.twnyq = h->dTwNyq;
  a.partials = h->dPartials;
  a.ldPart = h->maxOC;
  a.N = h->N;
  a.H = h->H;
  a.Hp = h->Hp;
  a.N1 = h->N1;
  a.nd = h->nd;
  a.maxD = h->pd.maxDisplaceCenter;
  a.nOC = nOC;
  a.nMaps = h->nMaps;
  a.algo = h->algo;
  a.pd = h->pd;
  a.gs = h->gs;
  a.ndx = a.ndy = h->nd;
  {
    // k_compare_fast / k_compare_fastm: a last column block of at most 32 columns is shared by the half-waves
    // (with the Nyquist split the blocks hold the H - 1 columns below N / 2: whole ones, or -- 192^2, 320^2, 448^2, which
    // were planned that way -- a last one of 32, and then the split is not optional)
    const int cols = h->nyq ? h->H - 1 : h->H;
    const int nblkF = (cols + 63) / 64, rem = cols - (nblkF - 1) * 64;
    a.split = h->fast && !h->fastm2 && !h->wide2 && !h->rowsK && rem <= 32 && h->N1 >= 2 &&
              (h->nyq || !getenv("BIOEM_NO_SPLIT_LAST"));
  }
  a.pchunk = h->pchunk > 0 ? std::min(h->pchunk, h->nMaps) : h->nMaps;
  if (a.pchunk >= 8) // a multiple of 8: a particle then stays on one XCD (65 particles: chunks of 64 + 1, 43.6 -> 45.3 M/s)
    a.pchunk &= ~7;
  const int ocGroups = (nOC + 3) / 4;
  // few particles: whole groups per XCD (fast_block_pair); the grid is padded to a multiple of 8 groups
  // (measured for 65...200 particles as well: 1-3 % slower than the chunk order there)
  const bool groupPerXcd = h->nMaps <= 64 && h->fast && !h->wide2 && !getenv("BIOEM_NO_GROUP_XCD");
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
  if (phase_begin(h, h->stream, BIOEM_HIP_PHASE_COMPARISON, ids ? -1 : orient0, ids ? -1 : orient0 + (nOC + convPerOrient - 1) / convPerOrient,
                  ids ? -1 : conv0, ids ? -1 : conv0 + convPerOrient))
    return 1;
  if (h->direct)
  {
    CompareArgs ad = a;
    ad.gs = h->pd.GridSpaceCenter;
    ad.maxD = h->pd.maxDisplaceCenter;
    hipLaunchKernelGGL(k_c2r_cols, dim3(h->H, nOC), dim3(128), sizeof(double2) * h->N, h->stream, bb.conv, h->N, h->H,
                       h->fast, h->N1, h->dTwD, h->dDirectZ);
    hipLaunchKernelGGL(k_c2r_rows, dim3(h->N, nOC), dim3(128), sizeof(double2) * h->H, h->stream, h->dDirectZ, h->N,
                       h->H, h->dTwD, h->dConvReal);
    const dim3 gridd((unsigned) ((size_t) nOC * ((h->nMaps + 31) / 32)));
    hipLaunchKernelGGL((k_compare_direct<3, 8>), gridd, dim3(512), direct_lds_bytes(h->N), h->stream, ad, h->dConvReal,
                       h->dMapsReal);
  }
  else if (h->wide2)
  {
    CompareArgs aw = a;
    aw.twk = h->dTwk2;
    aw.ts = h->w2TS;
    aw.nyqWD = h->nyqWD;
    if (h->nyq)
      launch_nyquist(h, aw, h->nyqWD == 20 || h->nyqWD == 31 ? h->nyqWD : 42, nOC);
    hipLaunchKernelGGL(reinterpret_cast<fast_kernel_t>(const_cast<void *>(h->fn)), dim3((unsigned) ((size_t) nOC * h->nMaps)), dim3(64 * h->w2NW), h->ldsBytes, h->stream, aw);
  }
  else if (h->fastm2)
  {
    CompareArgs aw = a;
    aw.twk = h->dTwk2;
    aw.nyqWD = h->nyqWD;
    if (h->nyq)
      launch_nyquist(h, aw, h->nyqWD == 20 ? 20 : 31, nOC);
    hipLaunchKernelGGL(reinterpret_cast<fast_kernel_t>(const_cast<void *>(h->fn)), grid, dim3(256), h->ldsBytes, h->stream, aw);
  }
  else if (h->fast || h->rowsK)
  {
    auto launch_window = [&](const CompareArgs &aw) {
      if (h->nyq)
        launch_nyquist(h, aw, h->winD == 5 || h->winD == 10 || h->winD == 13 ? h->winD : 15, nOC);
      hipLaunchKernelGGL(reinterpret_cast<fast_kernel_t>(const_cast<void *>(h->fn)), grid, dim3(256), h->ldsBytes, h->stream, aw);
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
    a.ts = h->genericRows;
    const dim3 gridg((unsigned) ((size_t) ((nOC + gw - 1) / gw) * h->nMaps));
    hipLaunchKernelGGL(reinterpret_cast<fast_kernel_t>(const_cast<void *>(h->fn)), gridg, dim3(64 * gw), h->ldsBytes, h->stream, a);
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
    if (h->nMaps <= 64 && nOC >= 1024)
      hipLaunchKernelGGL(k_fold_wave<4>, dim3(h->nMaps), dim3(256), 0, h->stream, h->dPartials, h->maxOC, nOC, h->nMaps,
                         bb.params, h->dSumRef, h->dDisp, h->nd, h->pd, orient0, conv0, convPerOrient, ids, pmap);
    else
      hipLaunchKernelGGL(k_fold_wave<1>, dim3((h->nMaps + 3) / 4), dim3(256), 0, h->stream, h->dPartials, h->maxOC, nOC,
                         h->nMaps, bb.params, h->dSumRef, h->dDisp, h->nd, h->pd, orient0, conv0, convPerOrient, ids,
                         pmap);
  }
  HIP_CHECK(h, hipGetLastError());
  if (phase_end(h, h->stream)) // comparison = the kernels of the launch and the fold behind them (what compareRefMaps does)
    return 1;
  if (h->evPending.size() > 512)
    drain_events(h);
  return 0;
}

// the exact-DFT r2c kernels keep one image row / column (and its twiddles) in dynamic LDS: 24 N + 16 and 32 N bytes.
// Above 64 KiB a launch needs the explicit opt-in; 32 N <= 160 KiB bounds the image size at 5120 pixels (the
// reference's MRC reader stops at 5000, mrc.h:128-133).
const int kMaxPixels = 5120;
// the exact-DFT kernels keep a row / column (and their intermediate) in LDS; up to kDftTwLds pixels the twiddle table, too
constexpr int kDftTwLds = 3072;
size_t dft_rows_lds(int N) { return sizeof(double) * ((N <= kDftTwLds ? 5 : 3) * (size_t) N + 2); }
size_t dft_cols_lds(int N) { return sizeof(double2) * (N <= kDftTwLds ? 3 : 2) * (size_t) N; }
void dft_split(int N, int &A, int &B)
{
  A = 1;
  for (int d = 1; d * d <= N; d++)
    if (N % d == 0)
      A = d;
  B = N / A;
}
// the fast transform (r2c_fft.hpp) wherever it has the lengths; BIOEM_R2C=dft keeps the exact-DFT kernels (A/B, tests)
bool r2c_use_fft(int N)
{
  const char *e = getenv("BIOEM_R2C");
  return bioem_r2c_fft_supported(N) && !(e && !strcmp(e, "dft"));
}
bool dft_use_mfma(int N)
{
  int A, B;
  dft_split(N, A, B);
  return dft_mfma_fits(N, A, B) && !getenv("BIOEM_DFT_VECTOR");
}
hipError_t dft_allow_lds(int N)
{
  hipError_t e;
  if (dft_use_mfma(N))
  {
    const void *fns[4] = {reinterpret_cast<const void *>(k_dft_rows_mfma<2>),
                          reinterpret_cast<const void *>(k_dft_rows_mfma<kDftMfmaTiles>),
                          reinterpret_cast<const void *>(k_dft_cols_mfma<2>),
                          reinterpret_cast<const void *>(k_dft_cols_mfma<kDftMfmaTiles>)};
    for (int f = 0; f < 4; f++)
    {
      e = hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int) (f < 2 ? dft_mfma_rows_lds(N) : dft_mfma_cols_lds(N)));
      if (e != hipSuccess)
        return e;
    }
    return hipSuccess;
  }
  if (N <= kDftTwLds)
  {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dft_rows<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int) dft_rows_lds(N));
    if (e != hipSuccess)
      return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_dft_cols<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int) dft_cols_lds(N));
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dft_rows<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int) dft_rows_lds(N));
  if (e != hipSuccess)
    return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k_dft_cols<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int) dft_cols_lds(N));
}

// k_convolve_sums keeps two tiles of terms for up to 20 (orientation, CTF) chains in dynamic LDS: 151 KiB
hipError_t conv_allow_lds()
{
  const void *fns[3] = {reinterpret_cast<const void *>(k_convolve_sums<3, 6>), reinterpret_cast<const void *>(k_convolve_sums<3, 5>),
                        reinterpret_cast<const void *>(k_convolve_sums<4, 4>)};
  for (const void *f : fns)
  {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int) conv_lanes_lds(kLaneRows));
    if (e != hipSuccess)
      return e;
  }
  return hipSuccess;
}

// r2c of nImg images (double projection maps scaled by NormDen / tempden, or float maps) into `out` (reference layout);
// rowSpec holds the row pass (N x H double2 per image).  dft_allow_lds(N) must have run on this device.
hipError_t launch_r2c(hipStream_t st, int nCU, const double *srcD, const float *srcF, const double *tempDen, float NormDen,
                      int N, int nImg, const double2 *tw, double2 *rowSpec, float2 *out, int lo = 0, int side = 0)
{
  side = side ? side : N; // (a box smaller than the map: the fast transform only, see project_batch)
  const int H = N / 2 + 1;
  int A, B;
  dft_split(N, A, B);
  if (r2c_use_fft(N))
    return bioem_r2c_fft_launch(st, nCU, srcD, srcF, tempDen, NormDen, N, nImg, tw, rowSpec, out, lo, side);
  if (dft_use_mfma(N))
  {
    // resident grids: as many blocks as the LDS of the device holds at once
    const int perCU = std::max(1, std::min(4, (int) (160 * 1024 / (dft_mfma_cols_lds(N) + 1024))));
    const int rowUnits = ((N + 15) / 16) * nImg, colUnits = ((H + 15) / 16) * nImg;
    if (A * ((B + 15) / 16) <= 2 * kDftMfmaWaves)
    {
      hipLaunchKernelGGL(k_dft_rows_mfma<2>, dim3(std::min(rowUnits, perCU * nCU)), dim3(kDftMfmaThreads),
                         dft_mfma_rows_lds(N), st, srcD, srcF, tempDen, NormDen, N, H, A, B, nImg, tw, rowSpec);
      hipLaunchKernelGGL(k_dft_cols_mfma<2>, dim3(std::min(colUnits, perCU * nCU)), dim3(kDftMfmaThreads),
                         dft_mfma_cols_lds(N), st, rowSpec, N, H, A, B, nImg, tw, out);
    }
    else
    {
      // registers hold one block of this variant per CU
      hipLaunchKernelGGL(k_dft_rows_mfma<kDftMfmaTiles>, dim3(std::min(rowUnits, nCU)), dim3(kDftMfmaThreads),
                         dft_mfma_rows_lds(N), st, srcD, srcF, tempDen, NormDen, N, H, A, B, nImg, tw, rowSpec);
      hipLaunchKernelGGL(k_dft_cols_mfma<kDftMfmaTiles>, dim3(std::min(colUnits, nCU)), dim3(kDftMfmaThreads),
                         dft_mfma_cols_lds(N), st, rowSpec, N, H, A, B, nImg, tw, out);
    }
  }
  else if (N <= kDftTwLds)
  {
    hipLaunchKernelGGL(k_dft_rows<true>, dim3(N, nImg), dim3(128), dft_rows_lds(N), st, srcD, srcF, tempDen, NormDen, N,
                       H, A, B, tw, rowSpec);
    hipLaunchKernelGGL(k_dft_cols<true>, dim3(H, nImg), dim3(256), dft_cols_lds(N), st, rowSpec, N, H, A, B, tw, out);
  }
  else
  {
    hipLaunchKernelGGL(k_dft_rows<false>, dim3(N, nImg), dim3(128), dft_rows_lds(N), st, srcD, srcF, tempDen, NormDen, N,
                       H, A, B, tw, rowSpec);
    hipLaunchKernelGGL(k_dft_cols<false>, dim3(H, nImg), dim3(256), dft_cols_lds(N), st, rowSpec, N, H, A, B, tw, out);
  }
  return hipGetLastError();
}

int run_r2c(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, const double *srcD, const float *srcF, int nImg,
            int lo = 0, int side = 0)
{
  HIP_CHECK(h, launch_r2c(st, h->nCU, srcD, srcF, bb.tempDen, h->NormDen, h->N, nImg, h->dTwD, bb.rowSpec, bb.specRef, lo,
                          side));
  return 0;
}

int project_batch(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, int o0, int nO)
{
  const int N = h->N;
  // the pixels a point of the model can reach in any orientation, with its footprint: a box around the map centre
  const double reach = h->modelRadius / (double) h->pixelSize;
  const int boxLo = std::max(0, (int) std::floor(N / 2.0 + 0.5 - reach) - 1 - h->iradMax - std::max(0, std::max(h->shiftX, h->shiftY)));
  const int boxHi = std::min(N - 1, (int) std::floor(N / 2.0 + 0.5 + reach) + 1 + h->iradMax - std::min(0, std::min(h->shiftX, h->shiftY)));
  const int boxSide = boxHi - boxLo + 1;
  // (orientations that stretch the model -- quaternions that are not of unit length -- take the band kernel: the box
  // has one pixel of margin; the matrix entries of a quaternion with |q|^2 = 1 + e are off by at most ~3 e, and half a
  // pixel is granted to that)
  const bool keepLength = h->quatNormDev * 4.0 * std::max(1.0, reach) < 0.5;
  if (h->dStamp && boxSide >= 1 && (size_t) boxSide * boxSide * sizeof(double) <= 52 * 1024 && N < 32768 &&
      keepLength && !getenv("BIOEM_PROJECT_BANDS") && !getenv("BIOEM_PROJECT_GLOBAL_ATOMICS"))
  {
    // with the fast r2c behind it the kernel stores the box alone and the transform skips everything outside it
    const int compact = r2c_use_fft(N) && !getenv("BIOEM_PROJECT_FULL_MAP");
    hipLaunchKernelGGL(k_project_box, dim3(std::min(nO, 3 * h->nCU)), dim3(256), sizeof(double) * boxSide * boxSide, st,
                       h->dPts, h->nPts, h->dAngles, o0, h->isQuat, N, h->pixelSize, h->shiftX, h->shiftY, h->iradMax,
                       h->dStamp, boxLo, boxSide, nO, compact, bb.projReal, bb.tempDen);
    HIP_CHECK(h, hipGetLastError());
    return compact ? run_r2c(h, bb, st, bb.projReal, nullptr, nO, boxLo, boxSide) : run_r2c(h, bb, st, bb.projReal, nullptr, nO);
  }
  HIP_CHECK(h, hipMemsetAsync(bb.tempDen, 0, sizeof(double) * nO, st)); // (k_project_box writes its sums itself)
  const int TR = 40960 / (8 * N); // rows of one LDS band: three blocks per CU
  // the record of every (orientation, point) borrows the row-pass buffer of the r2c that follows
  const bool coordsFit = (size_t) h->nPts * sizeof(ProjectRecord) <= (size_t) N * h->H * sizeof(double2);
  if (TR >= 12 && h->iradMax <= 16 && h->dStamp && N < 32768 && coordsFit && !getenv("BIOEM_PROJECT_GLOBAL_ATOMICS"))
  {
    ProjectRecord *coords = reinterpret_cast<ProjectRecord *>(bb.rowSpec);
    hipLaunchKernelGGL(k_project_coords, dim3((h->nPts + 255) / 256, nO), dim3(256), 0, st, h->dPts, h->nPts, h->dAngles,
                       o0, h->isQuat, N, h->pixelSize, h->shiftX, h->shiftY, coords);
    const int units = ((N + TR - 1) / TR) * nO;
    hipLaunchKernelGGL(k_project_bands, dim3(std::min(units, 3 * h->nCU)), dim3(256), sizeof(double) * TR * N, st, coords,
                       h->nPts, nO, N, TR, h->iradMax, h->dStamp, bb.projReal, bb.tempDen);
  }
  else
  {
    HIP_CHECK(h, hipMemsetAsync(bb.projReal, 0, sizeof(double) * (size_t) nO * N * N, st));
    hipLaunchKernelGGL(k_project, dim3((h->nPts + 255) / 256, nO), dim3(256), 0, st, h->dPts, h->nPts, h->dAngles, o0,
                       h->isQuat, N, h->pixelSize, h->shiftX, h->shiftY, bb.projReal, bb.tempDen);
  }
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
                     h->fast, h->N1, h->Hp);
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
  // few particles: the preparation is the longer half of the pipeline and the fused kernel shortens it; many particles:
  // it hides behind the comparison either way, and the one-wave blocks of k_parseval_ordered take less from the
  // comparison kernel than the 16-wave blocks of k_convolve_sums (round 4, the fused kernel with 3-4 orientations per
  // block: 65...200 particles +1...3 %, 400 particles equal, 1 000 particles 52.5 against 53.4 M comparisons/s)
  const char *fe = getenv("BIOEM_CONVOLVE_FUSED");
  if (fe ? atoi(fe) != 0 : h->nMaps <= 256)
  {
    // chains on the lanes of the adding wave: 4 orientations x up to 4 CTFs, 3 x 5, or 3 x 6 per block (16, 15, 18
    // products per producing thread and tile: more, and the producers -- ~25 vector instructions per product -- take
    // longer than the 960 additions)
    if (nC <= 4)
      hipLaunchKernelGGL((k_convolve_sums<4, 4>), dim3(1, (nO + 3) / 4), dim3(kConvThreads), conv_lanes_lds(4 * nC), st,
                         bb.specRef, h->dCTF, h->dCtfParam, h->N, h->H, h->fast, h->N1, c0, nC, nO, 4 * nC, bb.conv, bb.params, h->Hp);
    else if (nC == 5)
      hipLaunchKernelGGL((k_convolve_sums<3, 5>), dim3(1, (nO + 2) / 3), dim3(kConvThreads), conv_lanes_lds(15), st, bb.specRef,
                         h->dCTF, h->dCtfParam, h->N, h->H, h->fast, h->N1, c0, nC, nO, 15, bb.conv, bb.params, h->Hp);
    else
      hipLaunchKernelGGL((k_convolve_sums<3, 6>), dim3((nC + 5) / 6, (nO + 2) / 3), dim3(kConvThreads), conv_lanes_lds(18), st,
                         bb.specRef, h->dCTF, h->dCtfParam, h->N, h->H, h->fast, h->N1, c0, nC, nO, 18, bb.conv, bb.params, h->Hp);
    HIP_CHECK(h, hipGetLastError());
    return 0;
  }
  const int M4 = (int) ((h->M + 3) & ~(size_t) 3);
  hipLaunchKernelGGL(k_convolve, dim3(nC, nO), dim3(256), 0, st, bb.specRef, h->dCTF, h->dCtfParam, h->N, h->H,
                     h->fast, h->N1, c0, bb.conv, bb.scratch, M4, bb.params, h->Hp);
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
  HIP_CHECK(h, hipDeviceGetAttribute(&h->nCU, hipDeviceAttributeMultiprocessorCount, device));
  HIP_CHECK(h, dft_allow_lds(N));
  HIP_CHECK(h, conv_allow_lds());
  if (getenv("BIOEM_PREP_OCCUPANCY"))
  { // blocks per CU the runtime grants the preparation kernels at this image size
    int nb = 0;
    const int TRo = std::max(1, 40960 / (8 * N));
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_project_bands, 256, sizeof(double) * TRo * N);
    fprintf(stderr, "k_project_bands: %d blocks/CU (dynamic LDS %zu B)\n", nb, sizeof(double) * TRo * N);
    if (dft_use_mfma(N))
    {
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_dft_rows_mfma<2>, kDftMfmaThreads, dft_mfma_rows_lds(N));
      fprintf(stderr, "k_dft_rows_mfma<2>: %d blocks/CU (dynamic LDS %zu B)\n", nb, dft_mfma_rows_lds(N));
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_dft_cols_mfma<2>, kDftMfmaThreads, dft_mfma_cols_lds(N));
      fprintf(stderr, "k_dft_cols_mfma<2>: %d blocks/CU (dynamic LDS %zu B)\n", nb, dft_mfma_cols_lds(N));
    }
  }
  {
    int prLow = 0, prHigh = 0;
    HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    HIP_CHECK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prHigh));
  }

  // which kernel runs this shape: kernel_select.hpp (pure function of N and the displacement set)
  const int maxD = pd->maxDisplaceCenter;
  {
    KernelPlan P = plan_kernels(N, maxD, pd->GridSpaceCenter, algo);
    if (P.err || !P.fn)
    {
      h->err = P.err ? P.err : "no comparison kernel for this configuration";
      return 2;
    }
    h->disp = P.disp;
    h->nd = P.nd;
    h->gs = P.gs;
    h->winD = P.winD;
    h->family = P.family;
    h->fast = P.fast;
    h->N1 = P.N1;
    h->oddR = P.oddR;
    h->nyq = P.nyq;
    h->fastm = P.fastm;
    h->fastm2 = P.fastm2;
    h->rowsK = P.rowsK;
    h->wide2 = P.wide2;
    h->tileT = P.tileT;
    h->tilesPerAxis = P.tilesPerAxis;
    h->tileCenter = P.tileCenter;
    h->tileValid = P.tileValid;
    h->w2NRW = P.w2NRW;
    h->w2NBLK = P.w2NBLK;
    h->w2TS = P.w2TS;
    h->w2Rows2 = P.w2Rows2;
    h->nyqWD = P.nyqWD;
    h->w2Halves = P.w2Halves;
    h->w2NW = P.w2NW;
    h->genericWaves = P.genericWaves;
    h->genericRows = P.genericRows;
    h->fn = reinterpret_cast<const void *>(P.fn);
    h->ldsBytes = P.ldsBytes;
    HIP_CHECK(h, hipFuncSetAttribute(h->fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int) h->ldsBytes));
  }
  const int mD = maxD / h->gs;
  // Pitch of the comparison layout.  A lane block of the fast kernels keeps eight rows of one 64-column block in flight;
  // with N a multiple of 64 a row pair is 16 B more than a multiple of 512 B (4 112 B at 512^2, 2 064 B at 256^2) and all
  // of them start in the same few L2 channels.  Fifteen more words make the pitch an odd number of 256-byte lines
  // (timing build first: 512^2 8.3 -> 10.1 M/s; the real thing, same box, +-5 / +-10 px: 512^2 +20 %, 384^2 +6.7 %, 448^2
  // +7 / +4 %, 256^2 +3 %, 192^2 and 320^2 +2...3.5 %, +0...6 % on the matrix-core families; the headline shape, whose
  // kernel reads one argument more, 54.74 against 54.68 M/s).  64^2 and 128^2 gain nothing, 128^2 loses 1...3 % on
  // few-particle jobs: not padded.  k_compare_wide2 (operands once per comparison) gains nothing: its plans, the tiled
  // ones, odd sizes and the direct kernel keep Hp = H.
  // (BIOEM_PITCH_PAD: another number of words.  7 / 15 / 23 / 31 / 47 are within 2 % of each other at 256^2 ... 512^2,
  // profiles/r04_padded_pitch_ab.txt.)
  h->Hp = h->H;
  const bool directCC = getenv("BIOEM_CC_DIRECT") && atoi(getenv("BIOEM_CC_DIRECT")) != 0; // (its own kernels read conv)
  if (h->fast && !h->wide2 && !h->rowsK && !h->tileT && !directCC && N % 64 == 0 && N >= 192 &&
      !getenv("BIOEM_NO_PITCH_PAD"))
    h->Hp = h->H + (getenv("BIOEM_PITCH_PAD") ? atoi(getenv("BIOEM_PITCH_PAD")) : 15);
  h->Mc = (size_t) N * h->Hp;

  // batch sizing: conv buffer <= ~96 MiB, partial buffer <= ~128 MiB
  const size_t M = (size_t) h->M;
  size_t ocCap = (96u << 20) / (h->Mc * sizeof(float2));
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
      ocCap = std::min(partCap, (size_t) (1024u << 20) / (h->Mc * sizeof(float2)));
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

  HIP_CHECK(h, hipMalloc(&h->dRef, sizeof(float2) * h->Mc * nMaps));
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
  HIP_CHECK(h, hipMalloc(&h->dConv, sizeof(float2) * (size_t) h->maxOC * h->Mc));
  HIP_CHECK(h, hipMalloc(&h->dParams, sizeof(bioem_hip_param5) * h->maxOC));
  HIP_CHECK(h, hipMalloc(&h->dPostC, sizeof(double2) * h->maxOC));
  HIP_CHECK(h, hipMalloc(&h->dPartials, sizeof(Partial) * (size_t) nMaps * h->maxOC));
  if (const char *de = getenv("BIOEM_CC_DIRECT"))
    h->direct = atoi(de) != 0;
  if (h->direct)
  { // compare_direct.hpp: the sliding-window evaluation of the same cross-correlation values
    const int gsd = pd->GridSpaceCenter;
    if (N > kDirectMaxN || gsd < 1 || maxD % gsd != 0 || 2 * (maxD / gsd) + 1 > 24 || h->tileT)
    {
      h->err = "BIOEM_CC_DIRECT: the direct cross-correlation takes images up to 160 pixels and regular windows of at "
               "most 24 offsets per axis";
      return 2;
    }
    HIP_CHECK(h, hipMalloc(&h->dMapsReal, sizeof(float) * (size_t) nMaps * N * N));
    HIP_CHECK(h, hipMemset(h->dMapsReal, 0, sizeof(float) * (size_t) nMaps * N * N));
    HIP_CHECK(h, hipMalloc(&h->dConvReal, sizeof(float) * (size_t) h->maxOC * N * N));
    HIP_CHECK(h, hipMalloc(&h->dDirectZ, sizeof(double2) * (size_t) h->maxOC * M));
    HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_compare_direct<3, 8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) direct_lds_bytes(N)));
  }
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
  }
  h->devProbBytes = bioem_hip_prob_size(nMaps, angO1 - angO0, pd->writeAngles);
  h->probBytes = shard ? bioem_hip_prob_size(nMaps, 0, 0) : h->devProbBytes;
  HIP_CHECK(h, hipMalloc(&h->dProb, h->devProbBytes));
  {
    // projection/convolution are filler work with many particles: lowest priority so that comparison blocks win the
    // CUs.  With few particles they are the longer half of the pipeline and waiting behind every comparison block
    // stretches them four- to sixfold: same priority as the comparison then (20 particles: 8.41 -> 8.25 ms per pass;
    // 1 000 particles: no difference either way).  BIOEM_PREP_PRIORITY=low|high overrides.
    int prLow = 0, prHigh = 0;
    HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    bool prepHigh = nMaps <= 64;
    if (const char *e = getenv("BIOEM_PREP_PRIORITY"))
      prepHigh = e[0] == 'h';
    HIP_CHECK(h, hipStreamCreateWithPriority(&h->prepStream, hipStreamNonBlocking, prepHigh ? prHigh : prLow));
  }
  HIP_CHECK(h, hipMalloc(&h->dProjReal2, sizeof(double) * (size_t) h->OB * N * N));
  HIP_CHECK(h, hipMalloc(&h->dTempDen2, sizeof(double) * h->OB));
  HIP_CHECK(h, hipMalloc(&h->dRowSpec2, sizeof(double2) * (size_t) h->OB * M));
  HIP_CHECK(h, hipMalloc(&h->dSpecRef2, sizeof(float2) * (size_t) h->OB * M));
  HIP_CHECK(h, hipMalloc(&h->dScratch2, sizeof(float) * (size_t) ((h->maxOC + 31) & ~31) * ((M + 3) & ~(size_t) 3)));
  HIP_CHECK(h, hipMalloc(&h->dConv2, sizeof(float2) * (size_t) h->maxOC * h->Mc));
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
  if (h->fastm2)
  { // recombination twiddles of k_compare_fastm2<R, NYQ, GS>: [k1 pair s][accumulator a = OFF g + j] = {w^(dx 2s), w^(dx (2s+1))}
    // for the LOW-half row dx = (2 OFF g + j - 23) GS (the high half folds the rows OFF further with the same numbers); zero
    // for a k1 beyond N1 - 1 (rows outside the displacement list are masked by their rank in the kernel)
    const int Rl = 2 * h->fast, off = fastm2_off(Rl, h->gs), nAcc = fastm2_acc(Rl, h->gs);
    const int nS = (h->N1 + 1) / 2;
    std::vector<float4> t4((size_t) nS * nAcc, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int s2 = 0; s2 < nS; s2++)
      for (int ac = 0; ac < nAcc; ac++)
      {
        const long long dx = (long long) (2 * off * (ac / off) + (ac % off) - kFm2WD) * h->gs;
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < 2; e++)
        {
          const int k1 = 2 * s2 + e;
          if (k1 >= h->N1)
            continue;
          const double ang = 2.0 * M_PI * (double) (((dx * k1) % N + N) % N) / (double) N;
          w[2 * e] = (float) cos(ang);
          w[2 * e + 1] = (float) sin(ang);
        }
        t4[(size_t) s2 * nAcc + ac] = make_float4(w[0], w[1], w[2], w[3]);
      }
    HIP_CHECK(h, hipMalloc(&h->dTwk2, sizeof(float4) * t4.size()));
    HIP_CHECK(h, hipMemcpy(h->dTwk2, t4.data(), sizeof(float4) * t4.size(), hipMemcpyHostToDevice));
    // B operand of the matrix pass: lane l of k-step K, column tile ct supplies (l / 16 odd ? sin : cos)(2 pi ky dy / N),
    // ky = 2 K + l / 32, dy = (16 ct + l % 16 - 23) GS -- the float twiddles exp(2 pi i k / N) every kernel uses
    std::vector<float> bt(fastm2_btab_floats(h->H, h->nyq));
    for (size_t K = 0; K < bt.size() / 192; K++)
      for (int ct = 0; ct < 3; ct++)
        for (int l = 0; l < 64; l++)
        {
          const long long ky = 2 * (long long) K + (l >> 5), dy = (long long) (16 * ct + (l & 15) - kFm2WD) * h->gs;
          const double ang = 2.0 * M_PI * (double) (((ky * dy) % N + N) % N) / (double) N;
          bt[(K * 3 + ct) * 64 + l] = ((l >> 4) & 1) ? (float) sin(ang) : (float) cos(ang);
        }
    HIP_CHECK(h, hipMalloc(&h->dBtab, sizeof(float) * bt.size()));
    HIP_CHECK(h, hipMemcpy(h->dBtab, bt.data(), sizeof(float) * bt.size(), hipMemcpyHostToDevice));
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
        fprintf(f, "k_nyquist_rows<%d, %d>\n", (h->wide2 || h->fastm2) ? h->nyqWD : h->winD, h->nMaps <= 64 ? 4 : 1);
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
  drain_phases(h);
  for (hipEvent_t e : h->evPool)
    hipEventDestroy(e);
  void *ptrs[] = {h->dRef,     h->dSumRef,  h->dSumsqRef, h->dCTF,     h->dCtfParam, h->dPts,   h->dAngles,
                  h->dTw,      h->dTwD,     h->dDisp,     h->dLtab,    h->dTwk,     h->dProjReal, h->dTempDen,  h->dRowSpec, h->dSpecRef,
                  h->dScratch, h->dConv,    h->dParams,   h->dPartials, h->dProb,
                  h->dProjReal2, h->dTempDen2, h->dRowSpec2, h->dSpecRef2, h->dScratch2, h->dConv2, h->dParams2,
                  h->dTnyq, h->dTwNyq, h->dPartTiles, h->dConvShift, h->dDispLocal, h->dRankOfRow, h->dTileCenter, h->dTileValid,
                  h->dCand, h->dSend, h->dRecv, h->dMerged, h->dTwk2, h->dBtab, h->dPostC, h->dPostC2,
                  h->dMapsReal, h->dConvReal, h->dDirectZ, h->dStamp};
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
  if (h->direct)
  {
    h->err = "BIOEM_CC_DIRECT needs the particle images: bioem_hip_upload_particle_maps";
    return 2;
  }
  const size_t M = (size_t) h->M;
  HIP_CHECK(h, hipMemcpyAsync(h->dSumRef, sum, sizeof(float) * h->nMaps, hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipMemcpyAsync(h->dSumsqRef, sumsq, sizeof(float) * h->nMaps, hipMemcpyHostToDevice, h->stream));
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    HIP_CHECK(h, hipMemcpyAsync(h->dSpecRef, refFFT + 2 * M * (size_t) b, sizeof(float2) * M * n,
                                hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_reorder, dim3(1024), dim3(256), 0, h->stream, h->dSpecRef, h->dRef + h->Mc * (size_t) b, n, h->N,
                       h->H, h->fast, h->N1, h->Hp);
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
    if (h->direct)
      HIP_CHECK(h, hipMemcpyAsync(h->dMapsReal + (size_t) b * N * N, dMaps, sizeof(float) * (size_t) n * N * N,
                                  hipMemcpyDeviceToDevice, h->stream));
    if (run_r2c(h, batch_buf(h, 0), h->stream, nullptr, dMaps, n))
    {
      hipFree(dMaps);
      return 1;
    }
    hipLaunchKernelGGL(k_reorder, dim3(1024), dim3(256), 0, h->stream, h->dSpecRef, h->dRef + h->Mc * (size_t) b, n, N,
                       h->H, h->fast, h->N1, h->Hp);
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
  h->iradMax = 0;
  for (int n = 0; n < nPts; n++)
    if (pts[n].radius > pixelSize)
      h->iradMax = std::max(h->iradMax, (int) (pts[n].radius / pixelSize) + 1);
  h->modelRadius = 0.;
  for (int n = 0; n < nPts; n++)
    h->modelRadius = std::max(h->modelRadius, std::sqrt((double) pts[n].pos[0] * pts[n].pos[0] + (double) pts[n].pos[1] * pts[n].pos[1] +
                                                        (double) pts[n].pos[2] * pts[n].pos[2]));
  if (h->dStamp)
    hipFree(h->dStamp);
  h->dStamp = nullptr;
  const size_t cells = (size_t) nPts * (2 * h->iradMax + 1) * (2 * h->iradMax + 1);
  if (h->iradMax <= 16 && nPts > 0 && cells <= ((size_t) 1 << 27)) // at most 1 GiB of footprints, else k_project
  {
    HIP_CHECK(h, hipMalloc(&h->dStamp, sizeof(double) * cells));
    hipLaunchKernelGGL(k_project_stamps, dim3((unsigned) ((cells + 255) / 256)), dim3(256), 0, h->stream, h->dPts, nPts,
                       h->iradMax, pixelSize, h->dStamp);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
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
  // k_project_box relies on the rotated model staying inside its box: the reference's quaternion matrix
  // (bioem.cpp:1632-1646) is a rotation only for unit quaternions; project_batch holds the list's largest deviation
  // against the model's extent and sends a list that stretches the model by a fraction of a pixel to the band kernel
  h->quatNormDev = 0.;
  if (isQuat)
    for (int k = 0; k < n; k++)
    {
      const float *q = angles4 + 4 * (size_t) k;
      const double n2 = (double) q[0] * q[0] + (double) q[1] * q[1] + (double) q[2] * q[2] + (double) q[3] * q[3];
      const double dev = std::fabs(n2 - 1.0);
      h->quatNormDev = dev == dev ? std::max(h->quatNormDev, dev) : 1e30; // (a NaN never passes)
    }
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
  // orientations per batch: the handle's capacity, but at least six batches per call where that leaves 64 or more per
  // batch -- the preparation of batch b+1 hides behind the comparison of batch b, the first batch's does not
  // (20 particles x 2 304 orientations in 3 batches of 1 060: 11.5 ms per pass, a third of it the exposed first batch;
  // a ramp -- first batches a quarter and a half of the rest -- was measured too: 8.9 against 8.5 ms, more launches
  // cost more than the shorter exposed preparation saves)
  // ... as long as a batch still compares 33 000 ... 65 000 pairs (a job of 23 000 pairs -- BASELINE config 1 -- is one batch:
  // 0.46 ms against 0.57 ms in two)
  // (round 4, with the preparation 2.5x faster than when 32 768 was chosen: 65 536 pairs per batch at least for small
  // images -- 128^2 x 10 particles x 4 608 orientations 63.7 -> 66.2 M/s, x 1 152: 52.5 -> 57.5; 131 072 the same,
  // 16 384 and less lose; at 224^2 32 768 stays: 10 particles 28.6 against 27.3 M/s with 65 536)
  const long long minPairs = getenv("BIOEM_MIN_BATCH_PAIRS") ? atoll(getenv("BIOEM_MIN_BATCH_PAIRS")) : (h->N <= 160 ? 65536 : 32768);
  const int perBatch = (int) std::min<long long>(h->OB, (minPairs + (long long) nC * h->nMaps - 1) / ((long long) nC * h->nMaps));
  const int OBc = std::min(h->OB, std::max(std::max(64, perBatch), (iOrientEnd - iOrientBegin + 5) / 6));
  // (Round 4 measured batches that shrink towards the END as well -- a half, a quarter of the regular size, so that
  // the last comparison, which nothing overlaps, is short: 7.12 -> 7.66 ms per pass at 20 particles, 0.58 -> 0.74 ms
  // for the config-1 shape.  Equal batches it is.)
  std::vector<int> first; // first orientation of every batch, and the end
  for (int o = iOrientBegin; o < iOrientEnd; o += OBc)
    first.push_back(o);
  first.push_back(iOrientEnd);
  const int nb = (int) first.size() - 1;
  auto prep = [&](int b) -> int {
    const int slot = b & 1;
    const int o0 = first[b];
    const int nO = first[b + 1] - o0;
    const BatchBuf bb = batch_buf(h, slot);
    if (h->cmpPending[slot])
    {
      HIP_CHECK(h, hipStreamWaitEvent(h->prepStream, h->cmpDone[slot], 0));
      h->cmpPending[slot] = false;
    }
    if (phase_begin(h, h->prepStream, BIOEM_HIP_PHASE_PROJECTION, o0, o0 + nO, 0, 0) || project_batch(h, bb, h->prepStream, o0, nO) ||
        phase_end(h, h->prepStream))
      return 1;
    if (phase_begin(h, h->prepStream, BIOEM_HIP_PHASE_CONVOLUTION, o0, o0 + nO, iConvBegin, iConvBegin + nC) ||
        convolve_batch(h, bb, h->prepStream, nO, iConvBegin, nC) || phase_end(h, h->prepStream))
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
    const int o0 = first[b];
    const int nO = first[b + 1] - o0;
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

// ---- the three stages of the loop body as separate, asynchronous, batched entries (device-resident hand-over) ----
int bioem_hip_project(bioem_hip_handle h, int iPipeline, int iOrientBegin, int iOrientEnd)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const int slot = iPipeline & 1, nO = iOrientEnd - iOrientBegin;
  if (!h->dPts || iOrientBegin < 0 || iOrientEnd > h->nAnglesUp || nO < 1)
  {
    h->err = "bioem_hip_project: model/orientations not uploaded or range invalid";
    return 2;
  }
  if (iOrientBegin < h->angO0 || iOrientEnd > h->angO1)
  {
    h->err = "bioem_hip_project: orientations outside the range this shard handle was created for";
    return 2;
  }
  if (nO > h->OB)
  {
    char buf[160];
    snprintf(buf, sizeof(buf), "bioem_hip_project: at most %d orientations per call with this configuration", h->OB);
    h->err = buf;
    return 2;
  }
  if (compat_flush(h)) // rows staged through the reference-compatible entry go first (call order)
    return 1;
  // the comparison that last read this buffer set goes first (NOT the other set's: that one is what this call overlaps);
  // with none queued, whatever the main stream holds so far (start_run, debug hooks, the compat entry)
  if (!h->cmpPending[slot])
    HIP_CHECK(h, hipEventRecord(h->cmpDone[slot], h->stream));
  HIP_CHECK(h, hipStreamWaitEvent(h->prepStream, h->cmpDone[slot], 0));
  h->cmpPending[slot] = false;
  h->stageNO[slot] = h->stageNC[slot] = 0;
  if (phase_begin(h, h->prepStream, BIOEM_HIP_PHASE_PROJECTION, iOrientBegin, iOrientEnd, 0, 0) ||
      project_batch(h, batch_buf(h, slot), h->prepStream, iOrientBegin, nO) || phase_end(h, h->prepStream))
    return 1;
  h->stageO0[slot] = iOrientBegin;
  h->stageNO[slot] = nO;
  return 0;
}

int bioem_hip_convolve(bioem_hip_handle h, int iPipeline, int iConvBegin, int iConvEnd)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const int slot = iPipeline & 1, nC = iConvEnd - iConvBegin, nO = h->stageNO[slot];
  if (nO < 1)
  {
    h->err = "bioem_hip_convolve: no projections in this pipeline slot (call bioem_hip_project first)";
    return 2;
  }
  if (iConvBegin < 0 || iConvEnd > h->nCTF || nC < 1 || (long long) nO * nC > h->maxOC)
  {
    char buf[200];
    snprintf(buf, sizeof(buf), "bioem_hip_convolve: CTF range invalid or more than %d (orientation, CTF) rows per call", h->maxOC);
    h->err = buf;
    return 2;
  }
  if (h->cmpPending[slot])
  { // a comparison of this slot's previous conv rows is still queued: it reads what this call overwrites
    HIP_CHECK(h, hipStreamWaitEvent(h->prepStream, h->cmpDone[slot], 0));
    h->cmpPending[slot] = false;
  }
  if (phase_begin(h, h->prepStream, BIOEM_HIP_PHASE_CONVOLUTION, h->stageO0[slot], h->stageO0[slot] + nO, iConvBegin, iConvEnd) ||
      convolve_batch(h, batch_buf(h, slot), h->prepStream, nO, iConvBegin, nC) || phase_end(h, h->prepStream))
    return 1;
  HIP_CHECK(h, hipEventRecord(h->prepDone[slot], h->prepStream));
  h->stageC0[slot] = iConvBegin;
  h->stageNC[slot] = nC;
  return 0;
}

int bioem_hip_compare_device(bioem_hip_handle h, int iPipeline)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const int slot = iPipeline & 1, nO = h->stageNO[slot], nC = h->stageNC[slot];
  if (nO < 1 || nC < 1)
  {
    h->err = "bioem_hip_compare_device: no conv spectra in this pipeline slot (call bioem_hip_project and bioem_hip_convolve first)";
    return 2;
  }
  HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->prepDone[slot], 0));
  if (launch_compare_fold(h, batch_buf(h, slot), nO * nC, h->stageO0[slot], h->stageC0[slot], nC))
    return 1;
  HIP_CHECK(h, hipEventRecord(h->cmpDone[slot], h->stream));
  h->cmpPending[slot] = true;
  return 0;
}

int bioem_hip_max_batch(bioem_hip_handle h, int *maxOrientations, int *maxRows)
{
  if (!h)
    return 2;
  if (maxOrientations)
    *maxOrientations = h->OB;
  if (maxRows)
    *maxRows = h->maxOC;
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
  drain_phases(h);
  return 0;
}

int bioem_hip_set_phase_timing(bioem_hip_handle h, int on)
{
  if (!h)
    return 2;
  h->phaseTiming = on != 0;
  return 0;
}

int bioem_hip_phase_records(bioem_hip_handle h, bioem_hip_phase_record *out, int cap, int *n)
{
  if (!h || !n || cap < 0 || (cap > 0 && !out))
    return 2;
  HIP_CHECK(h, hipSetDevice(h->device));
  drain_phases(h);
  const int have = (int) h->phaseDone.size();
  *n = have;
  for (int i = 0; i < have && i < cap; i++)
    out[i] = h->phaseDone[i];
  if (cap >= have)
    h->phaseDone.clear();
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
  hipLaunchKernelGGL(k_unreorder, dim3(256), dim3(256), 0, h->stream, h->dConv + h->Mc * (size_t) iConv, tmp, 1, h->N,
                     h->H, h->fast, h->N1, h->Hp);
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
    hipLaunchKernelGGL(k_unreorder, dim3(1024), dim3(256), 0, h->stream, h->dRef + h->Mc * (size_t) b, h->dSpecRef, n, h->N,
                       h->H, h->fast, h->N1, h->Hp);
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

int bioem_hip_plan(int numberPixels, int maxDisplaceCenter, int gridSpaceCenter, int algo, char *signature, int cap)
{
  if (numberPixels < 2 || numberPixels > kMaxPixels || maxDisplaceCenter < 0 || gridSpaceCenter < 1 ||
      maxDisplaceCenter >= numberPixels / 2 || !signature || cap < 1)
    return 2;
  const KernelPlan P = plan_kernels(numberPixels, maxDisplaceCenter, gridSpaceCenter, algo);
  if (P.err || !P.fn)
    return 1;
  plan_signature(P, signature, (size_t) cap);
  if (P.tileT)
  {
    const size_t n = strlen(signature);
    snprintf(signature + n, (size_t) cap - n, " x %d^2 tiles of %d rows", P.tilesPerAxis, P.tileT);
  }
  return 0;
}

const char *bioem_hip_kernel_name(bioem_hip_handle h)
{
  if (!h)
    return "";
  if (h->direct)
    return "k_compare_direct";
  if (h->wide2)
    return "k_compare_wide2";
  if (h->fastm2)
    return "k_compare_fastm2";
  if (h->fastm)
    return "k_compare_fastm";
  if (h->fast)
    return "k_compare_fast";
  return h->rowsK ? (h->oddR ? "k_compare_oddfft" : "k_compare_rows") : "k_compare_generic";
}

const char *bioem_hip_kernel_signature(bioem_hip_handle h)
{
  if (!h)
    return "";
  static thread_local char buf[96];
  const char *nq = h->nyq ? "true" : "false";
  if (h->direct)
    snprintf(buf, sizeof(buf), "k_compare_direct<3, 8>");
  else if (h->wide2)
    if (h->w2Halves == 2)
      snprintf(buf, sizeof(buf), h->w2NW == 8 ? "k_compare_wide2<%d, %d, %d, %s, 2, 8>" : "k_compare_wide2<%d, %d, %d, %s, 2>",
               2 * h->fast, h->w2NRW, h->w2NBLK, nq);
    else
      snprintf(buf, sizeof(buf), h->w2NW == 8 ? "k_compare_wide2<%d, %d, %d, %s, 1, 8>" : "k_compare_wide2<%d, %d, %d, %s>",
               2 * h->fast, h->w2NRW, h->w2NBLK, nq);
  else if (h->fastm2)
    snprintf(buf, sizeof(buf), "k_compare_fastm2<%d, %s, %d>", 2 * h->fast, nq, h->gs);
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
  int nCU = 256;
  hipDeviceGetAttribute(&nCU, hipDeviceAttributeMultiprocessorCount, device);
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
    if (launch_r2c(0, nCU, nullptr, dIn, nullptr, 1.f, N, nImg, dTw, dRow, dOut) == hipSuccess &&
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

