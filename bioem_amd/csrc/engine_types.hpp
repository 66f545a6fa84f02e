// engine_types.hpp -- what every translation unit of libbioem_hip.so shares: the HIP runtime, the C ABI types, the
// per-comparison partial and the kernel registry entry.  (Round 4: one translation unit per kernel family, linked
// into the same library, so that a one-line change rebuilds one family and `make -j` uses the cores.)
#ifndef BIOEM_ENGINE_TYPES_HPP
#define BIOEM_ENGINE_TYPES_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bioem_hip.h"

#define MIN_PROB (-999999.)

// one line of the kernel registry (kernel_select.hpp): family, template arguments, the kernel's host stub.  The stub is
// kept as an untyped pointer because CompareArgs lives in every translation unit's own anonymous namespace (same
// layout everywhere: compare_args.hpp); the launcher casts it back.
struct BioemKernelEntry
{
  int family;
  int a[6];
  const void *fn;
};

// the family tables, one per translation unit (kernels_*.hip); hidden: not part of the C ABI
#define BIOEM_HIDDEN __attribute__((visibility("hidden")))
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_fast(int *n);
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_fastm(int *n);
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_fastm2(int *n);
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_wide2_short(int *n); // register FFTs of 8..12 points
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_wide2_16(int *n);
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_wide2_long(int *n);  // 20..32 points
BIOEM_HIDDEN const BioemKernelEntry *bioem_kernels_odd(int *n);         // k_compare_rows, k_compare_oddfft

// the fast r2c (kernels_r2c.hip): image sizes N = A * B with both factors at most 20; rowSpec holds the row pass.
// The maps are zero outside [lo, lo + side)^2 and stored as that square only (lo = 0, side = N: whole maps).
BIOEM_HIDDEN bool bioem_r2c_fft_supported(int N);
BIOEM_HIDDEN hipError_t bioem_r2c_fft_launch(hipStream_t st, int nCU, const double *srcD, const float *srcF,
                                             const double *tempDen, float NormDen, int N, int nImg, const double2 *tw,
                                             double2 *rowSpec, float2 *out, int lo, int side);

enum KernelFamily
{
  KF_GENERIC = 0,
  KF_FAST,   // k_compare_fast<WD, R, NYQ, GS>            windows of at most 21 rows
  KF_FASTM,  // k_compare_fastm<WD, R, NYQ, GS>           27- / 31-row windows, window pass on the matrix cores
  KF_WIDE2,  // k_compare_wide2<R, NRW, NBLK, NYQ, HALVES, NW>  wide windows, row FFT
  KF_ROWS,   // k_compare_rows<WD, GS>                    odd N, direct column sums
  KF_ODDFFT, // k_compare_oddfft<WD, R>                   odd N with a factor 3 / 5 / 9 / 15 / 25
  KF_FASTM2  // k_compare_fastm2<R, NYQ>                  33..47-row windows: rows split over the half-waves, 3 x 3 MFMA tiles
};

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

} // namespace

#endif
