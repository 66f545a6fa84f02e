// compare_args.hpp -- argument block shared by the comparison kernels
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_ARGS_HPP
#define BIOEM_COMPARE_ARGS_HPP

namespace
{

struct CompareArgs
{
  const float2 *ref;  // [nMaps][M] comparison layout
  const float2 *conv; // [nOC][M]
  const bioem_hip_param5 *params;
  const double2 *postc; // [nOC] {t2, prior} of logpro_consts, evaluated once per row by k_posterior_consts
  const float *sumRef, *sumsqRef;
  const float2 *tw; // N+1
  const int *disp;  // nd
  const double2 *ltab; // 64 x {c, -log c}
  const float2 *twk;   // [N1][2*WD+1] recombination twiddles exp(2 pi i d k1 / N), d = -WD..WD
  const float2 *twnyq; // [N/2][2*WD+1][2] twiddles of k_nyquist_rows (wave-uniform reads: wide scalar loads)
  const float *btab;   // k_compare_fastm2: B operand of the matrix pass, [column pass][16][3][64] (fastm2_btab_floats)
  float *tnyq;         // [nMaps][ldPart][2*WD+1] Nyquist-column rows (fast path with the Nyquist split only)
  Partial *partials; // [nMaps][ldPart]
  int ldPart;
  int N, H, N1, nd, maxD, nOC, nMaps, algo;
  int Hp; // row-pair pitch of ref / conv in 16-byte words: H, or H + 15 (fast families with the Nyquist split; bioem_hip.hip)
  int pchunk; // particles per block-order chunk of the fast kernel
  int gs;     // pixels per window row of the fast kernel (template GS)
  // window tiles (wide windows are covered by several launches over phase-shifted conv spectra): only the first
  // ndx rows (sorted order) and the first ndy lanes of the displacement list count; nd for an untiled launch
  int ndx, ndy;
  // k_compare_wide2: row stride of the LDS T block (float2 units); window half width of the k_nyquist_rows
  // instantiation that filled tnyq (its rows run -nyqWD..nyqWD)
  int ts, nyqWD;
  // k_compare_fast: the last column block holds at most 32 columns and its half-waves share them (compare_fast.hpp)
  int split;
  PD pd;
};

} // namespace

#endif
