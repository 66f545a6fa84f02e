// fold_kernels.hpp -- fold of per-comparison partials into the probability block
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_FOLD_KERNELS_HPP
#define BIOEM_FOLD_KERNELS_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// fold: one thread per particle walks its partials in (orientation, CTF) order.
// bioem_algorithm.h:94-141 / bioem.cpp:1527-1600.
// ------------------------------------------------------------------------------------------------
__global__ void k_fold(const Partial *__restrict__ partials, int ldPart, int nOC, int nMaps,
                       const bioem_hip_param5 *__restrict__ params, const float *__restrict__ sumRef,
                       const int *__restrict__ disp, int nd, PD pd, int orient0, int conv0, int convPerOrient,
                       const int2 *__restrict__ ids, bioem_hip_prob_map *__restrict__ pmap,
                       bioem_hip_prob_angle *__restrict__ pang)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nMaps)
    return;
  bioem_hip_prob_map pm = pmap[p];
  const float sumref = sumRef[p];
  const Partial *P = partials + (size_t) p * ldPart;
  for (int oc = 0; oc < nOC; oc++)
  {
    const Partial r = P[oc];
    const int iOrient = ids ? ids[oc].x : orient0 + oc / convPerOrient;
    const int iConv = ids ? ids[oc].y : conv0 + oc % convPerOrient;
    const double lp = (double) r.best;
    if (pm.Constoadd < lp)
    {
      pm.Total *= exp(-lp + pm.Constoadd);
      pm.Constoadd = lp;
      const int ix = r.id / nd, iy = r.id - ix * nd;
      pm.max_prob_cent_x = -disp[ix];
      pm.max_prob_cent_y = -disp[iy];
      pm.max_prob_orient = iOrient;
      pm.max_prob_conv = iConv;
      const bioem_hip_param5 q = params[oc];
      const float value = r.value;
      pm.max_prob_norm = -(-q.sumC * sumref + pd.Ntotpi * value) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
      pm.max_prob_mu = -(-q.sumC * value + q.sumsquareC * sumref) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
    }
    pm.Total += r.sumExp * exp(lp - pm.Constoadd);
    if (pd.writeAngles)
    {
      bioem_hip_prob_angle pa = pang[(size_t) iOrient * nMaps + p];
      if (pa.ConstAngle < lp)
      {
        pa.forAngles *= exp(-lp + pa.ConstAngle);
        pa.ConstAngle = lp;
      }
      pa.forAngles += r.sumExp * exp(lp - pa.ConstAngle);
      pang[(size_t) iOrient * nMaps + p] = pa;
    }
  }
  pmap[p] = pm;
}

// ------------------------------------------------------------------------------------------------
// WRITE_PROB_ANGLES table: one thread per (particle, orientation of this launch) folds that orientation's CTF
// partials into its angle entry, in CTF order -- the same arithmetic sequence per entry as k_fold
// (bioem_algorithm.h:130-141), but nMaps * nOrient threads instead of nMaps.  The particle entries are then
// folded by k_fold_wave.  Rows of orientation j: [j*convPerOrient, (j+1)*convPerOrient) (native path, segs == null)
// or the run segs[j] = {first row, end row, orientation} (compat ring: every orientation of a launch is ONE run).
// ------------------------------------------------------------------------------------------------
__global__ void k_fold_angles(const Partial *__restrict__ partials, int ldPart, int nOC, int nMaps, int orient0,
                              int convPerOrient, const int4 *__restrict__ segs, int nRuns,
                              bioem_hip_prob_angle *__restrict__ pang)
{
  const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long) nRuns * nMaps)
    return;
  const int j = (int) (t / nMaps), p = (int) (t - (long long) j * nMaps); // particle index fastest: coalesced table
  const Partial *P = partials + (size_t) p * ldPart;
  int ocBegin = j * convPerOrient, ocEnd = min(nOC, (j + 1) * convPerOrient), iOrient = orient0 + j;
  if (segs)
  {
    const int4 sg = segs[j];
    ocBegin = sg.x;
    ocEnd = sg.y;
    iOrient = sg.z;
  }
  bioem_hip_prob_angle pa = pang[(size_t) iOrient * nMaps + p];
  for (int oc = ocBegin; oc < ocEnd; oc++)
  {
    const Partial r = P[oc];
    const double lp = (double) r.best;
    if (pa.ConstAngle < lp)
    {
      pa.forAngles *= exp(-lp + pa.ConstAngle);
      pa.ConstAngle = lp;
    }
    pa.forAngles += r.sumExp * exp(lp - pa.ConstAngle);
  }
  pang[(size_t) iOrient * nMaps + p] = pa;
}

// ------------------------------------------------------------------------------------------------
// wave-parallel fold of the particle entries: one wave per particle; lane l folds a contiguous chunk of
// (orientation, CTF) partials in order, the 64 chunk results are merged by a shuffle reduction that keeps
// the FIRST maximum (lowest index), then combined with the running state exactly like the sequential fold.
// The log-sum-exp merge is associative, so the result equals k_fold's up to double rounding (1e-16).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_wave(const Partial *__restrict__ partials, int ldPart, int nOC,
                                                   int nMaps, const bioem_hip_param5 *__restrict__ params,
                                                   const float *__restrict__ sumRef, const int *__restrict__ disp,
                                                   int nd, PD pd, int orient0, int conv0, int convPerOrient,
                                                   const int2 *__restrict__ ids,
                                                   bioem_hip_prob_map *__restrict__ pmap)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wave;
  if (p >= nMaps)
    return;
  const Partial *P = partials + (size_t) p * ldPart;
  const int chunk = (nOC + 63) / 64;
  const int b = lane * chunk, e = min(nOC, b + chunk);
  double m = -INFINITY, sacc = 0.;
  int idx = 0x7fffffff;
  for (int oc = b; oc < e; oc++)
  {
    const Partial r = P[oc];
    const double lp = (double) r.best;
    if (m < lp)
    {
      sacc = (m == -INFINITY) ? 0. : sacc * exp(m - lp);
      m = lp;
      idx = oc;
    }
    sacc += r.sumExp * exp(lp - m);
  }
  for (int off = 32; off > 0; off >>= 1)
  {
    const double m2 = __shfl_xor(m, off);
    const double s2 = __shfl_xor(sacc, off);
    const int i2 = __shfl_xor(idx, off);
    if (m2 > m || (m2 == m && i2 < idx))
    {
      sacc = ((m == -INFINITY) ? 0. : sacc * exp(m - m2)) + s2;
      m = m2;
      idx = i2;
    }
    else
      sacc += (m2 == -INFINITY) ? 0. : s2 * exp(m2 - m);
  }
  if (lane == 0 && idx != 0x7fffffff)
  {
    bioem_hip_prob_map pm = pmap[p];
    if (pm.Constoadd < m)
    {
      pm.Total *= exp(-m + pm.Constoadd);
      pm.Constoadd = m;
      const Partial r = P[idx];
      const int ix = r.id / nd, iy = r.id - ix * nd;
      pm.max_prob_cent_x = -disp[ix];
      pm.max_prob_cent_y = -disp[iy];
      pm.max_prob_orient = ids ? ids[idx].x : orient0 + idx / convPerOrient;
      pm.max_prob_conv = ids ? ids[idx].y : conv0 + idx % convPerOrient;
      const bioem_hip_param5 q = params[idx];
      const float sumref = sumRef[p];
      const float value = r.value;
      pm.max_prob_norm = -(-q.sumC * sumref + pd.Ntotpi * value) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
      pm.max_prob_mu = -(-q.sumC * value + q.sumsquareC * sumref) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
    }
    pm.Total += sacc * exp(m - pm.Constoadd);
    pmap[p] = pm;
  }
}

} // namespace

#endif
