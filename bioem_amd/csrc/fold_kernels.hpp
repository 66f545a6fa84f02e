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
                       bioem_hip_prob_angle *__restrict__ pang, int angO0)
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
      bioem_hip_prob_angle pa = pang[(size_t) (iOrient - angO0) * nMaps + p];
      if (pa.ConstAngle < lp)
      {
        pa.forAngles *= exp(-lp + pa.ConstAngle);
        pa.ConstAngle = lp;
      }
      pa.forAngles += r.sumExp * exp(lp - pa.ConstAngle);
      pang[(size_t) (iOrient - angO0) * nMaps + p] = pa;
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
                              bioem_hip_prob_angle *__restrict__ pang, int angO0)
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
  // the table holds the orientations [angO0, ...) this handle owns (all of them unless it is a shard)
  bioem_hip_prob_angle pa = pang[(size_t) (iOrient - angO0) * nMaps + p];
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
  pang[(size_t) (iOrient - angO0) * nMaps + p] = pa;
}

// ------------------------------------------------------------------------------------------------
// wave-parallel fold of the particle entries: one wave per particle; lane l folds a contiguous chunk of
// (orientation, CTF) partials in order, the 64 chunk results are merged by a shuffle reduction that keeps
// the FIRST maximum (lowest index), then combined with the running state exactly like the sequential fold.
// The log-sum-exp merge is associative, so the result equals k_fold's up to double rounding (1e-16).
// ------------------------------------------------------------------------------------------------
// WPP = 4 (few particles: one wave per particle left the chip to ten waves walking 48 partials each, 40 us per launch):
// the four waves of a block share one particle, wave w the w-th quarter of its partials; their results are merged in
// wave order through LDS by the same first-maximum rule.
template <int WPP>
__global__ __launch_bounds__(256) void k_fold_wave(const Partial *__restrict__ partials, int ldPart, int nOC,
                                                   int nMaps, const bioem_hip_param5 *__restrict__ params,
                                                   const float *__restrict__ sumRef, const int *__restrict__ disp,
                                                   int nd, PD pd, int orient0, int conv0, int convPerOrient,
                                                   const int2 *__restrict__ ids,
                                                   bioem_hip_prob_map *__restrict__ pmap)
{
  static_assert(WPP == 1 || WPP == 4, "one wave or one block per particle");
  __shared__ double shM[4], shS[4];
  __shared__ int shI[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = WPP == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (p >= nMaps)
    return;
  const Partial *P = partials + (size_t) p * ldPart;
  const int nChunks = 64 * WPP;
  const int chunk = (nOC + nChunks - 1) / nChunks;
  const int b = min(nOC, (WPP == 1 ? lane : (int) threadIdx.x) * chunk), e = min(nOC, b + chunk);
  double m = -INFINITY, sacc = 0.;
  int idx = 0x7fffffff;
#pragma unroll 4
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
  if (WPP == 4)
  {
    if (lane == 0)
    {
      shM[wave] = m;
      shS[wave] = sacc;
      shI[wave] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int w = 1; w < 4; w++)
      {
        const double m2 = shM[w], s2 = shS[w];
        const int i2 = shI[w];
        if (m2 > m || (m2 == m && i2 < idx))
        {
          sacc = ((m == -INFINITY) ? 0. : sacc * exp(m - m2)) + s2;
          m = m2;
          idx = i2;
        }
        else
          sacc += (m2 == -INFINITY) ? 0. : s2 * exp(m2 - m);
      }
  }
  if (threadIdx.x % (64 * WPP) == 0 && idx != 0x7fffffff)
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

// ------------------------------------------------------------------------------------------------
// shard handles: the angle table never leaves the device
// ------------------------------------------------------------------------------------------------
__global__ void k_init_angles(bioem_hip_prob_angle *__restrict__ pang, size_t n)
{ // bioem.cpp:688-697
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
  {
    bioem_hip_prob_angle a;
    a.forAngles = 0.0;
    a.ConstAngle = MIN_PROB;
    pang[i] = a;
  }
}

// order of the reference's std::pair<double, int> heap items (bioem.cpp:1254-1256)
__host__ __device__ inline bool cand_less(double la, int ia, double lb, int ib)
{
  return la < lb || (la == lb && ia < ib);
}

// K best orientations per particle among the nO owned ones, by the reference writer's own rule (bioem.cpp:1251-1286):
// walk the orientations in order; fill a K-entry set; afterwards an entry replaces the set's minimum (by (logp,
// orientation)) when the minimum's logp is strictly below its own.  One thread per particle: the table is
// orientation-major, so the 64 particles of a wave read consecutive 16-byte entries of one orientation row.  The set
// lives in the thread's K output slots; replacements are rare (~K ln(nO/K) per particle), each followed by a scan
// of the K slots for the new minimum.  Output sorted best first (descending (logp, orientation), the order in which
// the reference prints after emptying its heap).
__global__ void k_topk_angles(const bioem_hip_prob_angle *__restrict__ pang, int nO, int nMaps, int o0, int K,
                              double numconst, bioem_hip_angle_candidate *__restrict__ out)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nMaps)
    return;
  bioem_hip_angle_candidate *mine = out + (size_t) p * K;
  int cnt = 0, minSlot = 0;
  double minLogp = 0.;
  int minIo = 0;
  for (int io = 0; io < nO; io++)
  {
    const bioem_hip_prob_angle pa = pang[(size_t) io * nMaps + p];
    const double logp = log(pa.forAngles) + pa.ConstAngle + numconst;
    bool rescan = false;
    if (cnt < K)
    {
      bioem_hip_angle_candidate c;
      c.forAngles = pa.forAngles;
      c.ConstAngle = pa.ConstAngle;
      c.logp = logp;
      c.orient = o0 + io;
      c.pad = 0;
      mine[cnt++] = c;
      rescan = cnt == K;
    }
    else if (minLogp < logp)
    {
      bioem_hip_angle_candidate c;
      c.forAngles = pa.forAngles;
      c.ConstAngle = pa.ConstAngle;
      c.logp = logp;
      c.orient = o0 + io;
      c.pad = 0;
      mine[minSlot] = c;
      rescan = true;
    }
    if (rescan)
    {
      minSlot = 0;
      minLogp = mine[0].logp;
      minIo = mine[0].orient;
      for (int j = 1; j < K; j++)
      {
        const double l = mine[j].logp;
        const int i = mine[j].orient;
        if (cand_less(l, i, minLogp, minIo))
        {
          minSlot = j;
          minLogp = l;
          minIo = i;
        }
      }
    }
  }
  // best first
  for (int a = 0; a < cnt; a++)
  {
    int best = a;
    for (int b = a + 1; b < cnt; b++)
      if (cand_less(mine[best].logp, mine[best].orient, mine[b].logp, mine[b].orient))
        best = b;
    if (best != a)
    {
      const bioem_hip_angle_candidate t = mine[a];
      mine[a] = mine[best];
      mine[best] = t;
    }
  }
  for (int a = cnt; a < K; a++)
  {
    bioem_hip_angle_candidate c;
    c.forAngles = 0.;
    c.ConstAngle = MIN_PROB;
    c.logp = -INFINITY;
    c.orient = -1;
    c.pad = 0;
    mine[a] = c;
  }
}

// fold of the gathered shards' map entries (bioem.cpp:909-994 restated for one address space): shard s's entries start
// at gathered + s * stride bytes.  Maximum Constoadd wins; ties go to the LOWEST shard (= lowest orientation block,
// the serial first-maximum semantics); Total = sum over shards of Total_s * exp(Constoadd_s - max), in shard order.
__global__ void k_merge_shards(const unsigned char *__restrict__ gathered, int nShards, size_t stride, int nMaps,
                               bioem_hip_prob_map *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nMaps)
    return;
  int who = 0;
  double cmax = reinterpret_cast<const bioem_hip_prob_map *>(gathered)[i].Constoadd;
  for (int s = 1; s < nShards; s++)
  {
    const double c = reinterpret_cast<const bioem_hip_prob_map *>(gathered + (size_t) s * stride)[i].Constoadd;
    if (c > cmax)
    {
      cmax = c;
      who = s;
    }
  }
  double tot = 0.;
  for (int s = 0; s < nShards; s++)
  {
    const bioem_hip_prob_map m = reinterpret_cast<const bioem_hip_prob_map *>(gathered + (size_t) s * stride)[i];
    tot += m.Total * exp(m.Constoadd - cmax);
  }
  bioem_hip_prob_map o = reinterpret_cast<const bioem_hip_prob_map *>(gathered + (size_t) who * stride)[i];
  o.Total = tot;
  o.Constoadd = cmax;
  out[i] = o;
}

} // namespace

#endif
