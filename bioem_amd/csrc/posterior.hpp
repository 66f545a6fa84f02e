// posterior.hpp -- log posterior of one displacement and the log-sum-exp accumulator (reference semantics)
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_POSTERIOR_HPP
#define BIOEM_POSTERIOR_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// log posterior, bioem_algorithm.h:18-70.  constPart = second log term, priorPart = Gaussian priors:
// both depend on the (orientation, CTF) pair only and are hoisted; the summation order
// (t1 + t2) - prior of the reference is kept.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void logpro_consts(const PD &pd, const bioem_hip_param5 &q, double &t2, double &prior)
{
  const float Np = pd.Ntotpi;
  const double ForLogProb = (double) (q.sumsquareC * Np - q.sumC * q.sumC);
  t2 = ((double) Np * 0.5 - 2) * log((double) (Np - 2) * ForLogProb);
  const float amp = q.amp, pha = q.pha, env = q.env;
  if (!pd.tousepsf)
  {
    prior = (double) (env * env) / 2. / (double) pd.sigmaPriorbctf / (double) pd.sigmaPriorbctf -
            (double) ((pha - pd.Priordefcent) * (pha - pd.Priordefcent)) / 2. / (double) pd.sigmaPriordefo /
                (double) pd.sigmaPriordefo -
            (double) ((amp - pd.Priorampcent) * (amp - pd.Priorampcent)) / 2. / (double) pd.sigmaPrioramp /
                (double) pd.sigmaPrioramp;
  }
  else
  {
    const double envF = 4. * M_PI * M_PI * (double) env / (double) (env * env + pha * pha);
    const double phaF = 4. * M_PI * M_PI * (double) pha / (double) (env * env + pha * pha);
    const double dp = phaF - (double) pd.Priordefcent;
    prior = envF * envF / 2. / (double) pd.sigmaPriorbctf / (double) pd.sigmaPriorbctf -
            dp * dp / 2. / (double) pd.sigmaPriordefo / (double) pd.sigmaPriordefo -
            (double) ((amp - pd.Priorampcent) * (amp - pd.Priorampcent)) / 2. / (double) pd.sigmaPrioramp /
                (double) pd.sigmaPrioramp;
  }
}

// the two constants of every (orientation, CTF) row of a launch, for the comparison kernels to read
#ifdef BIOEM_MAIN_TU
__global__ __launch_bounds__(64) void k_posterior_consts(const bioem_hip_param5 *__restrict__ params, const PD pd,
                                                         double2 *__restrict__ out, int n)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n)
  {
    double t2, prior;
    logpro_consts(pd, params[i], t2, prior);
    out[i] = make_double2(t2, prior);
  }
}
#endif

__device__ __forceinline__ double logpro_eval(const PD &pd, const bioem_hip_param5 &q, float cc, float sumref,
                                              float sumsqref, double t2, double prior)
{
  const float Np = pd.Ntotpi;
  const float sum = q.sumC, sumsq = q.sumsquareC;
  const float firstele_f = Np * (sumsqref * sumsq - cc * cc) + 2 * sumref * sum * cc - sumsqref * sum * sum -
                           sumref * sumref * sumsq;
  double logpro = (double) (3 - Np) * 0.5 * log((double) firstele_f) + t2;
  logpro -= prior;
  return logpro;
}

// online log-sum-exp state of one lane / wave
struct Lse
{
  float m;   // best logpro (narrowed to float as the reference does)
  double s;  // sum exp(logpro - m)
  int id;    // rank of the best displacement in the reference's visiting order
  float val; // cross-correlation value at the best displacement
};

__device__ __forceinline__ void lse_init(Lse &L)
{
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
}

// algo 1: logpro narrowed to float before use (bioem_algorithm.h:84); algo 2: double in the exponent,
// float for the running best (bioem.cpp:1470,1500-1507)
__device__ __forceinline__ void lse_push(Lse &L, double lp, int id, float val, int algo)
{
  const float lpf = (float) lp;
  const double lpe = (algo == 1) ? (double) lpf : lp;
  if (L.m < lpf)
  {
    L.s = (L.m == -INFINITY) ? 0. : L.s * exp((double) L.m - (double) lpf);
    L.m = lpf;
    L.id = id;
    L.val = val;
  }
  L.s += exp(lpe - (double) L.m);
}

__device__ __forceinline__ void lse_merge(Lse &L, float m2, double s2, int id2, float val2)
{
  if (m2 > L.m || (m2 == L.m && id2 < L.id))
  {
    const double sc = (L.m == -INFINITY) ? 0. : L.s * exp((double) L.m - (double) m2);
    L.s = sc + s2;
    L.m = m2;
    L.id = id2;
    L.val = val2;
  }
  else
  {
    const double sc = (m2 == -INFINITY) ? 0. : s2 * exp((double) m2 - (double) L.m);
    L.s += sc;
  }
}

__device__ __forceinline__ void lse_wave_reduce(Lse &L)
{
  for (int off = 32; off > 0; off >>= 1)
  {
    const float m2 = __shfl_xor(L.m, off);
    const double s2 = __shfl_xor(L.s, off);
    const int id2 = __shfl_xor(L.id, off);
    const float v2 = __shfl_xor(L.val, off);
    lse_merge(L, m2, s2, id2, v2);
  }
}

} // namespace

#endif
