// capi.cpp -- C entry points of the host layer for bindings (tests, bench.py): the one-off precompute
// the reference performs on the host before the hot loop.
#include "bioem_host.h"

extern "C" {

// == bioem_param::CalculateRefCTF kernels part (param.cpp:1336-1583), CTF mode only (no transform needed)
int bioem_host_ctf_kernels(int N, float pixelSize, float startAmp, float endAmp, int nAmp, float startPhase,
                           float endPhase, int nPhase, float startEnv, float endEnv, int nEnv, float *refCTF,
                           float *ctfParam, float *steps)
{
  return bioem_host::ctf_kernels(N, pixelSize, false, startAmp, endAmp, nAmp, startPhase, endPhase, nPhase, startEnv,
                                 endEnv, nEnv, refCTF, ctfParam, steps, nullptr, nullptr);
}

// == volume element (param.cpp:1600-1607)
float bioem_host_volume_element(float voluang, int gridSpaceCenter, int maxDisplaceCenter, float pixelSize, int nAmp,
                                float gridEnvelop, float gridPhase, float sigB, float sigDef, float sigAmp)
{
  return bioem_host::volume_element(voluang, gridSpaceCenter, maxDisplaceCenter, pixelSize, nAmp, gridEnvelop,
                                    gridPhase, sigB, sigDef, sigAmp);
}

// == bioem_model::centerDensityMass (model.cpp:604-672); returns NormDen
float bioem_host_center_model(bioem_hip_model_point *pts, int n)
{
  bioem_host::Model m;
  m.points.assign(pts, pts + n);
  m.NormDen = 0.f;
  for (int i = 0; i < n; i++)
    m.NormDen += pts[i].density;
  m.centerDensityMass();
  for (int i = 0; i < n; i++)
    pts[i] = m.points[i];
  return m.NormDen;
}
}
