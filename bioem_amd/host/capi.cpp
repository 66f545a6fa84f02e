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

// ------------------------------------------------------------------------------------------------
// file-level set-up without a device (CTF mode): readParameters + CalculateGridsParam + CalculateRefCTF.
// Used by the CPU tests to compare the host layer with the oracle's restatement on the golden inputs.
// Two-call protocol: sizes first (arrays NULL), then the arrays.
// ------------------------------------------------------------------------------------------------
struct bioem_host_setup
{
  bioem_hip_param_device pd;
  int nAngles, nCTF, isQuat, usepsf, shiftX, shiftY, nocentermass;
  float pixelSize, voluang, elecwavel;
};

int bioem_host_setup_from_files(const char *paramfile, const char *anglefile, bioem_host_setup *out, float *angles4,
                                float *refCTF, float *ctfParam3)
{
  bioem_host::InputParams P;
  P.notuniformangles = (anglefile && anglefile[0]);
  P.readParameters(paramfile);
  P.calculateGridsParam(anglefile ? anglefile : "");
  out->nAngles = P.nTotGridAngles;
  out->isQuat = P.doquater;
  out->usepsf = P.usepsf;
  out->shiftX = P.shiftX;
  out->shiftY = P.shiftY;
  out->nocentermass = P.nocentermass;
  out->pixelSize = P.pixelSize;
  out->voluang = P.voluang;
  out->elecwavel = P.elecwavel;
  out->nCTF = P.numberGridPointsCTF_amp * P.numberGridPointsCTF_phase * P.numberGridPointsEnvelop;
  if (!P.usepsf)
  {
    P.calculateRefCTF();
    if (refCTF)
      for (size_t e = 0; e < P.refCTF.size(); e++)
        refCTF[e] = P.refCTF[e];
    if (ctfParam3)
      for (size_t e = 0; e < P.ctfParam.size(); e++)
        ctfParam3[e] = P.ctfParam[e];
  }
  out->pd = P.pd;
  if (angles4)
    for (size_t e = 0; e < P.angles.size(); e++)
      angles4[e] = P.angles[e];
  return 0;
}

// model readers: returns the number of points (fills up to cap), NormDen through *normden
// isPDB: 0 text, 1 PDB, 2 MRC density map (pixelSize needed for the voxel positions)
int bioem_host_read_model(const char *file, int isPDB, int nocentermass, float pixelSize, bioem_hip_model_point *pts,
                          int cap, float *normden)
{
  bioem_host::InputParams P;
  P.nocentermass = nocentermass;
  P.ignorePDB = true;
  P.pixelSize = pixelSize;
  bioem_host::Model m;
  m.readPDB = (isPDB == 1);
  m.readModelMRC = (isPDB == 2);
  m.readModel(P, file);
  for (int i = 0; i < (int) m.points.size() && i < cap; i++)
    pts[i] = m.points[i];
  *normden = m.NormDen;
  return (int) m.points.size();
}

// particle readers: mode 0 text, 1 single MRC, 2 list of MRCs; returns the number of images
int bioem_host_read_particles(const char *file, int mode, int N, int notnormmap, float *maps, int cap)
{
  bioem_host::InputParams P;
  P.N = N;
  P.notnormmap = notnormmap;
  bioem_host::ParticleStack S;
  S.readMRC = mode >= 1;
  S.readMultMRC = mode == 2;
  S.readRefMaps(P, file);
  const size_t sz = (size_t) N * N;
  for (int i = 0; i < S.ntot && i < cap; i++)
    for (size_t e = 0; e < sz; e++)
      maps[(size_t) i * sz + e] = S.maps[(size_t) i * sz + e];
  return S.ntot;
}
}
