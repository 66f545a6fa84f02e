/* bioem_oracle.h -- CPU ORACLE for the BioEM compare path.  TEST INFRASTRUCTURE ONLY.
 *
 * A clean-room restatement in plain C of the reference CPU hot path (bio-phys/BioEM v2.1,
 * /root/reference): projection -> CTF/PSF convolution -> FFT cross-correlation -> log-posterior
 * -> log-sum-exp accumulation -> shard merge -> final log P.  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (bioem_amd/) never links, imports or executes it.
 *
 * Pinning: validated against outputs of the reference itself (oracle/_ref/bioEM_ref, the unmodified
 * reference sources linked against the image's hipFFTW) committed under tests/golden/ -- see
 * tests/test_oracle_golden.py.
 *
 * Third-party arithmetic outside the reference tree: FFTW 3 (>= 3.3.3, unpinned system library)
 * supplies r2c/c2r.  Here the transforms are the exact DFT definitions (forward sign -1,
 * unnormalised, half-spectrum [N][N/2+1]) evaluated in double and rounded to float where the
 * reference stores floats; c2r follows FFTW's rdft2 convention (complex inverse along dim 0,
 * then half-complex->real along dim 1 ignoring Im of column 0 and column N/2).
 */
#ifndef BIOEM_ORACLE_H
#define BIOEM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference: include/param.h:26-47 (bioem_param_device), same field order; bool -> int */
typedef struct
{
  int maxDisplaceCenter;
  int GridSpaceCenter;
  int NumberPixels;
  int NumberFFTPixels1D;
  int NxDisp;
  int NtotDisp;
  float Ntotpi;
  float volu;
  float sigmaPriorbctf;
  float sigmaPriordefo;
  float Priordefcent;
  float sigmaPrioramp;
  float Priorampcent;
  int writeAngles;
  int tousepsf;
} orc_param_device;

/* reference: include/map.h:116-128 (bioem_Probability_map), 40 bytes */
typedef struct
{
  double Total;
  double Constoadd;
  int max_prob_cent_x, max_prob_cent_y, max_prob_orient, max_prob_conv;
  float max_prob_norm, max_prob_mu;
} orc_prob_map;

/* reference: include/map.h:130-135 */
typedef struct
{
  double forAngles;
  double ConstAngle;
} orc_prob_angle;

/* reference: include/defs.h:128-135 (myparam5_t) */
typedef struct
{
  float amp, pha, env, sumC, sumsquareC;
} orc_param5;

/* reference: include/model.h (bioem_model_point): myfloat3_t point (16 B) + radius + density = 24 B */
typedef struct
{
  float pos[3];
  float quat4_unused;
  float radius;
  float density;
} orc_model_point;

/* CTF/PSF grid description (reference: param.h:104-126 after readParameters unit conversion) */
typedef struct
{
  float startAmp, endAmp;
  int nAmp;
  float startPhase, endPhase;
  int nPhase;
  float startEnv, endEnv;
  int nEnv;
} orc_ctf_grid;

/* ---- transforms (FFTW conventions) ---- */
void orc_fft2_r2c(int N, const float *in, float *out /* [N][N/2+1][2] */);
void orc_fft2_c2r(int N, const float *in /* [N][N/2+1][2] */, float *out /* [N][N] */);

/* ---- one-off precompute ---- */
/* param.cpp:1336-1620: refCTF [nCTF][N*(N/2+1)][2], ctfParam [nCTF][3]; returns nCTF; steps[3] = gridAmp, gridPhase, gridEnv */
int orc_ctf_kernels(int N, float pixelSize, int usepsf, const orc_ctf_grid *g, float *refCTF, float *ctfParam, float *steps);
/* param.cpp:1600-1607 */
float orc_volu(float voluang, int gridSpaceCenter, int maxDisplaceCenter, float pixelSize, int nAmp, float gridEnvelop,
               float gridPhase, float sigmaPriorbctf, float sigmaPriordefo, float sigmaPrioramp);
/* model.cpp:604-672 (sequential branch) */
void orc_center_model(orc_model_point *pts, int n, float NormDen);
/* bioem.cpp:2087-2107 */
void orc_map_sums(int N, const float *map, float *sum, float *sumsquare);

/* ---- hot path ---- */
/* bioem.cpp:1604-1853; angle = {pos0,pos1,pos2,quat4}; returns number of points dropped */
int orc_projection(const orc_model_point *pts, int nPts, float NormDen, const float *angle, int isQuat, int N,
                   float pixelSize, int shiftX, int shiftY, float *realmap_or_null, float *spec /* [N][H][2] */);
/* bioem.cpp:1855-1923 */
void orc_convolve(int N, const float *proj, const float *refCTF, float *out, float *sumC, float *sumsquareC);
/* bioem_algorithm.h:18-70 */
double orc_calc_logpro(const orc_param_device *pd, float amp, float pha, float env, float sum, float sumsquare,
                       float crossproMapConv, float sumref, float sumsquareref);
/* bioem.cpp:1435-1459: cross-correlation map lCC[N][N] (unnormalised) */
void orc_cc_map(int N, const float *convFFT, const float *refFFT, float *lCC);
/* bioem.cpp:1379-1433 + bioem_algorithm.h:72-198 (ALGO 1) / bioem.cpp:1461-1602 (ALGO 2) */
void orc_compare(const orc_param_device *pd, int algo, int nMaps, int nAnglesTotal, const float *refFFT,
                 const float *sumRef, const float *sumsqRef, int iOrient, int iConvStart, int nConv,
                 const float *convFFT, const orc_param5 *params, orc_prob_map *pmap, orc_prob_angle *pang);
/* bioem.cpp:681-699 */
void orc_init_prob(int nMaps, int nAngles, int writeAngles, orc_prob_map *pmap, orc_prob_angle *pang);
/* bioem.cpp:763-891 main loop over orientations [o0,o1) and all CTFs */
void orc_run(const orc_param_device *pd, int algo, const orc_model_point *pts, int nPts, float NormDen,
             const float *angles /* [nAngles][4] */, int nAnglesTotal, int isQuat, float pixelSize, int shiftX,
             int shiftY, int nCTF, const float *refCTF, const float *ctfParam, int nMaps, const float *refFFT,
             const float *sumRef, const float *sumsqRef, int o0, int o1, orc_prob_map *pmap, orc_prob_angle *pang);
/* bioem.cpp:909-994 (log-sum-exp merge of orientation shards; tie -> lowest shard = lowest orientation) */
void orc_merge(int nShards, int nMaps, const orc_prob_map *shards /* [nShards][nMaps] */, orc_prob_map *out);
/* bioem.cpp:1144-1150 */
double orc_final_logp(const orc_param_device *pd, double Total, double Constoadd);

int orc_sizeof_prob_map(void);
void orc_set_num_threads(int n);
int orc_get_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
