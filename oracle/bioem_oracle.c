/* bioem_oracle.c -- CPU ORACLE for the BioEM compare path.  TEST INFRASTRUCTURE ONLY.
 *
 * See bioem_oracle.h for scope, pinning status and the FFT convention.  Plain C, float/double
 * evaluation order follows the reference expressions (compiled with -ffp-contract=off, no
 * fast-math; x86-64 SSE => FLT_EVAL_METHOD 0, float expressions are evaluated in float).
 * This file is the checker and the timed CPU baseline ("port"); it is never shipped in or called
 * by the product path.
 */
#include "bioem_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
#endif

#define MIN_PROB (-999999.) /* defs.h:65 */

/* ============================================================================================
 * Mixed-radix Stockham FFT, double precision (exact-DFT stand-in for the FFTW calls of the
 * reference: bioem.cpp:1458,1848, map.cpp:585, param.cpp:1521).
 * ============================================================================================ */
typedef struct
{
  double re, im;
} cpx;

typedef struct
{
  int n;
  int nf;
  int radix[40];
  cpx *tw; /* tw[k] = exp(+2*pi*i*k/n) */
} fft_plan;

#define MAX_PLANS 256
static fft_plan g_plans[MAX_PLANS];
static int g_nplans = 0;

static const fft_plan *get_plan(int n)
{
  const fft_plan *res = NULL;
#pragma omp critical(orc_plan)
  {
    for (int i = 0; i < g_nplans; i++)
      if (g_plans[i].n == n)
        res = &g_plans[i];
    if (!res)
    {
      if (g_nplans >= MAX_PLANS)
      {
        fprintf(stderr, "oracle: too many FFT plans\n");
        abort();
      }
      fft_plan *P = &g_plans[g_nplans];
      P->n = n;
      P->nf = 0;
      int m = n;
      while (m % 4 == 0)
      {
        P->radix[P->nf++] = 4;
        m /= 4;
      }
      for (int r = 2; r <= m; r++)
        while (m % r == 0)
        {
          P->radix[P->nf++] = r;
          m /= r;
        }
      P->tw = (cpx *) malloc(sizeof(cpx) * (size_t) n);
      for (int k = 0; k < n; k++)
      {
        double a = 2.0 * M_PI * (double) k / (double) n;
        P->tw[k].re = cos(a);
        P->tw[k].im = sin(a);
      }
      g_nplans++;
      res = P;
    }
  }
  return res;
}

/* out-of-place ping-pong; result ends in x.  sign=+1: sum x[k] e^{+2 pi i jk/n}; sign=-1: conjugate kernel */
static void fft_exec(const fft_plan *P, int sign, cpx *x, cpx *y)
{
  const int n = P->n;
  int s = 1, nn = n;
  cpx *a = x, *b = y;
  for (int f = 0; f < P->nf; f++)
  {
    const int r = P->radix[f];
    const int m = nn / r;
    if (r == 2)
    {
      for (int p = 0; p < m; p++)
      {
        cpx w = P->tw[(p * s) % n];
        if (sign < 0)
          w.im = -w.im;
        for (int q = 0; q < s; q++)
        {
          const cpx u = a[q + s * p], v = a[q + s * (p + m)];
          cpx d = {u.re - v.re, u.im - v.im};
          b[q + s * (2 * p)].re = u.re + v.re;
          b[q + s * (2 * p)].im = u.im + v.im;
          b[q + s * (2 * p + 1)].re = d.re * w.re - d.im * w.im;
          b[q + s * (2 * p + 1)].im = d.re * w.im + d.im * w.re;
        }
      }
    }
    else if (r == 4)
    {
      for (int p = 0; p < m; p++)
      {
        cpx w1 = P->tw[(p * s) % n], w2 = P->tw[(2 * p * s) % n], w3 = P->tw[(3 * p * s) % n];
        if (sign < 0)
        {
          w1.im = -w1.im;
          w2.im = -w2.im;
          w3.im = -w3.im;
        }
        for (int q = 0; q < s; q++)
        {
          const cpx a0 = a[q + s * p], a1 = a[q + s * (p + m)], a2 = a[q + s * (p + 2 * m)],
                    a3 = a[q + s * (p + 3 * m)];
          const cpx t0 = {a0.re + a2.re, a0.im + a2.im}, t1 = {a0.re - a2.re, a0.im - a2.im};
          const cpx t2 = {a1.re + a3.re, a1.im + a3.im};
          /* i*sign*(a1-a3) */
          cpx t3;
          if (sign > 0)
          {
            t3.re = -(a1.im - a3.im);
            t3.im = (a1.re - a3.re);
          }
          else
          {
            t3.re = (a1.im - a3.im);
            t3.im = -(a1.re - a3.re);
          }
          cpx *o = &b[q + s * (4 * p)];
          o[0].re = t0.re + t2.re;
          o[0].im = t0.im + t2.im;
          const cpx b1 = {t1.re + t3.re, t1.im + t3.im};
          const cpx b2 = {t0.re - t2.re, t0.im - t2.im};
          const cpx b3 = {t1.re - t3.re, t1.im - t3.im};
          o[s].re = b1.re * w1.re - b1.im * w1.im;
          o[s].im = b1.re * w1.im + b1.im * w1.re;
          o[2 * s].re = b2.re * w2.re - b2.im * w2.im;
          o[2 * s].im = b2.re * w2.im + b2.im * w2.re;
          o[3 * s].re = b3.re * w3.re - b3.im * w3.im;
          o[3 * s].im = b3.re * w3.im + b3.im * w3.re;
        }
      }
    }
    else
    {
      /* generic radix-r butterfly, O(r^2) */
      cpx av[1024], wr[1024];
      if (r > 1024)
      {
        fprintf(stderr, "oracle: prime factor %d too large\n", r);
        abort();
      }
      for (int k = 0; k < r; k++)
      {
        wr[k] = P->tw[(size_t) k * (size_t) (n / r)];
        if (sign < 0)
          wr[k].im = -wr[k].im;
      }
      for (int p = 0; p < m; p++)
      {
        for (int q = 0; q < s; q++)
        {
          for (int k = 0; k < r; k++)
            av[k] = a[q + s * (p + m * k)];
          for (int j = 0; j < r; j++)
          {
            double sr = 0., si = 0.;
            for (int k = 0; k < r; k++)
            {
              const cpx w = wr[(j * k) % r];
              sr += av[k].re * w.re - av[k].im * w.im;
              si += av[k].re * w.im + av[k].im * w.re;
            }
            cpx w = P->tw[(int) (((long long) p * j * s) % n)];
            if (sign < 0)
              w.im = -w.im;
            b[q + s * (r * p + j)].re = sr * w.re - si * w.im;
            b[q + s * (r * p + j)].im = sr * w.im + si * w.re;
          }
        }
      }
    }
    cpx *t = a;
    a = b;
    b = t;
    nn = m;
    s *= r;
  }
  if (a != x)
    memcpy(x, a, sizeof(cpx) * (size_t) n);
}

/* r2c, FFTW layout out[u][k], k < N/2+1, forward sign -1, unnormalised */
void orc_fft2_r2c(int N, const float *in, float *out)
{
  const int H = N / 2 + 1;
  const fft_plan *P = get_plan(N);
  cpx *rows = (cpx *) malloc(sizeof(cpx) * (size_t) N * (size_t) H);
  cpx *x = (cpx *) malloc(sizeof(cpx) * (size_t) N * 2);
  cpx *y = x + N;
  for (int i = 0; i < N; i++)
  {
    for (int j = 0; j < N; j++)
    {
      x[j].re = (double) in[i * N + j];
      x[j].im = 0.;
    }
    fft_exec(P, -1, x, y);
    for (int k = 0; k < H; k++)
      rows[(size_t) i * H + k] = x[k];
  }
  for (int k = 0; k < H; k++)
  {
    for (int i = 0; i < N; i++)
      x[i] = rows[(size_t) i * H + k];
    fft_exec(P, -1, x, y);
    for (int i = 0; i < N; i++)
    {
      out[2 * ((size_t) i * H + k)] = (float) x[i].re;
      out[2 * ((size_t) i * H + k) + 1] = (float) x[i].im;
    }
  }
  free(x);
  free(rows);
}

/* c2r with scratch supplied by the caller: work = cpx[N*H + 2N] */
static void c2r_work(int N, const float *in, float *out, cpx *work)
{
  const int H = N / 2 + 1;
  const fft_plan *P = get_plan(N);
  cpx *t = work;               /* [N][H] after the column pass */
  cpx *x = work + (size_t) N * H;
  cpx *y = x + N;
  /* complex inverse along dim 0 for each stored column */
  for (int k = 0; k < H; k++)
  {
    for (int i = 0; i < N; i++)
    {
      x[i].re = (double) in[2 * ((size_t) i * H + k)];
      x[i].im = (double) in[2 * ((size_t) i * H + k) + 1];
    }
    fft_exec(P, +1, x, y);
    for (int i = 0; i < N; i++)
      t[(size_t) i * H + k] = x[i];
  }
  /* half-complex -> real along dim 1, two rows per complex transform; Im of k=0 and k=N/2 ignored */
  const int even = (N % 2 == 0);
  for (int i = 0; i < N; i += 2)
  {
    const cpx *ta = &t[(size_t) i * H];
    const int have_b = (i + 1 < N);
    const cpx *tb = have_b ? &t[(size_t) (i + 1) * H] : NULL;
    /* z = ext(a) + i * ext(b) */
    x[0].re = ta[0].re;
    x[0].im = have_b ? tb[0].re : 0.;
    const int kend = even ? H - 1 : H;
    for (int k = 1; k < kend; k++)
    {
      const double ar = ta[k].re, ai = ta[k].im;
      const double br = have_b ? tb[k].re : 0., bi = have_b ? tb[k].im : 0.;
      /* ext(a)[k] = a, ext(a)[N-k] = conj(a) */
      x[k].re = ar - bi;
      x[k].im = ai + br;
      x[N - k].re = ar + bi;
      x[N - k].im = -ai + br;
    }
    if (even)
    {
      x[N / 2].re = ta[N / 2].re;
      x[N / 2].im = have_b ? tb[N / 2].re : 0.;
    }
    fft_exec(P, +1, x, y);
    for (int j = 0; j < N; j++)
    {
      out[(size_t) i * N + j] = (float) x[j].re;
      if (have_b)
        out[(size_t) (i + 1) * N + j] = (float) x[j].im;
    }
  }
}

void orc_fft2_c2r(int N, const float *in, float *out)
{
  const int H = N / 2 + 1;
  cpx *work = (cpx *) malloc(sizeof(cpx) * ((size_t) N * H + 2 * (size_t) N));
  c2r_work(N, in, out, work);
  free(work);
}

/* ============================================================================================
 * One-off precompute
 * ============================================================================================ */

/* reference: param.cpp:1336-1583.  Index quirks reproduced: rows i and N-1-i both receive the
 * value of frequency index i (param.cpp:1560-1568), loop bounds i,j < N/2+1. */
int orc_ctf_kernels(int N, float pixelSize, int usepsf, const orc_ctf_grid *g, float *refCTF, float *ctfParam,
                    float *steps)
{
  const int H = N / 2 + 1;
  const size_t M = (size_t) N * H;
  const int nctfmax = N / 2;
  float gridAmp = (g->endAmp - g->startAmp) / (float) g->nAmp;       /* param.cpp:1365 */
  float gridPhase = (g->endPhase - g->startPhase) / (float) g->nPhase; /* :1367 */
  float gridEnv = (g->endEnv - g->startEnv) / (float) g->nEnv;        /* :1369 */
  if (g->nAmp == 1)
    gridAmp = g->startAmp; /* :1373-1376 */
  if (g->nPhase == 1)
    gridPhase = g->startPhase;
  if (g->nEnv == 1)
    gridEnv = g->startEnv;
  steps[0] = gridAmp;
  steps[1] = gridPhase;
  steps[2] = gridEnv;
  float *localCTF = usepsf ? (float *) malloc(sizeof(float) * (size_t) N * N) : NULL;
  int n = 0;
  for (int iamp = 0; iamp < g->nAmp; iamp++)
  {
    const float amp = (float) iamp * gridAmp + g->startAmp; /* :1426 */
    for (int iphase = 0; iphase < g->nPhase; iphase++)
    {
      const float phase = (float) iphase * gridPhase + g->startPhase; /* :1431 */
      for (int ienv = 0; ienv < g->nEnv; ienv++)
      {
        const float env = (float) ienv * gridEnv + g->startEnv; /* :1436 */
        float *cur = refCTF + 2 * M * (size_t) n;
        memset(cur, 0, sizeof(float) * 2 * M);
        if (usepsf)
        {
          /* :1466-1535 real-space PSF on the wrapped radius, normalised by its (float) sum, then r2c */
          float normctf = 0.0f;
          for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++)
            {
              const int ri = (i < nctfmax + 1) ? i : N - i;
              const int rj = (j < nctfmax + 1) ? j : N - j;
              const float radsq = (float) (ri * ri + rj * rj) * pixelSize * pixelSize;
              const float ctf =
                  (float) (exp(-radsq * env / 2.0) *
                           (-amp * cos(radsq * phase / 2.0) - sqrtf((1 - amp * amp)) * sin(radsq * phase / 2.0)));
              localCTF[i * N + j] = ctf;
              normctf += localCTF[i * N + j];
            }
          for (int i = 0; i < N * N; i++)
            localCTF[i] = localCTF[i] / normctf;
          orc_fft2_r2c(N, localCTF, cur);
        }
        else
        {
          /* :1539-1570 CTF directly in Fourier space */
          float normctf = 0.0f;
          for (int i = 0; i < H; i++)
            for (int j = 0; j < H; j++)
            {
              const float radsq = (float) (i * i + j * j) / N / N / pixelSize / pixelSize;
              const float ctf =
                  (float) (exp(-env * radsq / 2.) *
                           (-amp * cos(phase * radsq / 2.) - sqrtf((1 - amp * amp)) * sin(phase * radsq / 2.)));
              if (i == 0 && j == 0)
                normctf = ctf;
              cur[2 * ((size_t) i * H + j)] = ctf / normctf;
              cur[2 * ((size_t) i * H + j) + 1] = 0.f;
              cur[2 * ((size_t) (N - i - 1) * H + j)] = ctf / normctf;
              cur[2 * ((size_t) (N - i - 1) * H + j) + 1] = 0.f;
            }
        }
        ctfParam[3 * n + 0] = amp;
        ctfParam[3 * n + 1] = phase;
        ctfParam[3 * n + 2] = env;
        n++;
      }
    }
  }
  free(localCTF);
  return n;
}

/* reference: param.cpp:1600-1607 -- float until the first double literal, the second divisor is
 * 2*(maxD+1) as written. */
float orc_volu(float voluang, int g, int maxD, float pixelSize, int nAmp, float gridEnvelop, float gridPhase,
               float sigB, float sigDef, float sigAmp)
{
  const float t = voluang * (float) g * pixelSize * (float) g * pixelSize;
  double v = (double) t / ((2.f * (float) maxD + 1.));
  v = v / (double) (2.f * (float) (maxD + 1.));
  v = v / (double) (float) nAmp;
  v = v * (double) gridEnvelop;
  v = v * (double) gridPhase;
  v = v / (double) 4.f;
  v = v / M_PI;
  v = v / sqrt(2.f * M_PI);
  v = v / (double) sigB;
  v = v / (double) sigDef;
  v = v / (double) sigAmp;
  return (float) v;
}

/* reference: model.cpp:604-672, sequential branch (float accumulation) */
void orc_center_model(orc_model_point *pts, int n, float NormDen)
{
  float r[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++)
      r[k] += pts[i].pos[k] * pts[i].density;
  for (int k = 0; k < 3; k++)
    r[k] /= NormDen;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++)
      pts[i].pos[k] -= r[k];
}

/* reference: bioem.cpp:2087-2107 */
void orc_map_sums(int N, const float *map, float *sum, float *sumsquare)
{
  float s = 0.0f, s2 = 0.0f;
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++)
    {
      s += map[i * N + j];
      s2 += map[i * N + j] * map[i * N + j];
    }
  *sum = s;
  *sumsquare = s2;
}

/* ============================================================================================
 * Hot path
 * ============================================================================================ */

/* reference: bioem.cpp:1604-1853 */
int orc_projection(const orc_model_point *pts, int nPts, float NormDen, const float *angle, int isQuat, int N,
                   float pixelSize, int shiftX, int shiftY, float *realmap, float *spec)
{
  float rotmat[3][3];
  float *proj = (float *) calloc((size_t) N * N, sizeof(float));
  int dropped = 0;
  if (isQuat)
  {
    const float q0 = angle[0], q1 = angle[1], q2 = angle[2], q3 = angle[3];
    rotmat[0][0] = 1 - 2 * q1 * q1 - 2 * q2 * q2; /* :1638-1646 */
    rotmat[1][0] = 2 * (q0 * q1 - q2 * q3);
    rotmat[2][0] = 2 * (q0 * q2 + q1 * q3);
    rotmat[0][1] = 2 * (q0 * q1 + q2 * q3);
    rotmat[1][1] = 1 - 2 * q0 * q0 - 2 * q2 * q2;
    rotmat[2][1] = 2 * (q1 * q2 - q0 * q3);
    rotmat[0][2] = 2 * (q0 * q2 - q1 * q3);
    rotmat[1][2] = 2 * (q1 * q2 + q0 * q3);
    rotmat[2][2] = 1 - 2 * q0 * q0 - 2 * q1 * q1;
  }
  else
  {
    const float alpha = angle[0], beta = angle[1], gam = angle[2]; /* :1653-1672, float cos/sin */
    rotmat[0][0] = cosf(gam) * cosf(alpha) - cosf(beta) * sinf(alpha) * sinf(gam);
    rotmat[0][1] = cosf(gam) * sinf(alpha) + cosf(beta) * cosf(alpha) * sinf(gam);
    rotmat[0][2] = sinf(gam) * sinf(beta);
    rotmat[1][0] = -sinf(gam) * cosf(alpha) - cosf(beta) * sinf(alpha) * cosf(gam);
    rotmat[1][1] = -sinf(gam) * sinf(alpha) + cosf(beta) * cosf(alpha) * cosf(gam);
    rotmat[1][2] = cosf(gam) * sinf(beta);
    rotmat[2][0] = sinf(beta) * sinf(alpha);
    rotmat[2][1] = -sinf(beta) * cosf(alpha);
    rotmat[2][2] = cosf(beta);
  }
  float tempden = 0.0f;
  for (int n = 0; n < nPts; n++)
  {
    float rp[3] = {0.f, 0.f, 0.f}; /* :1677-1693 */
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 3; j++)
        rp[k] += rotmat[k][j] * pts[n].pos[j];
    const float radius = pts[n].radius, density = pts[n].density;
    if (radius <= pixelSize)
    {
      /* :1715-1741 point branch, no shift */
      const int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f);
      const int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f);
      if (i < 0 || j < 0 || i >= N || j >= N)
        dropped++;
      else
      {
        proj[i * N + j] += density;
        tempden += density;
      }
    }
    else
    {
      /* :1742-1803 sphere branch */
      const int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f) - shiftX;
      const int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f) - shiftY;
      const int irad = (int) (radius / pixelSize) + 1;
      const float rad2 = radius * radius;
      if (i < irad || j < irad || i >= N - irad || j >= N - irad)
        dropped++;
      else
      {
        for (int ii = i - irad; ii < i + irad + 1; ii++)
          for (int jj = j - irad; jj < j + irad + 1; jj++)
          {
            const float dist = ((float) (ii - i) * (ii - i) + (jj - j) * (jj - j)) * pixelSize * pixelSize;
            if (dist < rad2)
            {
              const double w = pixelSize * pixelSize * 2 * sqrtf(rad2 - dist) * density * 3 /
                               (4 * M_PI * radius * rad2);
              proj[ii * N + jj] += w;
              tempden += w;
            }
          }
      }
    }
  }
  const float ratioDen = NormDen / tempden; /* :1810 */
  for (int i = 0; i < N * N; i++)
    proj[i] *= ratioDen;
  if (realmap)
    memcpy(realmap, proj, sizeof(float) * (size_t) N * N);
  if (spec)
    orc_fft2_r2c(N, proj, spec); /* :1848 */
  free(proj);
  return dropped;
}

/* reference: bioem.cpp:1855-1923 */
void orc_convolve(int N, const float *proj, const float *ctf, float *out, float *sumC, float *sumsquareC)
{
  const int H = N / 2 + 1;
  for (int i = 0; i < N * H; i++)
  {
    out[2 * i] = (proj[2 * i] * ctf[2 * i] + proj[2 * i + 1] * ctf[2 * i + 1]);
    out[2 * i + 1] = (proj[2 * i + 1] * ctf[2 * i] - proj[2 * i] * ctf[2 * i + 1]);
  }
  *sumC = out[0];
  float ss = 0;
  int jloopend = H;
  if ((N & 1) == 0)
    jloopend--;
  for (int i = 0; i < N; i++)
  {
    for (int j = 1; j < jloopend; j++)
    {
      const int k = i * H + j;
      ss += (out[2 * k] * out[2 * k] + out[2 * k + 1] * out[2 * k + 1]) * 2;
    }
    int k = i * H;
    ss += out[2 * k] * out[2 * k] + out[2 * k + 1] * out[2 * k + 1];
    if ((N & 1) == 0)
    {
      k += H - 1;
      ss += out[2 * k] * out[2 * k] + out[2 * k + 1] * out[2 * k + 1];
    }
  }
  const float norm2 = (float) (N * N);
  *sumsquareC = ss / norm2;
}

/* reference: bioem_algorithm.h:18-70 */
double orc_calc_logpro(const orc_param_device *pd, float amp, float pha, float env, float sum, float sumsquare,
                       float cc, float sumref, float sumsquareref)
{
  const float Ntotpi = pd->Ntotpi;
  const double ForLogProb = sumsquare * Ntotpi - sum * sum; /* float expression, widened */
  const double firstele = Ntotpi * (sumsquareref * sumsquare - cc * cc) + 2 * sumref * sum * cc -
                          sumsquareref * sum * sum - sumref * sumref * sumsquare; /* float expression, widened */
  double logpro = (3 - Ntotpi) * 0.5 * log(firstele) + (Ntotpi * 0.5 - 2) * log((Ntotpi - 2) * ForLogProb);
  if (!pd->tousepsf)
  {
    logpro -= env * env / 2. / pd->sigmaPriorbctf / pd->sigmaPriorbctf -
              (pha - pd->Priordefcent) * (pha - pd->Priordefcent) / 2. / pd->sigmaPriordefo / pd->sigmaPriordefo -
              (amp - pd->Priorampcent) * (amp - pd->Priorampcent) / 2. / pd->sigmaPrioramp / pd->sigmaPrioramp;
  }
  else
  {
    const double envF = 4. * M_PI * M_PI * env / (env * env + pha * pha);
    const double phaF = 4. * M_PI * M_PI * pha / (env * env + pha * pha);
    logpro -= envF * envF / 2. / pd->sigmaPriorbctf / pd->sigmaPriorbctf -
              (phaF - pd->Priordefcent) * (phaF - pd->Priordefcent) / 2. / pd->sigmaPriordefo / pd->sigmaPriordefo -
              (amp - pd->Priorampcent) * (amp - pd->Priorampcent) / 2. / pd->sigmaPrioramp / pd->sigmaPrioramp;
  }
  return logpro;
}

/* reference: bioem_algorithm.h:72-142 (calProb) */
static void cal_prob(const orc_param_device *pd, int iRefMap, int iOrient, int iConv, const orc_param5 *p5,
                     float value, int disx, int disy, float sumref, float sumsqref, orc_prob_map *pm,
                     orc_prob_angle *pa)
{
  const float logpro =
      (float) orc_calc_logpro(pd, p5->amp, p5->pha, p5->env, p5->sumC, p5->sumsquareC, value, sumref, sumsqref);
  (void) iRefMap;
  if (pm->Constoadd < logpro)
  {
    pm->Total *= exp(-logpro + pm->Constoadd);
    pm->Constoadd = logpro;
    pm->max_prob_cent_x = -disx;
    pm->max_prob_cent_y = -disy;
    pm->max_prob_orient = iOrient;
    pm->max_prob_conv = iConv;
    pm->max_prob_norm =
        -(-p5->sumC * sumref + pd->Ntotpi * value) / (p5->sumC * p5->sumC - p5->sumsquareC * pd->Ntotpi);
    pm->max_prob_mu =
        -(-p5->sumC * value + p5->sumsquareC * sumref) / (p5->sumC * p5->sumC - p5->sumsquareC * pd->Ntotpi);
  }
  pm->Total += exp(logpro - pm->Constoadd);
  if (pd->writeAngles && pa)
  {
    if (pa->ConstAngle < logpro)
    {
      pa->forAngles *= exp(-logpro + pa->ConstAngle);
      pa->ConstAngle = logpro;
    }
    pa->forAngles += exp(logpro - pa->ConstAngle);
  }
}

/* reference: bioem.cpp:1435-1459 (calculateCCFFT): float spectrum product, unnormalised c2r */
static void cc_map_work(int N, const float *conv, const float *ref, float *cct, float *lCC, cpx *work)
{
  const int H = N / 2 + 1;
  for (int i = 0; i < N * H; i++)
  {
    cct[2 * i] = conv[2 * i] * ref[2 * i] + conv[2 * i + 1] * ref[2 * i + 1];
    cct[2 * i + 1] = conv[2 * i + 1] * ref[2 * i] - conv[2 * i] * ref[2 * i + 1];
  }
  c2r_work(N, cct, lCC, work);
}

void orc_cc_map(int N, const float *convFFT, const float *refFFT, float *lCC)
{
  const int H = N / 2 + 1;
  float *cct = (float *) malloc(sizeof(float) * 2 * (size_t) N * H);
  cpx *work = (cpx *) malloc(sizeof(cpx) * ((size_t) N * H + 2 * (size_t) N));
  cc_map_work(N, convFFT, refFFT, cct, lCC, work);
  free(work);
  free(cct);
}

/* reference: bioem_algorithm.h:144-198 (doRefMapFFT): quadrant visiting order */
static void do_refmap_fft(const orc_param_device *pd, int iRefMap, int iOrient, int iConv, const orc_param5 *p5,
                          const float *lCC, float sumref, float sumsqref, orc_prob_map *pm, orc_prob_angle *pa)
{
  const int N = pd->NumberPixels, maxD = pd->maxDisplaceCenter, g = pd->GridSpaceCenter;
  const float nn = (float) (N * N);
  for (int cx = 0; cx <= maxD; cx += g)
  {
    for (int cy = 0; cy <= maxD; cy += g)
      cal_prob(pd, iRefMap, iOrient, iConv, p5, lCC[cx * N + cy] / nn, cx, cy, sumref, sumsqref, pm, pa);
    for (int cy = N - maxD; cy < N; cy += g)
      cal_prob(pd, iRefMap, iOrient, iConv, p5, lCC[cx * N + cy] / nn, cx, cy - N, sumref, sumsqref, pm, pa);
  }
  for (int cx = N - maxD; cx < N; cx += g)
  {
    for (int cy = 0; cy <= maxD; cy += g)
      cal_prob(pd, iRefMap, iOrient, iConv, p5, lCC[cx * N + cy] / nn, cx - N, cy, sumref, sumsqref, pm, pa);
    for (int cy = N - maxD; cy < N; cy += g)
      cal_prob(pd, iRefMap, iOrient, iConv, p5, lCC[cx * N + cy] / nn, cx - N, cy - N, sumref, sumsqref, pm, pa);
  }
}

typedef struct
{
  double logpro;
  int id;
  double sumExp;
  float value;
} blk_t; /* defs.h:150-156 myblockCPU_t */

/* reference: bioem.cpp:1461-1515 (doRefMap_CPU_Parallel) */
static void algo2_parallel(const orc_param_device *pd, int iConv, const float *lCC, const orc_param5 *p5,
                           float sumref, float sumsqref, blk_t *blk)
{
  const int N = pd->NumberPixels;
  int myGlobalId = iConv * pd->NtotDisp;
  float bestLogpro = (float) MIN_PROB;
  const int dispC = N - pd->maxDisplaceCenter;
  int bestId = 0;
  float bestValue = 0.f;
  double sumExp = 0.;
  for (int myX = 0; myX < pd->NxDisp; myX++)
    for (int myY = 0; myY < pd->NxDisp; myY++, myGlobalId++)
    {
      const int cx = (myX * pd->GridSpaceCenter + dispC) % N;
      const int cy = (myY * pd->GridSpaceCenter + dispC) % N;
      const float value = lCC[cx * N + cy] / (float) (N * N);
      const double logpro =
          orc_calc_logpro(pd, p5->amp, p5->pha, p5->env, p5->sumC, p5->sumsquareC, value, sumref, sumsqref);
      if (bestLogpro < logpro)
      {
        sumExp *= exp(-logpro + bestLogpro);
        bestLogpro = logpro; /* narrowed to float as in the reference (:1503) */
        bestId = myGlobalId;
        bestValue = value;
      }
      sumExp += exp(logpro - bestLogpro);
    }
  blk->logpro = bestLogpro;
  blk->sumExp = sumExp;
  blk->id = bestId;
  blk->value = bestValue;
}

/* reference: bioem.cpp:1517-1602 (doRefMap_CPU_Reduce) */
static void algo2_reduce(const orc_param_device *pd, int iOrient, int iConvStart, int nConv, const orc_param5 *p5,
                         const blk_t *blk, float sumref, orc_prob_map *pm, orc_prob_angle *pa)
{
  const int N = pd->NumberPixels;
  for (int i = 0; i < nConv; i++)
  {
    if (pm->Constoadd < blk[i].logpro)
    {
      pm->Total *= exp(-blk[i].logpro + pm->Constoadd);
      pm->Constoadd = blk[i].logpro;
      int id = blk[i].id;
      const int myConv = id / pd->NtotDisp;
      id -= myConv * pd->NtotDisp;
      const int myX = id / pd->NxDisp;
      id -= myX * pd->NxDisp;
      const int myY = id;
      const int dispC = N - pd->maxDisplaceCenter;
      const float value = blk[i].value;
      pm->max_prob_cent_x = -((myX * pd->GridSpaceCenter + dispC) - N);
      pm->max_prob_cent_y = -((myY * pd->GridSpaceCenter + dispC) - N);
      pm->max_prob_orient = iOrient;
      pm->max_prob_conv = iConvStart + myConv;
      pm->max_prob_norm = -(-p5[myConv].sumC * sumref + pd->Ntotpi * value) /
                          (p5[myConv].sumC * p5[myConv].sumC - p5[myConv].sumsquareC * pd->Ntotpi);
      pm->max_prob_mu = -(-p5[myConv].sumC * value + p5[myConv].sumsquareC * sumref) /
                        (p5[myConv].sumC * p5[myConv].sumC - p5[myConv].sumsquareC * pd->Ntotpi);
    }
    pm->Total += blk[i].sumExp * exp(blk[i].logpro - pm->Constoadd);
    if (pd->writeAngles && pa)
    {
      if (pa->ConstAngle < blk[i].logpro)
      {
        pa->forAngles *= exp(-blk[i].logpro + pa->ConstAngle);
        pa->ConstAngle = blk[i].logpro;
      }
      pa->forAngles += blk[i].sumExp * exp(blk[i].logpro - pa->ConstAngle);
    }
  }
}

/* reference: bioem.cpp:1379-1433 (compareRefMaps).  ALGO 1: OpenMP over particles (:1392).
 * pang layout: [angle][map] (map.h:147-150). */
void orc_compare(const orc_param_device *pd, int algo, int nMaps, int nAnglesTotal, const float *refFFT,
                 const float *sumRef, const float *sumsqRef, int iOrient, int iConvStart, int nConv,
                 const float *convFFT, const orc_param5 *params, orc_prob_map *pmap, orc_prob_angle *pang)
{
  const int N = pd->NumberPixels, H = N / 2 + 1;
  const size_t M = (size_t) N * H;
  (void) nAnglesTotal;
#pragma omp parallel
  {
    float *cct = (float *) malloc(sizeof(float) * 2 * M);
    float *lCC = (float *) malloc(sizeof(float) * (size_t) N * N);
    cpx *work = (cpx *) malloc(sizeof(cpx) * (M + 2 * (size_t) N));
    blk_t *blk = (blk_t *) malloc(sizeof(blk_t) * (size_t) (nConv > 0 ? nConv : 1));
#pragma omp for schedule(dynamic, 1)
    for (int iRefMap = 0; iRefMap < nMaps; iRefMap++)
    {
      orc_prob_angle *pa = (pd->writeAngles && pang) ? &pang[(size_t) iOrient * nMaps + iRefMap] : NULL;
      for (int iConv = 0; iConv < nConv; iConv++)
      {
        cc_map_work(N, convFFT + 2 * M * (size_t) iConv, refFFT + 2 * M * (size_t) iRefMap, cct, lCC, work);
        if (algo == 1)
          do_refmap_fft(pd, iRefMap, iOrient, iConvStart + iConv, &params[iConv], lCC, sumRef[iRefMap],
                        sumsqRef[iRefMap], &pmap[iRefMap], pa);
        else
          algo2_parallel(pd, iConv, lCC, &params[iConv], sumRef[iRefMap], sumsqRef[iRefMap], &blk[iConv]);
      }
      if (algo != 1)
        algo2_reduce(pd, iOrient, iConvStart, nConv, params, blk, sumRef[iRefMap], &pmap[iRefMap], pa);
    }
    free(blk);
    free(work);
    free(lCC);
    free(cct);
  }
}

/* reference: bioem.cpp:681-699 */
void orc_init_prob(int nMaps, int nAngles, int writeAngles, orc_prob_map *pmap, orc_prob_angle *pang)
{
  for (int i = 0; i < nMaps; i++)
  {
    memset(&pmap[i], 0, sizeof(orc_prob_map));
    pmap[i].Total = 0.0;
    pmap[i].Constoadd = MIN_PROB;
  }
  if (writeAngles && pang)
    for (size_t i = 0; i < (size_t) nMaps * nAngles; i++)
    {
      pang[i].forAngles = 0.0;
      pang[i].ConstAngle = MIN_PROB;
    }
}

/* reference: bioem.cpp:763-891 (one conv per compareRefMaps call, the ALGO-1 default batching) */
void orc_run(const orc_param_device *pd, int algo, const orc_model_point *pts, int nPts, float NormDen,
             const float *angles, int nAnglesTotal, int isQuat, float pixelSize, int shiftX, int shiftY, int nCTF,
             const float *refCTF, const float *ctfParam, int nMaps, const float *refFFT, const float *sumRef,
             const float *sumsqRef, int o0, int o1, orc_prob_map *pmap, orc_prob_angle *pang)
{
  const int N = pd->NumberPixels, H = N / 2 + 1;
  const size_t M = (size_t) N * H;
  float *proj = (float *) malloc(sizeof(float) * 2 * M);
  float *conv = (float *) malloc(sizeof(float) * 2 * M);
  for (int iOrient = o0; iOrient < o1; iOrient++)
  {
    orc_projection(pts, nPts, NormDen, angles + 4 * (size_t) iOrient, isQuat, N, pixelSize, shiftX, shiftY, NULL,
                   proj);
    for (int iConv = 0; iConv < nCTF; iConv++)
    {
      orc_param5 p5;
      orc_convolve(N, proj, refCTF + 2 * M * (size_t) iConv, conv, &p5.sumC, &p5.sumsquareC);
      p5.amp = ctfParam[3 * iConv + 0];
      p5.pha = ctfParam[3 * iConv + 1];
      p5.env = ctfParam[3 * iConv + 2];
      orc_compare(pd, algo, nMaps, nAnglesTotal, refFFT, sumRef, sumsqRef, iOrient, iConv, 1, conv, &p5, pmap,
                  pang);
    }
  }
  free(conv);
  free(proj);
}

/* reference: bioem.cpp:909-994.  C* = max_s C_s; Total* = sum_s Total_s exp(C_s - C*); arg-max
 * parameters from the shard holding C*.  Ties: the reference's MPI path takes the highest rank
 * (bioem.cpp:946-949); the serial path the first orientation.  Here: lowest shard (= lowest
 * orientation index), matching the serial semantics. */
void orc_merge(int nShards, int nMaps, const orc_prob_map *shards, orc_prob_map *out)
{
  for (int i = 0; i < nMaps; i++)
  {
    double cmax = shards[i].Constoadd;
    int who = 0;
    for (int s = 1; s < nShards; s++)
      if (shards[(size_t) s * nMaps + i].Constoadd > cmax)
      {
        cmax = shards[(size_t) s * nMaps + i].Constoadd;
        who = s;
      }
    double tot = 0.;
    for (int s = 0; s < nShards; s++)
      tot += shards[(size_t) s * nMaps + i].Total * exp(shards[(size_t) s * nMaps + i].Constoadd - cmax);
    out[i] = shards[(size_t) who * nMaps + i];
    out[i].Total = tot;
    out[i].Constoadd = cmax;
  }
}

/* reference: bioem.cpp:1144-1150 */
double orc_final_logp(const orc_param_device *pd, double Total, double Constoadd)
{
  return log(Total) + Constoadd + 0.5 * log(M_PI) + (1 - pd->Ntotpi * 0.5) * (log(2 * M_PI) + 1) + log(pd->volu);
}

int orc_sizeof_prob_map(void) { return (int) sizeof(orc_prob_map); }

/* thread count of the OpenMP regions (callers pass the CPUs they may actually use) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
  if (n > 0)
    omp_set_num_threads(n);
#else
  (void) n;
#endif
}

int orc_get_max_threads(void) { return omp_get_max_threads(); }
