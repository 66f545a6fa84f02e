// prep_kernels.hpp -- comparison layout, projection, exact-DFT r2c, particle sums, CTF convolution
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_PREP_KERNELS_HPP
#define BIOEM_PREP_KERNELS_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// layout of a half spectrum used by the comparison kernels
//   fast   : N = N1*R (R = 32, 16, 8, 4 or 2), kx = N1*k2 + k1  ->  float2 index ((k1*R/2 + (k2>>1))*H + ky)*2 + (k2&1)
//            (lane = ky reads 16 B = two k2 of one k1 -> fully coalesced dwordx4, and the R inputs
//             of one register FFT arrive as R/2 such loads).  The `fast` argument carries R/2 (0 = generic).
//   generic: reference layout kx*H + ky
// Hp >= H is the pitch of a row pair in 16-byte words (comparison_pitch, bioem_hip.hip): H, or H + 15 where the rows
// of a column block would otherwise all start in the same few L2 channels (N a multiple of 128); an image then
// takes N * Hp float2.  The reference layout is never padded.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t layout_index(int fast, int N1, int H, int kx, int ky, int Hp = 0)
{
  if (!fast)
    return (size_t) kx * H + ky;
  const int k1 = kx % N1, k2 = kx / N1;
  return ((size_t) (k1 * fast + (k2 >> 1)) * (Hp ? Hp : H) + ky) * 2 + (k2 & 1);
}

__global__ void k_reorder(const float2 *__restrict__ src, float2 *__restrict__ dst, int nImg, int N, int H, int fast,
                          int N1, int Hp)
{
  const size_t M = (size_t) N * H, Mc = (size_t) N * Hp;
  const size_t total = M * nImg;
  for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t) gridDim.x * blockDim.x)
  {
    const size_t img = e / M;
    const int r = (int) (e - img * M);
    const int kx = r / H, ky = r - kx * H;
    dst[img * Mc + layout_index(fast, N1, H, kx, ky, Hp)] = src[e];
  }
}

__global__ void k_unreorder(const float2 *__restrict__ src, float2 *__restrict__ dst, int nImg, int N, int H,
                            int fast, int N1, int Hp)
{
  const size_t M = (size_t) N * H, Mc = (size_t) N * Hp;
  const size_t total = M * nImg;
  for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t) gridDim.x * blockDim.x)
  {
    const size_t img = e / M;
    const int r = (int) (e - img * M);
    const int kx = r / H, ky = r - kx * H;
    dst[e] = src[img * Mc + layout_index(fast, N1, H, kx, ky, Hp)];
  }
}

__device__ inline void rotation_matrix(const float4 a, int isQuat, float (&rotmat)[3][3])
{
  if (isQuat)
  {
    const float q0 = a.x, q1 = a.y, q2 = a.z, q3 = a.w; // bioem.cpp:1632-1646
    rotmat[0][0] = 1 - 2 * q1 * q1 - 2 * q2 * q2;
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
    const float alpha = a.x, beta = a.y, gam = a.z; // bioem.cpp:1653-1672
    const float ca = cosf(alpha), sa = sinf(alpha), cb = cosf(beta), sb = sinf(beta), cg = cosf(gam), sg = sinf(gam);
    rotmat[0][0] = cg * ca - cb * sa * sg;
    rotmat[0][1] = cg * sa + cb * ca * sg;
    rotmat[0][2] = sg * sb;
    rotmat[1][0] = -sg * ca - cb * sa * cg;
    rotmat[1][1] = -sg * sa + cb * ca * cg;
    rotmat[1][2] = cg * sb;
    rotmat[2][0] = sb * sa;
    rotmat[2][1] = -sb * ca;
    rotmat[2][2] = cb;
  }
}

// ------------------------------------------------------------------------------------------------
// projection: bioem.cpp:1604-1818 (rotation, point / sphere splat, tempden)
// one thread per model point, blockIdx.y = orientation inside the batch
// ------------------------------------------------------------------------------------------------
__global__ void k_project(const bioem_hip_model_point *__restrict__ pts, int nPts, const float4 *__restrict__ angles,
                          int o0, int isQuat, int N, float pixelSize, int shiftX, int shiftY,
                          double *__restrict__ proj, double *__restrict__ tempden)
{
  const int ob = blockIdx.y;
  float rotmat[3][3];
  rotation_matrix(angles[o0 + ob], isQuat, rotmat);
  double td = 0.;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  double *map = proj + (size_t) ob * N * N;
  if (n < nPts)
  {
    const bioem_hip_model_point p = pts[n];
    float rp[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 3; j++)
        rp[k] += rotmat[k][j] * p.pos[j];
    const float radius = p.radius, density = p.density;
    if (radius <= pixelSize)
    {
      const int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f);
      const int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f);
      if (!(i < 0 || j < 0 || i >= N || j >= N))
      {
        atomicAdd(&map[i * N + j], (double) density);
        td += (double) density;
      }
    }
    else
    {
      const int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f) - shiftX;
      const int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f) - shiftY;
      const int irad = (int) (radius / pixelSize) + 1;
      const float rad2 = radius * radius;
      if (!(i < irad || j < irad || i >= N - irad || j >= N - irad))
      {
        for (int ii = i - irad; ii < i + irad + 1; ii++)
          for (int jj = j - irad; jj < j + irad + 1; jj++)
          {
            const float dist = ((float) (ii - i) * (ii - i) + (jj - j) * (jj - j)) * pixelSize * pixelSize;
            if (dist < rad2)
            {
              const double w = (double) (pixelSize * pixelSize * 2 * sqrtf(rad2 - dist) * density * 3) /
                               (4 * M_PI * radius * rad2);
              atomicAdd(&map[ii * N + jj], w);
              td += w;
            }
          }
      }
    }
  }
  // block reduction of tempden
  __shared__ double red[256];
  red[threadIdx.x] = td;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1)
  {
    if ((int) threadIdx.x < s)
      red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0] != 0.)
    atomicAdd(&tempden[ob], red[0]);
}

// The same projection in two steps, without global atomics or a zero-filled map (used while a band of >= 12 rows fits
// 40 KiB of LDS, N <= 426):
//   k_project_stamps  once per model: the weights of every sphere's footprint.  The reference evaluates them from the
//                     integer pixel offsets to the sphere's centre pixel (bioem.cpp:1760-1790), so they do not depend
//                     on the orientation: (2 iradMax + 1)^2 doubles per point, 0 outside the sphere;
//   k_project_coords  rotates every model point once per orientation and leaves a 16-byte record: its pixel
//                     (i << 16 | j, or -1 when the reference skips it: outside the map / sphere touching the border),
//                     the point's index, density and the sphere's half width in pixels (0 for a point);
//   k_project_bands   one block per band of TR map rows: lists the records that reach the band (512 points at a time,
//                     2 048 loaded at once), splats them with one thread per (sphere, column of its
//                     footprint) into LDS with double atomics and stores the band once.
// The additions are the same doubles in another order; the weight is the reference's expression.
struct alignas(16) ProjectRecord
{
  int ij;
  int n;
  float density;
  int irad;
};

// stamp[n][di + iradMax][dj + iradMax]: what the sphere of point n adds to the pixel (di, dj) away from its centre pixel
__global__ void k_project_stamps(const bioem_hip_model_point *__restrict__ pts, int nPts, int iradMax, float pixelSize,
                                 double *__restrict__ stamp)
{
  const int S = 2 * iradMax + 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nPts * S * S)
    return;
  const int n = e / (S * S), c = e - n * S * S;
  const int di = c / S - iradMax, dj = c - (c / S) * S - iradMax;
  const float radius = pts[n].radius, density = pts[n].density;
  double w = 0.;
  if (radius > pixelSize)
  {
    const int irad = (int) (radius / pixelSize) + 1;
    const float rad2 = radius * radius;
    // the reference's loop variables: ii - i = di, jj - j = dj
    const float dist = ((float) (di) * (di) + (dj) * (dj)) * pixelSize * pixelSize;
    if (di >= -irad && di <= irad && dj >= -irad && dj <= irad && dist < rad2)
      w = (double) (pixelSize * pixelSize * 2 * sqrtf(rad2 - dist) * density * 3) / (4 * M_PI * radius * rad2);
  }
  stamp[e] = w;
}

__global__ void k_project_coords(const bioem_hip_model_point *__restrict__ pts, int nPts,
                                 const float4 *__restrict__ angles, int o0, int isQuat, int N, float pixelSize,
                                 int shiftX, int shiftY, ProjectRecord *__restrict__ coords)
{
  const int ob = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nPts)
    return;
  float rotmat[3][3];
  rotation_matrix(angles[o0 + ob], isQuat, rotmat);
  const bioem_hip_model_point p = pts[n];
  float rp[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 3; j++)
      rp[k] += rotmat[k][j] * p.pos[j];
  int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f);
  int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f);
  bool ok;
  int irad = 0;
  if (p.radius <= pixelSize)
    ok = !(i < 0 || j < 0 || i >= N || j >= N);
  else
  {
    i -= shiftX;
    j -= shiftY;
    irad = (int) (p.radius / pixelSize) + 1;
    ok = !(i < irad || j < irad || i >= N - irad || j >= N - irad);
  }
  ProjectRecord r;
  r.ij = ok ? (i << 16 | j) : -1;
  r.n = n;
  r.density = p.density;
  r.irad = irad;
  coords[(size_t) ob * nPts + n] = r;
}

constexpr int kProjectList = 512;

__global__ __launch_bounds__(256) void k_project_bands(const ProjectRecord *__restrict__ coords, int nPts, int nO, int N,
                                                        int TR, int iradMax, const double *__restrict__ stamp,
                                                        double *__restrict__ proj, double *__restrict__ tempden)
{
  extern __shared__ double band[]; // TR x N
  __shared__ ProjectRecord list[kProjectList];
  __shared__ int cnt;
  __shared__ double red[4];
  const int nBands = (N + TR - 1) / TR;
  const int S = 2 * iradMax + 1;
  // a resident grid walks the (orientation, band) units: launching a 50 KiB-LDS block costs more than one short unit
  for (int unit = blockIdx.x; unit < nO * nBands; unit += gridDim.x)
  {
    const int ob = unit / nBands;
    const int r0 = (unit - ob * nBands) * TR, r1 = min(N, r0 + TR);
    const ProjectRecord *C = coords + (size_t) ob * nPts;
    __syncthreads(); // the previous band is stored
    for (int e = threadIdx.x; e < (r1 - r0) * N; e += blockDim.x)
      band[e] = 0.;
    double td = 0.;
    // 2 048 records per round sit in registers (one wait for global memory), listed and splatted 512 at a time
    for (int n00 = 0; n00 < nPts; n00 += 4 * kProjectList)
    {
      ProjectRecord rec[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        const int n = n00 + u * 256 + threadIdx.x;
        rec[u] = C[min(n, nPts - 1)];
        rec[u].ij = n < nPts ? rec[u].ij : -1;
      }
#pragma unroll
      for (int g = 0; g < 4; g++)
      {
        if (n00 + g * kProjectList >= nPts)
          break;
        if (threadIdx.x == 0)
          cnt = 0;
        __syncthreads();
#pragma unroll
        for (int u = 2 * g; u < 2 * g + 2; u++)
        {
          const ProjectRecord r = rec[u];
          const int i = r.ij >> 16;
          if (r.ij >= 0 && i + r.irad >= r0 && i - r.irad < r1)
            list[atomicAdd(&cnt, 1)] = r;
        }
        __syncthreads();
        const int items = cnt * S;
        for (int it = threadIdx.x; it < items; it += blockDim.x)
        {
          const int en = it / S, dj = it - en * S - iradMax;
          const ProjectRecord q = list[en];
          const int i = q.ij >> 16, j = q.ij & 0xffff;
          if (q.irad == 0)
          { // a point: bioem.cpp:1700-1712
            if (dj == 0)
            {
              atomicAdd(&band[(i - r0) * N + j], (double) q.density);
              td += (double) q.density;
            }
            continue;
          }
          if (dj < -q.irad || dj > q.irad)
            continue;
          const int jj = j + dj;
          const double *st = stamp + ((size_t) q.n * S + iradMax - i) * S + dj + iradMax; // row ii: st[ii * S]
          for (int ii = max(i - q.irad, r0); ii < min(i + q.irad + 1, r1); ii++)
          {
            const double w = st[ii * S];
            if (w != 0.) // inside the sphere (dist < rad2: the weight is positive there)
            {
              atomicAdd(&band[(ii - r0) * N + jj], w);
              td += w;
            }
          }
        }
        __syncthreads(); // the list is rewritten by the next points
      }
    }
    for (int o = 32; o > 0; o >>= 1)
      td += __shfl_down(td, o);
    if ((threadIdx.x & 63) == 0)
      red[threadIdx.x >> 6] = td;
    __syncthreads(); // also: every splat of this band is done
    if (threadIdx.x == 0)
    {
      const double t = (red[0] + red[1]) + (red[2] + red[3]);
      if (t != 0.)
        atomicAdd(&tempden[ob], t);
    }
    double *map = proj + (size_t) ob * N * N + (size_t) r0 * N;
    for (int e = threadIdx.x; e < (r1 - r0) * N; e += blockDim.x)
      map[e] = band[e];
  }
}

// The projection of a model that stays inside a box of pixels around the map centre whatever the orientation
// (box side = 2 (max |point| / pixelSize + widest footprint) + a margin, 52 KiB of doubles at most: 81 pixels): one block
// per orientation keeps the box in LDS, every thread takes (point, footprint column) items straight from the model --
// rotation, pixel, the reference's skip rules as in k_project_coords, the footprint from the stamps -- and the whole
// map (zeros around the box) is stored once.  Against k_project_coords + k_project_bands there is no record list, no
// band that scans all points to find its few, and no barrier between zeroing and storing; tempden is the block's own
// sum.  COMPACT: only the box is stored, [ob][side][side] (the fast r2c skips what lies outside); else the whole map.
// The host guarantees that no point leaves the box (unit quaternions only, bioem_hip_upload_orientations); one that
// did would be dropped.
__global__ __launch_bounds__(256) void k_project_box(const bioem_hip_model_point *__restrict__ pts, int nPts,
                                                      const float4 *__restrict__ angles, int o0, int isQuat, int N,
                                                      float pixelSize, int shiftX, int shiftY, int iradMax,
                                                      const double *__restrict__ stamp, int lo, int side, int nO,
                                                      int compact, double *__restrict__ proj,
                                                      double *__restrict__ tempden)
{
  extern __shared__ double box[]; // side x side
  __shared__ double red[4];
  const int S = 2 * iradMax + 1;
  const int hi = lo + side - 1;
  for (int ob = blockIdx.x; ob < nO; ob += gridDim.x)
  {
    __syncthreads(); // the previous map is stored
    for (int e = threadIdx.x; e < side * side; e += blockDim.x)
      box[e] = 0.;
    float rotmat[3][3];
    rotation_matrix(angles[o0 + ob], isQuat, rotmat);
    __syncthreads();
    double td = 0.;
    {
      // Two items per thread and round: both points, then both footprint columns (five stamp entries each) are on
      // their way before the first is used -- an item alone is a chain of two memory latencies and an LDS atomic.
      struct Item
      {
        bool act, point;
        int irad;
        float density;
        double *dst;
        const double *st;
      };
      auto decode = [&](int it) -> Item {
        Item q;
        const bool in = it < nPts * S;
        const int n = in ? it / S : 0, dj = it - n * S - iradMax;
        const bioem_hip_model_point p = pts[n];
        float rp[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; k++)
          for (int j = 0; j < 3; j++)
            rp[k] += rotmat[k][j] * p.pos[j];
        int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f);
        int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f);
        bool ok;
        int irad = 0;
        if (p.radius <= pixelSize)
          ok = !(i < 0 || j < 0 || i >= N || j >= N) && dj == 0;
        else
        {
          i -= shiftX;
          j -= shiftY;
          irad = (int) (p.radius / pixelSize) + 1;
          ok = !(i < irad || j < irad || i >= N - irad || j >= N - irad) && dj >= -irad && dj <= irad;
        }
        ok = ok && in;
        const bool inbox = i - irad >= lo && i + irad <= hi && j - irad >= lo && j + irad <= hi;
        q.act = ok && inbox;
        q.point = irad == 0;
        q.irad = irad;
        q.density = p.density;
        q.dst = box + (i - lo) * side + (j + dj - lo);
        q.st = stamp + ((size_t) n * S + iradMax) * S + dj + iradMax;
        return q;
      };
      const int ld = side;
      auto fetch = [&](const Item &q, int d0, double (&w)[5]) {
#pragma unroll
        for (int u = 0; u < 5; u++)
          w[u] = (q.act && !q.point && d0 + u <= q.irad) ? q.st[(d0 + u) * S] : 0.;
      };
      auto splat = [&](const Item &q, int d0, const double (&w)[5]) {
#pragma unroll
        for (int u = 0; u < 5; u++)
          if (w[u] != 0.) // inside the sphere (dist < rad2: the weight is positive there)
          {
            atomicAdd(q.dst + (d0 + u) * ld, w[u]);
            td += w[u];
          }
      };
      for (int it = threadIdx.x; it < nPts * S; it += 2 * blockDim.x)
      {
        const Item qa = decode(it), qb = decode(it + blockDim.x);
        double wa[5], wb[5];
        fetch(qa, -qa.irad, wa);
        fetch(qb, -qb.irad, wb);
        if (qa.act && qa.point)
        { // a point: bioem.cpp:1700-1712
          atomicAdd(qa.dst, (double) qa.density);
          td += (double) qa.density;
        }
        if (qb.act && qb.point)
        {
          atomicAdd(qb.dst, (double) qb.density);
          td += (double) qb.density;
        }
        splat(qa, -qa.irad, wa);
        splat(qb, -qb.irad, wb);
        // footprints wider than five pixels: the rest of the column
        for (int d0 = -qa.irad + 5; d0 <= qa.irad; d0 += 5)
        {
          fetch(qa, d0, wa);
          splat(qa, d0, wa);
        }
        for (int d0 = -qb.irad + 5; d0 <= qb.irad; d0 += 5)
        {
          fetch(qb, d0, wb);
          splat(qb, d0, wb);
        }
      }
      __syncthreads(); // every splat is done
      if (compact)
      {
        double *dst = proj + (size_t) ob * side * side;
        for (int e = threadIdx.x; e < side * side; e += blockDim.x)
          dst[e] = box[e];
      }
      else
      {
        double *map = proj + (size_t) ob * N * N;
        for (int e = threadIdx.x; e < N * N; e += blockDim.x)
        {
          const int r = e / N, c = e - r * N;
          map[e] = (r >= lo && r <= hi && c >= lo && c <= hi) ? box[(r - lo) * side + c - lo] : 0.;
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1)
      td += __shfl_down(td, o);
    if ((threadIdx.x & 63) == 0)
      red[threadIdx.x >> 6] = td;
    __syncthreads();
    if (threadIdx.x == 0)
      tempden[ob] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// r2c as two exact-DFT passes with double accumulation (FFTW forward convention, unnormalised).
// rows: src is either the double projection map scaled by NormDen/tempden in float (bioem.cpp:1808-1818)
//       or float particle maps.
// ------------------------------------------------------------------------------------------------
// Both passes use one Cooley-Tukey split N = A*B (A the largest divisor <= sqrt(N); A = 1 for prime N):
//   Y[j1][kb] = sum_{j2<B} x[A*j2 + j1] * w_B^(j2*kb),   X[k] = sum_{j1<A} w_N^(j1*k) * Y[j1][k mod B]
// i.e. N*(A+B) instead of N*N terms per 1-D transform, still exact-DFT arithmetic in double.
// TWLDS: the twiddle table sits in LDS as well (every term of both passes reads one entry); images too large for
// that (beyond ~3 400 pixels) read it from global memory as before
template <bool TWLDS>
__global__ void k_dft_rows(const double *__restrict__ srcD, const float *__restrict__ srcF,
                           const double *__restrict__ tempden, float NormDen, int N, int H, int A, int B,
                           const double2 *__restrict__ twD, double2 *__restrict__ rowspec)
{
  extern __shared__ double srow[];          // N doubles, then N double2 (Y), then N double2 (twiddles)
  double2 *Y = reinterpret_cast<double2 *>(srow + N + (N & 1));
  double2 *twl = Y + N;
  if (TWLDS)
    for (int j = threadIdx.x; j < N; j += blockDim.x)
      twl[j] = twD[j];
  const double2 *tw = TWLDS ? twl : twD;
  const int i = blockIdx.x, b = blockIdx.y;
  float ratio = 1.f;
  if (srcD)
    ratio = NormDen / (float) tempden[b];
  for (int j = threadIdx.x; j < N; j += blockDim.x)
  {
    float v;
    if (srcD)
    {
      v = (float) srcD[((size_t) b * N + i) * N + j];
      v = v * ratio;
    }
    else
      v = srcF[((size_t) b * N + i) * N + j];
    srow[j] = (double) v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N; e += blockDim.x)
  {
    const int j1 = e / B, kb = e - j1 * B;
    double ar = 0., ai = 0.;
    int idx = 0;
    const int step = (A * kb) % N;
    for (int j2 = 0; j2 < B; j2++)
    {
      const double2 w = tw[idx];
      const double x = srow[A * j2 + j1];
      ar = fma(x, w.x, ar);
      ai = fma(-x, w.y, ai); // forward: e^{-i}
      idx += step;
      if (idx >= N)
        idx -= N;
    }
    Y[e] = make_double2(ar, ai);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < H; k += blockDim.x)
  {
    const int kb = k % B;
    double ar = 0., ai = 0.;
    int idx = 0;
    for (int j1 = 0; j1 < A; j1++)
    {
      const double2 w = tw[idx]; // multiply by conj(w)
      const double2 y = Y[j1 * B + kb];
      ar = fma(y.x, w.x, ar);
      ar = fma(y.y, w.y, ar);
      ai = fma(y.y, w.x, ai);
      ai = fma(-y.x, w.y, ai);
      idx += k;
      if (idx >= N)
        idx -= N;
    }
    rowspec[((size_t) b * N + i) * H + k] = make_double2(ar, ai);
  }
}

template <bool TWLDS>
__global__ void k_dft_cols(const double2 *__restrict__ rowspec, int N, int H, int A, int B,
                           const double2 *__restrict__ twD, float2 *__restrict__ out)
{
  extern __shared__ double srow[];
  double2 *col = reinterpret_cast<double2 *>(srow);
  double2 *Y = col + N;
  double2 *twl = Y + N;
  if (TWLDS)
    for (int i = threadIdx.x; i < N; i += blockDim.x)
      twl[i] = twD[i];
  const double2 *tw = TWLDS ? twl : twD;
  const int k = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < N; i += blockDim.x)
    col[i] = rowspec[((size_t) b * N + i) * H + k];
  __syncthreads();
  for (int e = threadIdx.x; e < N; e += blockDim.x)
  {
    const int j1 = e / B, kb = e - j1 * B;
    double ar = 0., ai = 0.;
    int idx = 0;
    const int step = (A * kb) % N;
    for (int j2 = 0; j2 < B; j2++)
    {
      const double2 w = tw[idx];
      const double2 x = col[A * j2 + j1];
      ar = fma(x.x, w.x, ar);
      ar = fma(x.y, w.y, ar);
      ai = fma(x.y, w.x, ai);
      ai = fma(-x.x, w.y, ai);
      idx += step;
      if (idx >= N)
        idx -= N;
    }
    Y[e] = make_double2(ar, ai);
  }
  __syncthreads();
  for (int u = threadIdx.x; u < N; u += blockDim.x)
  {
    const int kb = u % B;
    double ar = 0., ai = 0.;
    int idx = 0;
    for (int j1 = 0; j1 < A; j1++)
    {
      const double2 w = tw[idx];
      const double2 y = Y[j1 * B + kb];
      ar = fma(y.x, w.x, ar);
      ar = fma(y.y, w.y, ar);
      ai = fma(y.y, w.x, ai);
      ai = fma(-y.x, w.y, ai);
      idx += u;
      if (idx >= N)
        idx -= N;
    }
    out[(size_t) b * N * H + (size_t) u * H + k] = make_float2((float) ar, (float) ai);
  }
}

// particle sums, bioem.cpp:2087-2107: sequential float accumulation in row-major order.
__global__ void k_map_sums(const float *__restrict__ maps, int NN, float *__restrict__ sum, float *__restrict__ sumsq)
{
  __shared__ float buf[4096];
  const float *m = maps + (size_t) blockIdx.x * NN;
  float s = 0.0f, s2 = 0.0f;
  for (int base = 0; base < NN; base += 4096)
  {
    const int cnt = min(4096, NN - base);
    for (int t = threadIdx.x; t < cnt; t += blockDim.x)
      buf[t] = m[base + t];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int t = 0; t < cnt; t++)
      {
        s += buf[t];
        s2 += buf[t] * buf[t];
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    sum[blockIdx.x] = s;
    sumsq[blockIdx.x] = s2;
  }
}

// ------------------------------------------------------------------------------------------------
// convolution: bioem.cpp:1855-1923.  grid (CTFs of the launch, nOrientInBatch); CTF index = c0 + blockIdx.x.
// k_convolve forms conv = proj * conj(CTF) in the comparison layout, sumC, and the terms of sumsquareC in the
// reference's summation order (rows; inside a row the interior columns doubled, then column 0, then column N/2 for
// even N) in `scratch` (row stride M4 = M rounded up to 4 floats).  The sum itself is a SEQUENTIAL float chain in the
// reference (bioem.cpp:1896-1914) and bit-pinned to it: ~7 cycles per term whatever is done.  k_parseval_ordered makes
// the chains cheap to run side by side: one WAVE carries four chains (lanes 0..3 add, one spectrum each) and feeds
// itself -- all 64 lanes fetch the four spectra's next tiles of 512 terms with coalesced 16-byte loads, two tiles
// ahead, and pass them through 16 KiB of LDS -- so ten waves (40 chains) fit a CU and a batch of 10 000 spectra takes
// about one chain's duration (before: one block of four waves and 32 KiB per chain, 1 280 chains in flight).
// ------------------------------------------------------------------------------------------------
__global__ void k_convolve(const float2 *__restrict__ proj, const float2 *__restrict__ ctf,
                           const float *__restrict__ ctfParam, int N, int H, int fast, int N1, int c0,
                           float2 *__restrict__ conv, float *__restrict__ scratch, int M4,
                           bioem_hip_param5 *__restrict__ params, int Hp)
{
  const int c = c0 + blockIdx.x, ob = blockIdx.y;
  const int oc = ob * gridDim.x + blockIdx.x;
  const int M = N * H;
  const float2 *P = proj + (size_t) ob * M;
  const float2 *K = ctf + (size_t) c * M;
  float2 *O = conv + (size_t) oc * N * Hp; // (Hp == H without the fast layout)
  float *S = scratch + (size_t) oc * M4;
  const int even = ((N & 1) == 0);
  const int jend = even ? H - 1 : H;
  // one spectrum element: product, its Parseval term at its place in the reference's summation order
  auto element = [&](int i, int j, float2 &o) {
    const float2 p = P[(size_t) i * H + j], k = K[(size_t) i * H + j];
    o.x = (p.x * k.x + p.y * k.y);
    o.y = (p.y * k.x - p.x * k.y);
    const float t = o.x * o.x + o.y * o.y;
    int pos;
    if (j >= 1 && j < jend)
      pos = i * H + (j - 1);
    else if (j == 0)
      pos = i * H + (jend - 1);
    else
      pos = i * H + jend; // j == H-1, even N
    S[pos] = (j >= 1 && j < jend) ? t * 2 : t;
    if (i == 0 && j == 0)
    {
      bioem_hip_param5 r;
      r.amp = ctfParam[3 * c + 0];
      r.pha = ctfParam[3 * c + 1];
      r.env = ctfParam[3 * c + 2];
      r.sumC = o.x;
      r.sumsquareC = 0.f; // k_parseval_ordered
      params[oc] = r;
    }
  };
  if (fast)
  { // comparison layout: the rows kx = N1 (2 k2p) + k1 and kx + N1 of a (k1, k2 pair) sit side by side -- one thread
    // forms both and writes them as ONE 16-byte word, consecutive threads (ky) consecutive words
    for (int e = threadIdx.x; e < (M >> 1); e += blockDim.x)
    {
      const int rp = e / H, j = e - rp * H; // row pair rp = k1 * fast + k2p
      const int k1 = rp / fast, k2p = rp - k1 * fast;
      const int i0 = N1 * (2 * k2p) + k1;
      float2 o0, o1;
      element(i0, j, o0);
      element(i0 + N1, j, o1);
      reinterpret_cast<float4 *>(O)[(size_t) rp * Hp + j] = make_float4(o0.x, o0.y, o1.x, o1.y);
    }
  }
  else
    for (int e = threadIdx.x; e < M; e += blockDim.x)
    {
      const int i = e / H, j = e - i * H;
      float2 o;
      element(i, j, o);
      O[(size_t) i * H + j] = o;
    }
}

__global__ __launch_bounds__(64) void k_parseval_ordered(const float *__restrict__ scratch, int M, int M4, int nSpec,
                                                          float norm2, bioem_hip_param5 *__restrict__ params)
{
  constexpr int CH = 4;        // chains per wave
  constexpr int TILE = 512;    // terms per chain and tile
  constexpr int TS = TILE + 4; // chain stride in LDS (floats): the four adding lanes read different banks
  __shared__ __align__(16) float buf[2][CH][TS];
  const int lane = threadIdx.x;
  const int oc0 = blockIdx.x * CH;
  const int n4 = M >> 2; // whole float4 per spectrum; the last M % 4 terms are added at the end
  const int nt = (n4 + TILE / 4 - 1) / (TILE / 4);
  // load j of a tile (0..7): chain j / 2, float4 (j % 2) * 64 + lane of the tile
  auto fetch = [&](int tile, float4 (&dst)[2 * CH]) {
#pragma unroll
    for (int j = 0; j < 2 * CH; j++)
    {
      const int c = j >> 1, f = tile * (TILE / 4) + (j & 1) * 64 + lane;
      const int oc = min(oc0 + c, nSpec - 1);
      dst[j] = f < n4 ? reinterpret_cast<const float4 *>(scratch + (size_t) oc * M4)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float4 pre0[2 * CH], pre1[2 * CH];
  fetch(0, pre0);
  if (nt > 1)
    fetch(1, pre1);
  float ss = 0.f;
  for (int i = 0; i < nt; i++)
  {
    const int b = i & 1;
    // (the previous tile in this buffer was consumed two iterations ago; LDS operations of one wave execute in order)
    if (b == 0)
    {
#pragma unroll
      for (int j = 0; j < 2 * CH; j++)
        *reinterpret_cast<float4 *>(&buf[0][j >> 1][((j & 1) * 64 + lane) * 4]) = pre0[j];
      if (i + 2 < nt)
        fetch(i + 2, pre0);
    }
    else
    {
#pragma unroll
      for (int j = 0; j < 2 * CH; j++)
        *reinterpret_cast<float4 *>(&buf[1][j >> 1][((j & 1) * 64 + lane) * 4]) = pre1[j];
      if (i + 2 < nt)
        fetch(i + 2, pre1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < CH)
    { // the reference's order: one term after the other (tiles beyond the spectrum hold zeros: x + 0 = x)
      const float4 *q = reinterpret_cast<const float4 *>(&buf[b][lane][0]);
#pragma unroll 8
      for (int k = 0; k < TILE / 4; k++)
      {
        const float4 v = q[k];
        ss += v.x;
        ss += v.y;
        ss += v.z;
        ss += v.w;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < CH && oc0 + lane < nSpec)
  {
    const float *row = scratch + (size_t) (oc0 + lane) * M4;
    for (int e = n4 << 2; e < M; e++)
      ss += row[e];
    params[oc0 + lane].sumsquareC = ss / norm2;
  }
}

// ------------------------------------------------------------------------------------------------
// k_convolve_sums: the two kernels above in one -- no Parseval terms through global memory.
// One block per (R orientations, group of up to NC CTFs).  Waves 1..15 form the products of a tile of kConvTile
// consecutive positions of the reference's summation order for every CTF of the group (the projection element is read
// once), store the spectra and leave the terms in LDS; wave 0 adds the previous tile meanwhile, one lane per
// (orientation, CTF), one term after the other as the reference does.  In the comparison layout a 16-byte word holds the rows kx and kx + N1:
// it is written when the first of them comes by (its partner's product is formed there for the store, and once more
// when its own position in the order is reached: same operands, same bits).
// ------------------------------------------------------------------------------------------------
constexpr int kConvThreads = 1024; // one adding wave, fifteen producing waves
// a tile is one position per producing thread: its loads (the projection element, its partner row, the CTFs' values)
// go out together and a tile costs the producers one round trip to memory.  (Round 3 had tiles of 1 024: a second,
// nearly empty round trip for 64 of them -- 256 orientations x 5 CTFs at 224^2: 233 -> 132 us.  Two or three positions
// per thread change nothing: the adding wave is the longer side.)
constexpr int kConvTile = kConvThreads - 64;
constexpr int kConvStride = kConvTile + 4; // chain stride in LDS (floats): the adding lanes read different banks
// block barrier that orders LDS traffic only: __syncthreads() would also wait for the spectra on their way to memory
__device__ inline void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A block takes R orientations x up to NC CTFs, their chains (at most 20: two tiles of 960 terms each in 160 KiB of LDS)
// on the LANES of the adding wave: its instruction stream is the one of a single orientation whatever R is, and a
// producing thread forms the products of its position for the R orientations, the CTF values read once.  Round 4, in the order measured (5 CTFs at 224^2, alone):
//   one orientation per block (round 3, tiles of 1 024 = a second round trip for 64 positions)  233 us per 256 orientations
//   tiles of 960                                                                                  132
//   two / three orientations as interleaved instructions of the adding wave: 200 / 265 us per block (130 for one)
//   chains on lanes, tiles of 240 positions, 12 orientations per block: the producers' round trip (4.3 us) per 1 us of
//     additions, 455 us per block; tiles of 960, 4 orientations: 152 us per block
//   a second operand set fetched a tile ahead: 137 us for 3 orientations (128 registers hold no more)
//   the adding wave's tile as straight-line code (below): 11.2 -> 8.9 cycles per term; the producers (~550 vector
//     instructions per thread and tile at 20 products) are then the longer side of a 4 x 5 block, hence 3 x 6 / 4 x 4
//     blocks with the operands fetched ahead
// Timing-only builds of a one-orientation block on an otherwise idle chip: no stores 130 -> 127 us, no CTF loads 134,
// no additions 73; 32, 64 or 128 terms in flight from LDS, or the adding wave alone on its SIMD, change nothing.
constexpr int kLaneRows = 20; // chains per block at most
inline size_t conv_lanes_lds(int rows) { return sizeof(float) * 2 * rows * kConvStride; }

template <int R, int NC>
__global__ __launch_bounds__(kConvThreads) void
k_convolve_sums(const float2 *__restrict__ proj, const float2 *__restrict__ ctf, const float *__restrict__ ctfParam,
                 int N, int H, int fast, int N1, int c0, int nC, int nO, int rows, float2 *__restrict__ conv,
                 bioem_hip_param5 *__restrict__ params, int Hp)
{
  extern __shared__ __align__(16) float terms[]; // [2][rows][kConvStride], rows >= R x CTFs of the block
  __shared__ float sC[kLaneRows];
  const int ob0 = blockIdx.y * R, cg = blockIdx.x * NC;
  const int nJ = min(R, nO - ob0);
  const int nCb = min(NC, nC - cg);
  const int M = N * H;
  const size_t Mc = (size_t) N * Hp; // an image of the comparison layout (Hp == H without the fast layout)
  const int even = ((N & 1) == 0);
  const int jend = even ? H - 1 : H;
  const float2 *K0 = ctf + (size_t) (c0 + cg) * M;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nt = (M + kConvTile - 1) / kConvTile;
  const int e = (int) threadIdx.x - 64; // producers: position in the tile
  // what a producing thread holds of one tile: the operands of its position and where the results go
  struct Operands
  {
    float2 p[R], p1[R], kv[NC], k1[NC];
    int ij, word; // (a spectrum has fewer than 2^31 elements: kMaxPixels)
    bool valid, twice, first, origin;
  };
  auto fetch = [&](int t, Operands &L) {
    const int pos0 = t * kConvTile + e;
    L.valid = pos0 < M; // beyond the spectrum: the last element is read, a zero term left, nothing stored
    const int pos = L.valid ? pos0 : M - 1;
    const int i = pos / H, r = pos - i * H;
    const int j = r < jend - 1 ? r + 1 : (r == jend - 1 ? 0 : H - 1);
    L.twice = j >= 1 && j < jend;
    L.origin = i == 0 && j == 0;
    L.ij = i * H + j;
    L.first = false;
    L.word = 0;
    int partner = L.ij;
    if (fast)
    {
      const int k2 = i / N1, k1i = i - k2 * N1;
      L.first = !(k2 & 1);
      L.word = (k1i * fast + (k2 >> 1)) * Hp + j;
      partner = L.first ? L.ij + N1 * H : L.ij;
    }
#pragma unroll
    for (int q = 0; q < R; q++)
    {
      const float2 *P = proj + (size_t) (ob0 + min(q, nJ - 1)) * M;
      L.p[q] = P[L.ij];
      L.p1[q] = P[partner];
    }
#pragma unroll
    for (int c = 0; c < NC; c++)
      if (c < nCb)
      {
        const float2 *Kc = K0 + (size_t) c * M;
        L.kv[c] = Kc[L.ij];
        L.k1[c] = Kc[partner];
      }
  };
  auto produce = [&](int t, const Operands &L) {
    float *T = terms + (size_t) (t & 1) * rows * kConvStride;
#pragma unroll
    for (int q = 0; q < R; q++)
    {
      if (q >= nJ)
        break;
      float2 *O0 = conv + ((size_t) (ob0 + q) * nC + cg) * Mc;
#pragma unroll
      for (int c = 0; c < NC; c++)
        if (c < nCb)
        {
          float2 *O = O0 + (size_t) c * Mc;
          const float2 k = L.kv[c];
          float2 v;
          v.x = (L.p[q].x * k.x + L.p[q].y * k.y);
          v.y = (L.p[q].y * k.x - L.p[q].x * k.y);
          const float tt = v.x * v.x + v.y * v.y;
          T[(q * nCb + c) * kConvStride + e] = L.valid ? (L.twice ? tt * 2 : tt) : 0.f; // x + 0 = x
          if (!L.valid)
            continue;
          if (!fast)
            O[L.ij] = v;
          else if (L.first)
          {
            const float2 k1v = L.k1[c];
            float2 v1;
            v1.x = (L.p1[q].x * k1v.x + L.p1[q].y * k1v.y);
            v1.y = (L.p1[q].y * k1v.x - L.p1[q].x * k1v.y);
            reinterpret_cast<float4 *>(O)[L.word] = make_float4(v.x, v.y, v1.x, v1.y);
          }
          if (L.origin)
            sC[q * nCb + c] = v.x;
        }
    }
  };
  float ss = 0.f;
  auto add_tile = [&](int t) {
    if (lane < nJ * nCb)
    {
      const float4 *q = reinterpret_cast<const float4 *>(terms + ((size_t) (t & 1) * rows + lane) * kConvStride);
      const int n4 = (min(kConvTile, M - t * kConvTile) + 3) >> 2; // the last tile ends with the spectrum
      // one term after the other as the reference does (bioem.cpp:1896-1914).  The chain is this wave's instruction
      // stream: a dependent v_add_f32 issues every 6.3 cycles (scripts/micro/dep_add.hip), and every other vector
      // instruction between two of them costs its own four issue cycles on top -- with the address arithmetic of a
      // rolled loop the chain ran at 11-12 cycles per term.  A full tile is therefore straight-line code: 240 reads at
      // immediate offsets from one base register, a ring of eight 16-byte words in flight (28 additions between a read
      // and its use; eight, not more: the LDS counter of s_waitcnt has four bits), nothing else.
      constexpr int RING = 8, W4 = kConvTile / 4;
      if (n4 == W4)
      {
        float4 r[RING];
#pragma unroll
        for (int u = 0; u < RING; u++)
          r[u] = q[u];
#pragma unroll
        for (int k = 0; k < W4; k++)
        {
          const float4 v = r[k % RING];
          ss += v.x;
          ss += v.y;
          ss += v.z;
          ss += v.w;
          if (k + RING < W4)
            r[k % RING] = q[k + RING];
          __builtin_amdgcn_sched_barrier(0); // (left alone, the scheduler sinks the reads next to their use)
        }
      }
      else
      { // the last tile ends with the spectrum
#pragma unroll 4
        for (int k = 0; k < n4; k++)
        {
          const float4 v = q[k];
          ss += v.x;
          ss += v.y;
          ss += v.z;
          ss += v.w;
        }
      }
    }
  };
  // The producers fetch a tile ahead: the operands of tile t + 2 are requested before the products of tile t + 1 are
  // formed (two operand sets: what 128 registers hold for 3 orientations x 6 CTFs or 4 x 4), so that a tile costs them
  // their ~550 vector instructions and not a round trip to memory on top.  The two roles are two loops with the same
  // number of barriers (in one loop the operand sets would stay live across the adding wave's ring of terms).
#ifdef BIOEM_CONV_TIMING // timing-only build: the adding wave's cycles in additions / in total come back as sumC / sumsquareC
  long long cyAdd = 0, cy0 = clock64();
#endif
  if (wave == 0)
  {
    __builtin_amdgcn_s_setprio(3); // the adding wave is the critical path: it issues ahead of the producers of its SIMD
    lds_barrier();                 // tile 0 is there
    for (int t = 0; t < nt; t++)
    {
#ifdef BIOEM_CONV_TIMING
      asm volatile("" : "+v"(ss));
      const long long c0 = clock64();
#endif
      add_tile(t);
#ifdef BIOEM_CONV_TIMING
      asm volatile("" : "+v"(ss));
      cyAdd += clock64() - c0;
#endif
      lds_barrier();
    }
  }
  else
  {
    Operands A, B;
    fetch(0, A);
    if (1 < nt)
      fetch(1, B);
    produce(0, A);
    lds_barrier();
    for (int t = 0; t < nt; t += 2)
    {
      // while the adding wave has tile t: tile t + 1 from B, tile t + 2 on its way into A
      if (t + 1 < nt)
      {
        if (t + 2 < nt)
          fetch(t + 2, A);
        produce(t + 1, B);
      }
      lds_barrier();
      if (t + 1 >= nt)
        break;
      if (t + 2 < nt)
      {
        if (t + 3 < nt)
          fetch(t + 3, B);
        produce(t + 2, A);
      }
      lds_barrier();
    }
  }
#ifdef BIOEM_CONV_TIMING
  if (wave == 0 && lane < nJ * nCb)
  {
    sC[lane] = (float) cyAdd;
    ss = (float) (clock64() - cy0) * (float) (N * N);
  }
#endif
  if (wave == 0 && lane < nJ * nCb)
  {
    const int o = lane / nCb, cl = lane - o * nCb;
    const int c = c0 + cg + cl;
    bioem_hip_param5 r;
    r.amp = ctfParam[3 * c + 0];
    r.pha = ctfParam[3 * c + 1];
    r.env = ctfParam[3 * c + 2];
    r.sumC = sC[lane];
    r.sumsquareC = ss / (float) (N * N);
    params[(size_t) (ob0 + o) * nC + cg + cl] = r;
  }
}

// sums only (compat entry supplies conv spectra but we never trust host params blindly: they are used as given)

} // namespace

#endif
