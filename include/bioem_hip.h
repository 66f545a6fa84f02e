/* bioem_hip.h -- C ABI of the MI355X-native BioEM likelihood engine (libbioem_hip.so).
 *
 * This is the drop-in boundary for the reference's accelerator plugin (class bioem_cuda,
 * /root/reference/include/bioem_cuda_internal.h:28-85, created by bioem_cuda_create(),
 * /root/reference/include/bioem_cuda.h:20): plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference interface it replaces.  All functions return 0 on
 * success and a non-zero code on failure (bioem_hip_last_error() gives the text); the C++
 * shim (bioem_amd/host) maps non-zero to the reference's print-and-exit (defs.h:18-26).
 *
 * Threading: all entry points of one handle are called from one host thread (the reference
 * calls its plugin from the main thread only, bioem.cpp:853).  One handle drives one GPU.
 */
#ifndef BIOEM_HIP_H
#define BIOEM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == bioem_param_device, /root/reference/include/param.h:26-47 (60 bytes, bool tousepsf widened) */
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
} bioem_hip_param_device;

/* == myparam5_t, /root/reference/include/defs.h:128-135 (20 bytes) */
typedef struct
{
  float amp, pha, env, sumC, sumsquareC;
} bioem_hip_param5;

/* == bioem_Probability_map, /root/reference/include/map.h:116-128 (40 bytes) */
typedef struct
{
  double Total;
  double Constoadd;
  int max_prob_cent_x, max_prob_cent_y, max_prob_orient, max_prob_conv;
  float max_prob_norm, max_prob_mu;
} bioem_hip_prob_map;

/* == bioem_Probability_angle, /root/reference/include/map.h:130-135 (16 bytes) */
typedef struct
{
  double forAngles;
  double ConstAngle;
} bioem_hip_prob_angle;

/* == bioem_model::bioem_model_point, /root/reference/include/model.h:23-29 (24 bytes) */
typedef struct
{
  float pos[3];
  float quat4_unused;
  float radius;
  float density;
} bioem_hip_model_point;

/* One of the K most probable orientations of a particle (WRITE_PROB_ANGLES): the angle entry of orientation
 * `orient` (global index) and logp = log(forAngles) + ConstAngle + numconst, the key the reference's writer ranks
 * by (/root/reference/bioem.cpp:1251-1286).  32 bytes. */
typedef struct
{
  double forAngles;
  double ConstAngle;
  double logp;
  int orient;
  int pad;
} bioem_hip_angle_candidate;

typedef struct bioem_hip_ctx *bioem_hip_handle;

/* Number of visible HIP devices (replaces bioem_cuda::selectCudaDevice, bioem_cuda.cu:686-816). */
int bioem_hip_device_count(void);

/* Replaces bioem_cuda::deviceInit (bioem_cuda.cu:818-951): allocate device state for nMaps particles,
 * nAngles orientations, nCTF kernels.  algo = BIOEM_ALGO (1 or 2: displacement set + reduction
 * semantics of bioem_algorithm.h:144-198 / bioem.cpp:1461-1602).  device = HIP ordinal.
 * Environment, read here: BIOEM_CC_DIRECT=1 makes the handle evaluate the cross-correlation of bioem.cpp:1435-1459 as
 * a sliding window in real space instead of through the transform (BASELINE config 4; no counterpart in the
 * reference, doc/index.rst:1658-1663): images up to 160 pixels, regular windows of at most 24 offsets per axis,
 * particles through bioem_hip_upload_particle_maps; create fails with a message otherwise. */
int bioem_hip_create(bioem_hip_handle *out, int device, const bioem_hip_param_device *pd, int nMaps, int nAngles,
                     int nCTF, int algo);
/* Shard variant of bioem_hip_create for orientation-sharded runs (the reference's MPI blocks, bioem.cpp:748-753): this
 * handle only ever compares orientations [iOrientBegin, iOrientEnd) of the nAngles global ones.  With
 * WRITE_PROB_ANGLES its angle table is [iOrientEnd - iOrientBegin][nMaps] (each orientation has one owner) and stays on
 * the device: start_run / finish_run move only the nMaps 40-byte map entries (bioem_hip_prob_size(nMaps, 0, 0)
 * bytes), the K best orientations per particle come from bioem_hip_topk_angles.  Orientation and CTF indices in all
 * calls and results stay GLOBAL. */
int bioem_hip_create_shard(bioem_hip_handle *out, int device, const bioem_hip_param_device *pd, int nMaps, int nAngles,
                           int nCTF, int algo, int iOrientBegin, int iOrientEnd);
int bioem_hip_destroy(bioem_hip_handle h); /* bioem_cuda::deviceExit, bioem_cuda.cu:1023-1053 */
const char *bioem_hip_last_error(bioem_hip_handle h);

/* Particle side.  refFFT = bioem_RefMap::RefMapsFFT [nMaps][N][N/2+1] (re,im), sum/sumsq =
 * sum_RefMap / sumsquare_RefMap (map.h:65-74; uploaded at bioem_cuda.cu:824-846). */
int bioem_hip_upload_particles(bioem_hip_handle h, const float *refFFT, const float *sum, const float *sumsq);
/* North-star variant: real-space maps [nMaps][N][N]; sums (bioem.cpp:2087-2107, same float summation
 * order) and r2c (map.cpp:557-601) run on the device. */
int bioem_hip_upload_particle_maps(bioem_hip_handle h, const float *maps);

/* bioem_param::refCTF [nCTF][N][N/2+1] (re,im) and CtfParam [nCTF] as {amp, phase, env} triples
 * (param.h:76-77, filled at param.cpp:1336-1583). */
int bioem_hip_upload_ctf(bioem_hip_handle h, const float *refCTF, const float *ctfParam3);
/* bioem_model::points / NormDen and the projection parameters used by bioem::createProjection
 * (bioem.cpp:1604-1853). */
int bioem_hip_upload_model(bioem_hip_handle h, const bioem_hip_model_point *pts, int nPts, float NormDen,
                           float pixelSize, int shiftX, int shiftY);
/* bioem_param::angles [n] as {pos0,pos1,pos2,quat4} (defs.h:105-110); isQuat = param.doquater. */
int bioem_hip_upload_orientations(bioem_hip_handle h, const float *angles4, int n, int isQuat);

/* bioem::malloc_device_host / free_device_host (bioem.h:56-57; bioem_cuda.cu:1037-1053): pinned host memory
 * for the probability block. */
void *bioem_hip_host_alloc(size_t size);
void bioem_hip_host_free(void *ptr);
/* size of the probability block, == bioem_Probability::get_size (map.h:156-162) */
size_t bioem_hip_prob_size(int nMaps, int nAngles, int writeAngles);

/* bioem_cuda::deviceStartRun (bioem_cuda.cu:953-1011): upload the (initialised) probability block. */
int bioem_hip_start_run(bioem_hip_handle h, const void *pProb_host);

/* Reference-compatible hot-path entry == bioem::compareRefMaps (bioem.h:52-54; CUDA override
 * bioem_cuda.cu:527-684).  conv_mapsFFT / comp_params are the BASE pointers of the caller's 2-slot buffers;
 * slot offset k = (iPipeline & 1) * nTotParallelConv as in bioem.cpp:1388.  Asynchronous like the CUDA
 * plugin: returns after enqueue; slot k may be overwritten after the next call with the same parity
 * has returned. */
int bioem_hip_compare(bioem_hip_handle h, int iPipeline, int iOrient, int iConvStart, int maxParallelConv,
                      int nTotParallelConv, const float *conv_mapsFFT, const bioem_hip_param5 *comp_params);

/* North-star entry: bioem::createProjection + createConvolutedProjectionMap + compareRefMaps
 * (the body of the run() loop, bioem.cpp:763-891) for orientations [iOrientBegin, iOrientEnd) and all CTFs,
 * entirely on the device. */
int bioem_hip_project_convolve_compare(bioem_hip_handle h, int iOrientBegin, int iOrientEnd);

/* The same for orientations [iOrientBegin, iOrientEnd) and CTFs [iConvBegin, iConvEnd) only: when there are fewer
 * orientations than GPUs the CTF grid is split as well (north_star: "orientations x CTF-envelope grid shard"). */
int bioem_hip_project_convolve_compare_ctf(bioem_hip_handle h, int iOrientBegin, int iOrientEnd, int iConvBegin,
                                           int iConvEnd);

/* The same three stages as separate entries, for an integrator who keeps the reference's loop (bioem.cpp:763-891) and
 * replaces its body piece by piece: every call is asynchronous and batched, and what one stage produces stays on the
 * device, in the comparison's layout, for the next.  iPipeline & 1 names one of two buffer sets, as in
 * bioem::compareRefMaps (bioem.cpp:1388): stage calls on one set are ordered, the two sets overlap (projection and
 * convolution of set 1 run beside the comparison of set 0).  Results are bit-identical to
 * bioem_hip_project_convolve_compare over the same (orientation, CTF) rows in the same order.
 *   bioem_hip_project         == bioem::createProjection (bioem.h:61, bioem.cpp:1604-1853) for [iOrientBegin, iOrientEnd),
 *                                at most maxOrientations of bioem_hip_max_batch per call
 *   bioem_hip_convolve        == bioem::createConvolutedProjectionMap (bioem.h:43-45, bioem.cpp:1855-1923) for every
 *                                projection of the set x CTFs [iConvBegin, iConvEnd); rows (orientation-major) <= maxRows
 *   bioem_hip_compare_device  == bioem::compareRefMaps (bioem.h:52-54) for the conv spectra of the set against all particles */
int bioem_hip_project(bioem_hip_handle h, int iPipeline, int iOrientBegin, int iOrientEnd);
int bioem_hip_convolve(bioem_hip_handle h, int iPipeline, int iConvBegin, int iConvEnd);
int bioem_hip_compare_device(bioem_hip_handle h, int iPipeline);
/* capacity of a buffer set: orientations per bioem_hip_project call, (orientation, CTF) rows per bioem_hip_convolve call */
int bioem_hip_max_batch(bioem_hip_handle h, int *maxOrientations, int *maxRows);

/* bioem_cuda::deviceFinishRun (bioem_cuda.cu:1013-1021): synchronise, download the probability block. */
int bioem_hip_finish_run(bioem_hip_handle h, void *pProb_host);

/* WRITE_PROB_ANGLES without moving the table: selects on the device the K best orientations of every particle among
 * the orientations this handle owns, with the reference writer's own rule (a K-entry min-heap on (logp, orientation)
 * walked in orientation order, bioem.cpp:1251-1286; numconst as computed at bioem.cpp:1141-1150).  out = [nMaps][K],
 * best first; entries beyond the owned orientation count have orient = -1.  Call after finish_run. */
int bioem_hip_topk_angles(bioem_hip_handle h, int K, double numconst, bioem_hip_angle_candidate *out);
/* K-way merge of the shards' candidate lists (shards in ascending orientation-block order): the K best of the union,
 * same heap rule.  cands[s] = [nMaps][K]; out = [nMaps][K], best first. */
int bioem_hip_merge_topk_host(int nShards, int nMaps, int K, const bioem_hip_angle_candidate *const *cands,
                              bioem_hip_angle_candidate *out);

/* Log-sum-exp merge of orientation shards (replaces the MPI merge of bioem.cpp:909-1044) on the host:
 * shards = nShards probability blocks of identical shape, out = merged block.  Ties on Constoadd go to
 * the lowest shard (= lowest orientation index, the serial semantics). */
int bioem_hip_merge_host(int nShards, int nMaps, int nAngles, int writeAngles, const void *const *shards, void *out);

/* The path's single exchange step over RCCL / xGMI (replaces the MPI merge of bioem.cpp:909-1044) for n handles on n
 * DIFFERENT GPUs of this process: every device contributes its nMaps map entries -- and, when K > 0, its K best
 * orientations per particle (bioem_hip_topk_angles) -- to ONE ncclAllGather; the device of handles[0] folds the
 * gathered shards (log-sum-exp of Total / Constoadd, arg-max record from the lowest shard holding the maximum =
 * lowest orientation index; candidates by the heap rule) and the result is copied to the host.
 * pProbMaps_host = [nMaps] bioem_hip_prob_map, cand_host = [nMaps][K] or NULL when K == 0.  Call after every handle's
 * finish_run.  The communicator is created on first use (ncclCommInitAll) and cached; librccl.so is loaded lazily.
 * Threading: ONE thread hands over all handles of a merge; calls are serialised process-wide (a mutex spans communicator
 * creation and the grouped all-gather), so merges of different handle sets from different threads queue -- they cannot
 * interleave their RCCL group calls. */
int bioem_hip_merge(bioem_hip_handle *handles, int n, void *pProbMaps_host, int K, double numconst,
                    bioem_hip_angle_candidate *cand_host);

/* Stand-alone forward transform (FFTW r2c convention, [N][N/2+1] (re,im)) of nImg real N x N images on
 * `device`; replaces the fftwf_execute_dft_r2c call of the PSF kernel set-up (param.cpp:1521). */
int bioem_hip_r2c(int device, int N, int nImg, const float *in, float *out);

/* The reference's only built-in profile, BIOEM_DEBUG_OUTPUT >= 1 (TimeStat, timer.cpp:138-165; the "Time Projection /
 * Convolution / Comparison" lines of bioem.cpp:769-889): device time of every phase of every batch, from HIP events on
 * the streams the phases run on.  Off by default; records accumulate from set_phase_timing(h, 1) on and are handed over
 * (and dropped) by phase_records after finish_run: *n = records available, the first min(*n, cap) are written. */
enum
{
  BIOEM_HIP_PHASE_PROJECTION = 0, /* bioem::createProjection of orientations [iOrientBegin, iOrientEnd) */
  BIOEM_HIP_PHASE_CONVOLUTION = 1, /* createConvolutedProjectionMap of those x CTFs [iConvBegin, iConvEnd) */
  BIOEM_HIP_PHASE_COMPARISON = 2   /* compareRefMaps of those rows against all particles, fold included */
};
typedef struct
{
  int phase;
  int iOrientBegin, iOrientEnd, iConvBegin, iConvEnd; /* -1: rows handed over through bioem_hip_compare */
  double seconds;
} bioem_hip_phase_record;
int bioem_hip_set_phase_timing(bioem_hip_handle h, int on);
int bioem_hip_phase_records(bioem_hip_handle h, bioem_hip_phase_record *out, int cap, int *n);

/* ---- instrumentation / test hooks (no reference equivalent) ---- */
/* projection spectrum of one orientation in reference layout [N][N/2+1][2] */
int bioem_hip_debug_projection(bioem_hip_handle h, int iOrient, float *spec_out);
/* conv spectrum + {sumC, sumsquareC} of (iOrient, iConv) in reference layout */
int bioem_hip_debug_convolution(bioem_hip_handle h, int iOrient, int iConv, float *spec_out, float *sumC,
                                float *sumsquareC);
/* download device-side particle precompute (reference layout) */
int bioem_hip_debug_particles(bioem_hip_handle h, float *refFFT_out, float *sum_out, float *sumsq_out);
/* accumulated HIP-event time of the comparison kernel on the engine's stream since the last reset:
 * total ms, launches, comparisons processed */
int bioem_hip_kernel_stats(bioem_hip_handle h, double *compare_ms, long long *launches, long long *comparisons);
int bioem_hip_reset_kernel_stats(bioem_hip_handle h);
/* 1 if the LDS-FFT fast path is used for this configuration, 0 for the generic pruned-DFT path */
int bioem_hip_uses_fast_path(bioem_hip_handle h);
/* name of the comparison kernel this configuration runs: "k_compare_fast", "k_compare_wide" (tiled wide window),
 * "k_compare_oddfft" / "k_compare_rows" (odd image size with / without a factor 3 or 5) or "k_compare_generic" */
const char *bioem_hip_kernel_name(bioem_hip_handle h);
/* the same with the template arguments of the instantiation, spelled as rocprofv3 prints them (e.g.
 * "k_compare_fast<10, 32, false, 1>"): lets bench.py tie its live timing to the committed counter profile of exactly
 * this kernel */
const char *bioem_hip_kernel_signature(bioem_hip_handle h);
/* Which comparison kernel a configuration would run, without a device (no reference counterpart; the selection is a
 * pure function of the image size and the displacement set, bioem_amd/csrc/kernel_select.hpp).  Writes the
 * instantiation -- and " x T^2 tiles of R rows" for a tiled wide window -- into signature[cap]; 0 on success, 1 when no
 * kernel fits, 2 on invalid arguments. */
int bioem_hip_plan(int numberPixels, int maxDisplaceCenter, int gridSpaceCenter, int algo, char *signature, int cap);
int bioem_hip_synchronize(bioem_hip_handle h);

#ifdef __cplusplus
}
#endif
#endif
