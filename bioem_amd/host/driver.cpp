// driver.cpp -- the bioem-shaped driver: command line, configure(), run(), output files.
// Mirrors the control flow of /root/reference/bioem.cpp (readOptions 170-436, configure 438-585,
// run 659-1377) with the hot loop (createProjection -> createConvolutedProjectionMap -> compareRefMaps,
// bioem.cpp:763-891) delegated to the device through include/bioem_hip.h.  Orientations are sharded over
// the visible GPUs in the same contiguous blocks as the reference's MPI ranks (bioem.cpp:748-753) and the
// shards are merged with the log-sum-exp rule of bioem.cpp:909-1044.
#include <getopt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <queue>
#include <thread>

#include "bioem_host.h"

namespace bioem_host
{

namespace
{
void r2c_on_device(void *ctx, int N, const float *in, float *out)
{
  const int device = ctx ? *static_cast<const int *>(ctx) : 0; // the device the run itself uses (GPUDEVICE)
  if (bioem_hip_r2c(device, N, 1, in, out))
    fatal("device r2c failed");
}

// The reference's performance knobs (bioem.cpp:99-135, bioem_cuda.cu:216-222) balance a GPU against CPU cores and tune
// its CUDA kernels.  A job script written for the reference may set any of them: they are read, reported and have
// no effect here -- every comparison runs on the device (there is no CPU path to share the work with), the kernels
// pick their own launch shapes, projection and convolution are batched on the device.
void report_reference_knobs()
{
  static const char *const knobs[] = {"GPUWORKLOAD", "GPUASYNC", "GPUDUALSTREAM", "BIOEM_CUDA_THREAD_COUNT",
                                      "BIOEM_PROJ_CONV_AT_ONCE", "OMP_NUM_THREADS"};
  for (const char *k : knobs)
    if (getenv(k))
      printf("Note - %s=%s accepted, no effect (100 %% of the comparisons run on the device)\n", k, getenv(k));
  if (getenv("GPU") && atoi(getenv("GPU")) == 0)
    printf("Note - GPU=0 accepted, no effect: this engine has no CPU path, the run uses the device\n");
}

void check(bioem_hip_handle h, int rc, const char *what)
{
  if (rc)
    fatal("%s: %s", what, bioem_hip_last_error(h));
}
} // namespace

Driver::Driver()
{
  // environment knobs shared with the reference (bioem.cpp:99-135)
  algo = getenv("BIOEM_ALGO") == NULL ? 1 : atoi(getenv("BIOEM_ALGO"));
  debugOutput = getenv("BIOEM_DEBUG_OUTPUT") == NULL ? 0 : atoi(getenv("BIOEM_DEBUG_OUTPUT"));
}

Driver::~Driver() { cleanup(); }

int Driver::readOptions(int ac, char **av)
{
  std::string infile, modelfile, mapfile, anglefile;
  std::cout << " ++++++++++++ FROM COMMAND LINE +++++++++++\n\n";
  static const struct option opts[] = {{"Modelfile", required_argument, 0, 0},
                                       {"Particlesfile", required_argument, 0, 0},
                                       {"Inputfile", required_argument, 0, 0},
                                       {"PrintBestCalMap", required_argument, 0, 0},
                                       {"ReadOrientation", required_argument, 0, 0},
                                       {"ReadPDB", no_argument, 0, 0},
                                       {"ReadModelMRC", no_argument, 0, 0},
                                       {"ReadMRC", no_argument, 0, 0},
                                       {"ReadMultipleMRC", no_argument, 0, 0},
                                       {"DumpMaps", no_argument, 0, 0},
                                       {"LoadMapDump", no_argument, 0, 0},
                                       {"DumpModel", no_argument, 0, 0},
                                       {"LoadModelDump", no_argument, 0, 0},
                                       {"PrintCOORDREAD", no_argument, 0, 0},
                                       {"OutputFile", required_argument, 0, 0},
                                       {"help", no_argument, 0, 0},
                                       {0, 0, 0, 0}};
  auto usage = []() {
    printf("\nCommand line inputs:\n");
    printf("  --Modelfile arg        (Mandatory) Name of model file\n");
    printf("  --Particlesfile arg    (Mandatory) Name of particle-image file\n");
    printf("  --Inputfile arg        (Mandatory) Name of input parameter file\n");
    printf("  --ReadOrientation arg  (Optional) Read file name containing orientations\n");
    printf("  --ReadPDB              (Optional) If reading model file in PDB format\n");
    printf("  --ReadModelMRC         (Optional) If reading model file in MRC format\n");
    printf("  --ReadMRC              (Optional) If reading particle file in MRC format\n");
    printf("  --ReadMultipleMRC      (Optional) If reading multiple MRCs\n");
    printf("  --OutputFile arg       (Optional) For changing the outputfile name\n");
    printf("  --help                 (Optional) Produce help message\n");
  };
  if (ac < 2)
  {
    printf("Error - Need to specify all mandatory options\n");
    usage();
    return 1;
  }
  optind = 1;
  while (true)
  {
    int idx = 0;
    const int c = getopt_long(ac, av, "", opts, &idx);
    if (c == -1)
      break;
    if (c == '?')
    {
      usage();
      return 1;
    }
    const std::string name = opts[idx].name;
    if (name == "help")
    {
      std::cout << "Usage: options_description [options]\n";
      usage();
      return 1;
    }
    else if (name == "Inputfile")
    {
      std::cout << "Input file is: " << optarg << "\n";
      infile = optarg;
    }
    else if (name == "Modelfile")
    {
      std::cout << "Model file is: " << optarg << "\n";
      modelfile = optarg;
    }
    else if (name == "Particlesfile")
    {
      std::cout << "Particle file is: " << optarg << "\n";
      mapfile = optarg;
    }
    else if (name == "ReadPDB")
    {
      std::cout << "Reading model file in PDB format.\n";
      model.readPDB = true;
    }
    else if (name == "ReadModelMRC")
    {
      std::cout << "Reading model file in MRC format.\n";
      model.readModelMRC = true;
    }
    else if (name == "ReadOrientation")
    {
      std::cout << "Reading Orientation from file: " << optarg << "\n";
      anglefile = optarg;
      param.notuniformangles = true;
    }
    else if (name == "OutputFile")
    {
      std::cout << "Writing OUTPUT to: " << optarg << "\n";
      outfileName = optarg;
    }
    else if (name == "ReadMRC")
    {
      std::cout << "Reading particle file in MRC format.\n";
      particles.readMRC = true;
    }
    else if (name == "ReadMultipleMRC")
    {
      std::cout << "Reading multiple MRCs.\n";
      particles.readMultMRC = true;
    }
    else
      fatal("Option --%s is outside the compare path served by this build (SURVEY.md 8, out of scope)",
            name.c_str());
  }
  if (optind < ac)
  {
    printf("Error - Non-option ARGV-elements: ");
    while (optind < ac)
      printf("%s ", av[optind++]);
    putchar('\n');
    usage();
    return 1;
  }
  if (particles.readMultMRC && !particles.readMRC)
    fatal("For multiple MRCs command --ReadMRC is necessary too");
  param.readParameters(infile.c_str());
  particles.readRefMaps(param, mapfile.c_str());
  model.readModel(param, modelfile.c_str());
  param.calculateGridsParam(anglefile.c_str());
  return 0;
}

int Driver::configure(int ac, char **av)
{
  if (readOptions(ac, av))
    return 1;
  const int ndev = bioem_hip_device_count();
  if (ndev < 1)
    fatal("No HIP device found: this engine has no CPU path");
  report_reference_knobs();
  nGpus = ndev;
  if (getenv("BIOEM_GPUS"))
    nGpus = std::max(1, std::min(ndev, atoi(getenv("BIOEM_GPUS"))));
  // BIOEM_SHARDS=n: n shards dealt round-robin over the selected GPUs (default: one per GPU); more shards than GPUs
  // is only useful to rehearse the multi-GPU control flow and merge on a smaller machine
  int nDevUsed = nGpus;
  int nShards = nGpus;
  if (getenv("BIOEM_SHARDS"))
    nShards = std::max(1, atoi(getenv("BIOEM_SHARDS")));
  firstDev = 0;
  if (getenv("GPUDEVICE") && atoi(getenv("GPUDEVICE")) >= 0) // bioem_cuda.cu:719-732
  {
    firstDev = atoi(getenv("GPUDEVICE"));
    if (!getenv("BIOEM_SHARDS"))
      nShards = 1;
    nDevUsed = 1;
    if (firstDev >= ndev)
      fatal("GPUDEVICE %d out of range (%d devices)", firstDev, ndev);
  }
  param.r2c = r2c_on_device;
  param.r2c_ctx = &firstDev;
  param.calculateRefCTF();
  if (getenv("BIOEM_DEBUG_BREAK")) // bioem.cpp:518-525: after volu was computed with the full counts
  {
    const int cut = atoi(getenv("BIOEM_DEBUG_BREAK"));
    if (param.nTotGridAngles > cut)
      param.nTotGridAngles = cut;
    if (param.nTotCTFs > cut)
      param.nTotCTFs = cut;
  }
  // Work split (north_star: "orientations x CTF-envelope grid shard"): orientation blocks as the reference's MPI ranks
  // (bioem.cpp:748-753); with fewer orientations than shards the (orientation, CTF) pairs are split instead, in the
  // serial visiting order (orientation outer, CTF inner), so that shard s still precedes shard s+1
  const int nA = param.nTotGridAngles, nC = param.nTotCTFs;
  splitCTF = nShards > nA;
  if (splitCTF && (long long) nShards > (long long) nA * nC)
    nShards = nA * nC;
  nGpus = nShards;
  nDevUsed = std::min(nDevUsed, nShards);
  const int nMaps = particles.ntot;
  shards.assign(nShards, Shard());
  for (int g = 0; g < nShards; g++)
  {
    Shard &sh = shards[g];
    sh.device = firstDev + g % nDevUsed;
    if (!splitCTF)
    {
      sh.o0 = (int) ((long long) g * nA / nShards);
      sh.o1 = (int) ((long long) (g + 1) * nA / nShards);
      sh.u0 = (long long) sh.o0 * nC;
      sh.u1 = (long long) sh.o1 * nC;
    }
    else
    {
      sh.u0 = (long long) g * nA * nC / nShards;
      sh.u1 = (long long) (g + 1) * nA * nC / nShards;
      sh.o0 = (int) (sh.u0 / nC);
      sh.o1 = (int) ((sh.u1 + nC - 1) / nC);
    }
    bioem_hip_handle h = nullptr;
    // an orientation block owns its angle entries: shard handle (table [o1-o0][nMaps] stays on the device, K best
    // selected there).  With the CTF split an orientation has several owners: plain handles, full (tiny) table,
    // host merge of the angle entries
    const int rc = splitCTF ? bioem_hip_create(&h, sh.device, &param.pd, nMaps, nA, nC, algo)
                            : bioem_hip_create_shard(&h, sh.device, &param.pd, nMaps, nA, nC, algo, sh.o0, sh.o1);
    sh.h = h;
    check(h, rc, "bioem_hip_create");
    check(h, bioem_hip_upload_particle_maps(h, particles.maps.data()), "upload particles");
    check(h, bioem_hip_upload_ctf(h, param.refCTF.data(), param.ctfParam.data()), "upload CTF");
    check(h, bioem_hip_upload_model(h, model.points.data(), (int) model.points.size(), model.NormDen, param.pixelSize,
                                    param.shiftX, param.shiftY),
          "upload model");
    check(h, bioem_hip_upload_orientations(h, param.angles.data(), nA, param.doquater ? 1 : 0), "upload orientations");
  }
  // RCCL carries the merge when every shard has a GPU of its own
  // (BIOEM_FORCE_RCCL=1: also with a single shard -- a one-rank communicator, to exercise this path on a one-GPU box)
  useRccl = (nShards > 1 || getenv("BIOEM_FORCE_RCCL")) && nDevUsed == nShards && !splitCTF && !getenv("BIOEM_HOST_MERGE");
  return 0;
}

// BIOEM_DEBUG_OUTPUT >= 1: the reference's per-phase report (bioem.cpp:769-889 "Time Projection / Convolution /
// Comparison" lines, TimeStat::PrintTimeStat timer.cpp:156-165 SUMMARY lines) from the engine's HIP-event records -- device
// time per batch of the pipeline; a shard stands where the reference prints its MPI rank
void Driver::print_phase_report()
{
  static const char *names[4] = {"Total time of projection", "Projection", "Convolution", "Comparison"};
  for (int g = 0; g < (int) shards.size(); g++)
  {
    int n = 0;
    if (bioem_hip_phase_records(shards[g].h, nullptr, 0, &n) || n == 0)
      continue;
    std::vector<bioem_hip_phase_record> rec(n);
    if (bioem_hip_phase_records(shards[g].h, rec.data(), n, &n))
      continue;
    std::vector<double> logs[4];
    double batchTotal = 0.;
    for (int i = 0; i < n; i++)
    {
      const bioem_hip_phase_record &r = rec[i];
      if (r.phase == BIOEM_HIP_PHASE_PROJECTION)
      {
        if (debugOutput >= 2)
          printf("\tTime Projection %d-%d: %f (rank %d)\n", r.iOrientBegin, r.iOrientEnd - 1, r.seconds, g);
        logs[1].push_back(r.seconds);
      }
      else if (r.phase == BIOEM_HIP_PHASE_CONVOLUTION)
      {
        if (debugOutput >= 2)
          printf("\t\tTime Convolution %d-%d %d-%d: %f (rank %d)\n", r.iOrientBegin, r.iOrientEnd - 1, r.iConvBegin,
                 r.iConvEnd - 1, r.seconds, g);
        logs[2].push_back(r.seconds);
      }
      else
      {
        if (debugOutput >= 2)
          printf("\t\tTime Comparison %d-%d %d-%d: %f sec (rank %d)\n", r.iOrientBegin, r.iOrientEnd - 1, r.iConvBegin,
                 r.iConvEnd - 1, r.seconds, g);
        logs[3].push_back(r.seconds);
      }
      batchTotal += r.seconds;
      // a batch ends with its (last) comparison: the reference's "Total time for projection" per orientation
      if (r.phase == BIOEM_HIP_PHASE_COMPARISON && (i + 1 == n || rec[i + 1].phase != BIOEM_HIP_PHASE_COMPARISON))
      {
        logs[0].push_back(batchTotal);
        batchTotal = 0.;
      }
    }
    for (int k = 0; k < 4; k++)
    {
      if (logs[k].empty())
        continue;
      double sum = 0., sq = 0.;
      for (double v : logs[k])
        sum += v;
      const double mean = sum / logs[k].size();
      for (double v : logs[k])
        sq += (v - mean) * (v - mean);
      printf("SUMMARY -> %s: Total %f sec; Mean %f sec; Std.Dev. %f (rank %d)\n", names[k], sum, mean,
             sqrt(sq / logs[k].size()), g);
    }
  }
}

int Driver::run()
{
  printf("\tInitializing Probabilities\n");
  const int nMaps = particles.ntot, nAngles = param.nTotGridAngles, nC = param.nTotCTFs;
  const int nShards = (int) shards.size();
  const int K = param.pd.writeAngles;
  const double numconst = 0.5 * log(M_PI) + (1 - param.pd.Ntotpi * 0.5) * (log(2 * M_PI) + 1) + log(param.pd.volu);
  // per shard: the host block start_run / finish_run move -- map entries only for orientation-block shards, map
  // entries + the (tiny) full angle table when the CTF grid is split
  const size_t bytes = splitCTF ? bioem_hip_prob_size(nMaps, nAngles, K) : bioem_hip_prob_size(nMaps, 0, 0);
  for (Shard &sh : shards)
  {
    void *p = bioem_hip_host_alloc(bytes); // == bioem::malloc_device_host (map.cpp:637)
    if (!p)
      fatal("Memory allocation");
    sh.prob = p;
    bioem_hip_prob_map *pm = (bioem_hip_prob_map *) p;
    for (int i = 0; i < nMaps; i++) // bioem.cpp:681-699
    {
      memset(&pm[i], 0, sizeof(pm[i]));
      pm[i].Total = 0.0;
      pm[i].Constoadd = -999999.;
    }
    if (K && splitCTF)
    {
      bioem_hip_prob_angle *pa = (bioem_hip_prob_angle *) (pm + nMaps);
      for (size_t e = 0; e < (size_t) nMaps * nAngles; e++)
      {
        pa[e].forAngles = 0.0;
        pa[e].ConstAngle = -999999.;
      }
    }
  }
  if (debugOutput >= 1)
    printf("\tMain Loop GridAngles %d, CTFs %d, RefMaps %d, Shifts (%d/%d)², Pixels %d², Shards %d%s, merge %s\n", nAngles,
           nC, nMaps, 2 * param.pd.maxDisplaceCenter + param.pd.GridSpaceCenter, param.pd.GridSpaceCenter, param.N,
           nShards, splitCTF ? " (orientation x CTF split)" : "", useRccl ? "RCCL" : "host");
  std::vector<std::thread> th;
  std::vector<std::string> errs(nShards);
  for (int g = 0; g < nShards; g++)
  {
    th.emplace_back([&, g]() {
      Shard &sh = shards[g];
      bioem_hip_handle h = sh.h;
      int rc = bioem_hip_set_phase_timing(h, debugOutput >= 1);
      if (!rc)
        rc = bioem_hip_start_run(h, sh.prob);
      if (!rc && !splitCTF)
        rc = bioem_hip_project_convolve_compare(h, sh.o0, sh.o1);
      // CTF split: the shard's run of (orientation, CTF) pairs, one rectangle per orientation it touches
      for (long long u = sh.u0; !rc && splitCTF && u < sh.u1;)
      {
        const int o = (int) (u / nC), c0 = (int) (u % nC);
        const int c1 = (int) std::min<long long>(nC, c0 + (sh.u1 - u));
        rc = bioem_hip_project_convolve_compare_ctf(h, o, o + 1, c0, c1);
        u += c1 - c0;
      }
      if (!rc)
        rc = bioem_hip_finish_run(h, sh.prob);
      if (rc)
        errs[g] = bioem_hip_last_error(h);
    });
  }
  for (auto &t : th)
    t.join();
  for (int g = 0; g < nShards; g++)
    if (!errs[g].empty())
      fatal("shard %d: %s", g, errs[g].c_str());
  if (debugOutput >= 1)
    print_phase_report();

  // ---- the path's single exchange step: merge of the shards (bioem.cpp:909-1044) ----
  prob.assign(sizeof(bioem_hip_prob_map) * (size_t) nMaps, 0);
  cand.assign((size_t) nMaps * K, bioem_hip_angle_candidate());
  if (useRccl)
  { // one GPU per shard: all-gather over xGMI, fold on the first device
    std::vector<bioem_hip_handle> hs;
    for (Shard &sh : shards)
      hs.push_back(sh.h);
    check(hs[0], bioem_hip_merge(hs.data(), nShards, prob.data(), K, numconst, K ? cand.data() : nullptr), "RCCL merge");
  }
  else if (!splitCTF)
  {
    std::vector<const void *> blocks;
    for (Shard &sh : shards)
      blocks.push_back(sh.prob);
    if (bioem_hip_merge_host(nShards, nMaps, 0, 0, blocks.data(), prob.data()))
      fatal("merge failed");
    if (K)
    {
      std::vector<std::vector<bioem_hip_angle_candidate>> lists(nShards);
      std::vector<const bioem_hip_angle_candidate *> ptrs(nShards);
      for (int g = 0; g < nShards; g++)
      {
        lists[g].resize((size_t) nMaps * K);
        check(shards[g].h, bioem_hip_topk_angles(shards[g].h, K, numconst, lists[g].data()), "top-K orientations");
        ptrs[g] = lists[g].data();
      }
      if (bioem_hip_merge_topk_host(nShards, nMaps, K, ptrs.data(), cand.data()))
        fatal("merge failed");
    }
  }
  else
  { // CTF split: an orientation's angle entry is spread over shards -> log-sum-exp merge of the full tables, then the
    // writer's selection on the host (bioem.cpp:1251-1286)
    std::vector<unsigned char> full(bytes);
    std::vector<const void *> blocks;
    for (Shard &sh : shards)
      blocks.push_back(sh.prob);
    if (bioem_hip_merge_host(nShards, nMaps, nAngles, K, blocks.data(), full.data()))
      fatal("merge failed");
    memcpy(prob.data(), full.data(), prob.size());
    if (K)
    {
      const bioem_hip_prob_angle *pang = (const bioem_hip_prob_angle *) (full.data() + prob.size());
      typedef std::pair<double, int> Item;
      for (int i = 0; i < nMaps; i++)
      {
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> q;
        for (int io = 0; io < nAngles; io++)
        {
          const bioem_hip_prob_angle &pa = pang[(size_t) io * nMaps + i];
          const double logp = log(pa.forAngles) + pa.ConstAngle + numconst;
          if ((int) q.size() < K)
            q.push(Item(logp, io));
          else if (q.top().first < logp)
          {
            q.pop();
            q.push(Item(logp, io));
          }
        }
        const int cnt = (int) q.size();
        for (int r = 0; r < K; r++)
          cand[(size_t) i * K + r].orient = -1;
        for (int r = cnt - 1; r >= 0; r--)
        {
          const bioem_hip_prob_angle &pa = pang[(size_t) q.top().second * nMaps + i];
          bioem_hip_angle_candidate &c = cand[(size_t) i * K + r];
          c.forAngles = pa.forAngles;
          c.ConstAngle = pa.ConstAngle;
          c.logp = q.top().first;
          c.orient = q.top().second;
          q.pop();
        }
      }
    }
  }
  writeOutput();
  return 0;
}

void Driver::cleanup()
{
  for (Shard &sh : shards)
  {
    bioem_hip_host_free(sh.prob);
    if (sh.h)
      bioem_hip_destroy(sh.h);
  }
  shards.clear();
}

// Output_Probabilities and ANG_PROB, text layout of bioem.cpp:1047-1374 (fixed, 4 decimals).
void Driver::writeOutput()
{
  const int nMaps = particles.ntot;
  const bioem_hip_prob_map *pmap = (const bioem_hip_prob_map *) prob.data();
  const bioem_hip_param_device &pd = param.pd;
  const double numconst = 0.5 * log(M_PI) + (1 - pd.Ntotpi * 0.5) * (log(2 * M_PI) + 1) + log(pd.volu);
  const char *bar = "************************* HEADER:: NOTATION *******************************************\n";
  const float *A = param.angles.data();
  const float *K = param.ctfParam.data();

  std::ofstream ang;
  ang.precision(4);
  ang.setf(std::ios::fixed);
  if (pd.writeAngles)
  {
    ang.open("ANG_PROB");
    ang << bar;
    if (!param.doquater)
      ang << " RefMap:  MapNumber ; alpha[rad] - beta[rad] - gamma[rad] - logP - cal log Probability + Constant: "
             "Numerical Const.+ log (volume) + prior ang\n";
    else
      ang << " RefMap:  MapNumber ; q1 - q2 -q3 - logP- cal log Probability + Constant: Numerical Const. + log "
             "(volume) + prior ang\n";
    ang << bar;
  }

  std::ofstream out;
  out.precision(4);
  out.setf(std::ios::fixed);
  out.open(outfileName.c_str());
  out << bar;
  out << "Notation= RefMap:  MapNumber ; LogProb natural logarithm of posterior Probability ; Constant: Numerical "
         "Const. for adding Probabilities \n";
  out << "Notation= RefMap:  MapNumber ; Maximizing Param: MaxLogProb - ";
  if (!param.doquater)
    out << "alpha[rad] - beta[rad] - gamma[rad] - ";
  else
    out << (param.usepsf ? "q1 - q2 - q3 - q4 -" : "q1 - q2 - q3 - q4 - ");
  if (param.usepsf)
    out << (param.doquater ? "PSF amp - PSF phase - PSF envelope" : "PSF amp - PSF phase - PSF envelope");
  else
    out << "CTF amp - CTF defocus - CTF B-Env";
  out << " - center x - center y - normalization - offsett \n";
  if (param.writeCTF)
    out << " RefMap:  MapNumber ; CTFMaxParm: defocus - b-Env (B ref. Penzeck 2010)\n";
  if (param.yespriorAngles)
    out << "**** Remark: Using Prior Proability in Angles ****\n";
  out << bar << "\n";

  for (int i = 0; i < nMaps; i++)
  {
    const bioem_hip_prob_map &pm = pmap[i];
    if (pm.Total > 1.e-38)
    {
      const double lp = log(pm.Total) + pm.Constoadd + 0.5 * log(M_PI) +
                        (1 - pd.Ntotpi * 0.5) * (log(2 * M_PI) + 1) + log(pd.volu);
      out << "RefMap: " << i << " LogProb:  " << lp << " Constant: " << pm.Constoadd << "\n";
      out << "RefMap: " << i << " Maximizing Param: " << lp << " ";
    }
    else
    {
      out << "Warning - RefMap: " << i << "Numerical Integrated Probability without constant = 0.0;\n";
      out << "Warning - RefMap: " << i << "Check that constant is finite: " << pm.Constoadd << "\n";
      out << "Warning - RefMap: i) check model, ii) check refmap , iii) check GPU on/off command inconsitency\n";
    }
    const float *a = A + 4 * (size_t) pm.max_prob_orient;
    const float *k = K + 3 * (size_t) pm.max_prob_conv;
    out << a[0] << " [] " << a[1] << " [] " << a[2] << " [] ";
    if (param.doquater)
      out << a[3] << " [] ";
    out << k[0] << " [] ";
    if (!param.usepsf)
      out << k[1] / 2.f / M_PI / param.elecwavel * 0.0001 << " [micro-m] " << k[2] << " [A²] ";
    else
      out << k[1] << " [1/A²] " << k[2] << " [1/A²] ";
    out << pm.max_prob_cent_x << " [pix] " << pm.max_prob_cent_y << " [pix] " << pm.max_prob_norm << " [] "
        << pm.max_prob_mu << " [] \n";
    if (param.writeCTF && param.usepsf)
    {
      const float denomi = k[1] * k[1] + k[2] * k[2];
      out << "RefMap: " << i << " CTFMaxParam: " << 2 * M_PI * k[1] / denomi / param.elecwavel * 0.0001
          << " [micro-m] " << 4 * M_PI * M_PI * k[2] / denomi << " [A²] \n";
    }
    if (pd.writeAngles)
    {
      // the K best orientations (selected by the reference's min-heap rule, bioem.cpp:1251-1286 -- on the device for
      // orientation-block shards, see Driver::run), best first
      const int Kbest = pd.writeAngles;
      for (int r = 0; r < Kbest; r++)
      {
        const bioem_hip_angle_candidate &c = cand[(size_t) i * Kbest + r];
        if (c.orient < 0)
          break;
        const int io = c.orient;
        double logp = c.logp;
        if (param.yespriorAngles)
          logp += param.angprior[io];
        const float *q4 = A + 4 * (size_t) io;
        ang << " " << i << " " << q4[0] << " " << q4[1] << " " << q4[2] << " ";
        if (param.doquater)
          ang << q4[3] << " ";
        ang << logp << " Separated: " << log(c.forAngles) << " " << c.ConstAngle << " " << numconst;
        if (param.yespriorAngles)
          ang << " " << param.angprior[io];
        ang << "\n";
      }
    }
  }
  if (pd.writeAngles)
    ang.close();
  out.close();
}

} // namespace bioem_host
