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
  (void) ctx;
  if (bioem_hip_r2c(0, N, 1, in, out))
    fatal("device r2c failed");
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
  param.r2c = r2c_on_device;
  param.calculateRefCTF();
  if (getenv("BIOEM_DEBUG_BREAK")) // bioem.cpp:518-525: after volu was computed with the full counts
  {
    const int cut = atoi(getenv("BIOEM_DEBUG_BREAK"));
    if (param.nTotGridAngles > cut)
      param.nTotGridAngles = cut;
    if (param.nTotCTFs > cut)
      param.nTotCTFs = cut;
  }
  nGpus = ndev;
  if (getenv("BIOEM_GPUS"))
    nGpus = std::max(1, std::min(ndev, atoi(getenv("BIOEM_GPUS"))));
  // BIOEM_SHARDS=n: n orientation shards dealt round-robin over the selected GPUs (default: one per GPU); more
  // shards than GPUs is only useful to rehearse the multi-GPU control flow and merge on a smaller machine
  int nDevUsed = nGpus;
  if (getenv("BIOEM_SHARDS"))
    nGpus = std::max(1, atoi(getenv("BIOEM_SHARDS")));
  if (nGpus > param.nTotGridAngles)
    nGpus = param.nTotGridAngles;
  nDevUsed = std::min(nDevUsed, nGpus);
  int firstDev = 0;
  if (getenv("GPUDEVICE") && atoi(getenv("GPUDEVICE")) >= 0) // bioem_cuda.cu:719-732
  {
    firstDev = atoi(getenv("GPUDEVICE"));
    if (!getenv("BIOEM_SHARDS"))
      nGpus = 1;
    nDevUsed = 1;
    if (firstDev >= ndev)
      fatal("GPUDEVICE %d out of range (%d devices)", firstDev, ndev);
  }
  const int nMaps = particles.ntot;
  handles.assign(nGpus, nullptr);
  for (int g = 0; g < nGpus; g++)
  {
    bioem_hip_handle h = nullptr;
    const int rc = bioem_hip_create(&h, firstDev + g % nDevUsed, &param.pd, nMaps, param.nTotGridAngles,
                                    param.nTotCTFs, algo);
    handles[g] = h;
    check(h, rc, "bioem_hip_create");
    check(h, bioem_hip_upload_particle_maps(h, particles.maps.data()), "upload particles");
    check(h, bioem_hip_upload_ctf(h, param.refCTF.data(), param.ctfParam.data()), "upload CTF");
    check(h, bioem_hip_upload_model(h, model.points.data(), (int) model.points.size(), model.NormDen, param.pixelSize,
                                    param.shiftX, param.shiftY),
          "upload model");
    check(h, bioem_hip_upload_orientations(h, param.angles.data(), param.nTotGridAngles, param.doquater ? 1 : 0),
          "upload orientations");
  }
  return 0;
}

int Driver::run()
{
  printf("\tInitializing Probabilities\n");
  const int nMaps = particles.ntot, nAngles = param.nTotGridAngles;
  const size_t bytes = bioem_hip_prob_size(nMaps, nAngles, param.pd.writeAngles);
  shardProb.assign(nGpus, nullptr);
  for (int g = 0; g < nGpus; g++)
  {
    void *p = bioem_hip_host_alloc(bytes); // == bioem::malloc_device_host (map.cpp:637)
    if (!p)
      fatal("Memory allocation");
    shardProb[g] = p;
    bioem_hip_prob_map *pm = (bioem_hip_prob_map *) p;
    for (int i = 0; i < nMaps; i++) // bioem.cpp:681-699
    {
      memset(&pm[i], 0, sizeof(pm[i]));
      pm[i].Total = 0.0;
      pm[i].Constoadd = -999999.;
    }
    if (param.pd.writeAngles)
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
    printf("\tMain Loop GridAngles %d, CTFs %d, RefMaps %d, Shifts (%d/%d)², Pixels %d², GPUs %d\n", nAngles,
           param.nTotCTFs, nMaps, 2 * param.pd.maxDisplaceCenter + param.pd.GridSpaceCenter, param.pd.GridSpaceCenter,
           param.N, nGpus);
  std::vector<std::thread> th;
  std::vector<std::string> errs(nGpus);
  for (int g = 0; g < nGpus; g++)
  {
    th.emplace_back([&, g]() {
      bioem_hip_handle h = handles[g];
      // same contiguous blocks as `mpirun -n nGpus` (bioem.cpp:748-753)
      const int o0 = (int) ((long long) g * nAngles / nGpus);
      const int o1 = (int) ((long long) (g + 1) * nAngles / nGpus);
      if (bioem_hip_start_run(h, shardProb[g]) || bioem_hip_project_convolve_compare(h, o0, o1) ||
          bioem_hip_finish_run(h, shardProb[g]))
        errs[g] = bioem_hip_last_error(h);
    });
  }
  for (auto &t : th)
    t.join();
  for (int g = 0; g < nGpus; g++)
    if (!errs[g].empty())
      fatal("device %d: %s", g, errs[g].c_str());
  prob.assign(bytes, 0);
  if (nGpus == 1)
    memcpy(prob.data(), shardProb[0], bytes);
  else if (bioem_hip_merge_host(nGpus, nMaps, nAngles, param.pd.writeAngles, (const void *const *) shardProb.data(),
                                prob.data()))
    fatal("merge failed");
  writeOutput();
  return 0;
}

void Driver::cleanup()
{
  for (void *p : shardProb)
    bioem_hip_host_free(p);
  shardProb.clear();
  for (bioem_hip_handle h : handles)
    if (h)
      bioem_hip_destroy(h);
  handles.clear();
}

// Output_Probabilities and ANG_PROB, text layout of bioem.cpp:1047-1374 (fixed, 4 decimals).
void Driver::writeOutput()
{
  const int nMaps = particles.ntot, nAngles = param.nTotGridAngles;
  const bioem_hip_prob_map *pmap = (const bioem_hip_prob_map *) prob.data();
  const bioem_hip_prob_angle *pang = (const bioem_hip_prob_angle *) (pmap + nMaps);
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
      // K best orientations through a min-heap on (logp, orientation), best first (bioem.cpp:1251-1286)
      typedef std::pair<double, int> Item;
      std::priority_queue<Item, std::vector<Item>, std::greater<Item>> q;
      const unsigned Kbest = (unsigned) pd.writeAngles;
      for (int io = 0; io < nAngles; io++)
      {
        const bioem_hip_prob_angle &pa = pang[(size_t) io * nMaps + i];
        const double logp = log(pa.forAngles) + pa.ConstAngle + numconst;
        if (q.size() < Kbest)
          q.push(Item(logp, io));
        else if (q.top().first < logp)
        {
          q.pop();
          q.push(Item(logp, io));
        }
      }
      std::vector<Item> best(q.size());
      for (int r = (int) q.size() - 1; r >= 0; r--)
      {
        best[r] = q.top();
        q.pop();
      }
      for (const Item &it : best)
      {
        const int io = it.second;
        const bioem_hip_prob_angle &pa = pang[(size_t) io * nMaps + i];
        double logp = it.first;
        if (param.yespriorAngles)
          logp += param.angprior[io];
        const float *q4 = A + 4 * (size_t) io;
        ang << " " << i << " " << q4[0] << " " << q4[1] << " " << q4[2] << " ";
        if (param.doquater)
          ang << q4[3] << " ";
        ang << logp << " Separated: " << log(pa.forAngles) << " " << pa.ConstAngle << " " << numconst;
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
