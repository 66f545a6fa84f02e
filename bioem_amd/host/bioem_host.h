// bioem_host.h -- C++ host layer of the MI355X BioEM engine: the parts of the reference's driver that
// sit on either side of the device plugin for the compare path (parameter file, orientation lists/grids,
// CTF/PSF kernels, model and particle readers, the run() loop, the Output_Probabilities writer), shaped
// after the reference's classes so that a reference user finds the same hooks:
//
//   InputParams      <-> bioem_param         (/root/reference/include/param.h:49-153)
//   Model            <-> bioem_model         (/root/reference/include/model.h:20-57)
//   ParticleStack    <-> bioem_RefMap        (/root/reference/include/map.h:26-88)
//   Driver           <-> bioem / bioem_cuda  (/root/reference/include/bioem.h:24-99)
//
// All heavy lifting is delegated to libbioem_hip.so through include/bioem_hip.h.  Errors follow the
// reference's behaviour: print "Error - ..." and exit(1) (defs.h:18-26).
#ifndef BIOEM_HOST_H
#define BIOEM_HOST_H

#include <string>
#include <vector>

#include "bioem_hip.h"

namespace bioem_host
{

[[noreturn]] void fatal(const char *fmt, ...);
void warn(const char *fmt, ...);

struct InputParams
{
  // parameter-file content (param.cpp:64-627)
  float pixelSize = 0;
  int N = 0;
  int angleGridPointsAlpha = 0, angleGridPointsBeta = 0, GridPointsQuatern = -1;
  bool doquater = false, usepsf = false, writeCTF = false, nocentermass = false, notnormmap = false;
  bool yespriorAngles = false, ignorePDB = false, printrotmod = false, doaaradius = true;
  float elecwavel = 0.019866f;
  float priorMod = 1.f;
  int shiftX = 0, shiftY = 0;
  float startBfactor = 0, endBfactor = 0, startDefocus = 0, endDefocus = 0;
  float startGridEnvelop = 0, endGridEnvelop = 0, startGridCTF_phase = 0, endGridCTF_phase = 0;
  float startGridCTF_amp = 0, endGridCTF_amp = 0;
  int numberGridPointsEnvelop = 0, numberGridPointsCTF_phase = 0, numberGridPointsCTF_amp = 0;
  float gridEnvelop = 0, gridCTF_phase = 0, gridCTF_amp = 0;
  bool notuniformangles = false; // set by --ReadOrientation (bioem.cpp:162)
  bioem_hip_param_device pd{};

  // derived
  std::vector<float> angles; // [n][4] {pos0,pos1,pos2,quat4}
  std::vector<float> angprior;
  int nTotGridAngles = 0;
  float voluang = 0;
  int nTotCTFs = 0;
  std::vector<float> refCTF;   // [nCTF][N][H][2]
  std::vector<float> ctfParam; // [nCTF][3]

  void readParameters(const char *file);          // param.cpp:64-627
  void calculateGridsParam(const char *anglefile); // param.cpp:988-1334
  void calculateRefCTF();                          // param.cpp:1336-1620 (PSF r2c runs through `r2c`)
  // r2c hook used only in PSF mode; supplied by the driver (device transform)
  void (*r2c)(void *ctx, int N, const float *in, float *out) = nullptr;
  void *r2c_ctx = nullptr;
};

// the pure functions (also exported through the C ABI in capi.cpp for tests and bench.py)
int ctf_kernels(int N, float pixelSize, bool usepsf, float startAmp, float endAmp, int nAmp, float startPhase,
                float endPhase, int nPhase, float startEnv, float endEnv, int nEnv, float *refCTF, float *ctfParam,
                float *steps, void (*r2c)(void *, int, const float *, float *), void *ctx);
float volume_element(float voluang, int gridSpaceCenter, int maxDisplaceCenter, float pixelSize, int nAmp,
                     float gridEnvelop, float gridPhase, float sigB, float sigDef, float sigAmp);

// MRC header fields used by the readers (include/mrc.h of the reference): endianness guessed from header range
// violations, data start at 1024 + nsymbt bytes
struct MrcHeader
{
  int nc, nr, ns, mode, nsymbt, swap;
};
MrcHeader mrc_read_header(const char *file);
unsigned int mrc_bswap32(unsigned int v);

struct Model
{
  std::vector<bioem_hip_model_point> points;
  float NormDen = 0;
  bool readPDB = false, readModelMRC = false;
  void readModel(const InputParams &p, const char *file); // model.cpp:674-710
  void readTextFile(const InputParams &p, const char *file);
  void readPDBFile(const char *file);
  void readMRCFile(const InputParams &p, const char *file); // model.cpp:332-416
  void centerDensityMass();
};

struct ParticleStack
{
  int ntot = 0, N = 0;
  std::vector<float> maps; // [ntot][N][N]
  bool readMRC = false, readMultMRC = false;
  void readRefMaps(const InputParams &p, const char *file); // map.cpp:520-555
  void readTextMaps(const char *file);
  void readMRCMaps(const InputParams &p, const char *file);
};

class Driver
{
public:
  Driver();
  ~Driver();
  int configure(int argc, char **argv); // bioem.cpp:438-585
  int run();                            // bioem.cpp:659-1377
  void print_phase_report();            // BIOEM_DEBUG_OUTPUT >= 1: timer.cpp:156-165, bioem.cpp:769-889
  void cleanup();

  InputParams param;
  Model model;
  ParticleStack particles;
  std::string outfileName = "Output_Probabilities";
  std::vector<unsigned char> prob;              // merged map entries [nMaps]
  std::vector<bioem_hip_angle_candidate> cand;  // merged K best orientations [nMaps][K] (WRITE_PROB_ANGLES)

  int algo = 1;
  int debugOutput = 0;
  int nGpus = 1;     // number of shards
  int firstDev = 0;  // GPUDEVICE
  bool splitCTF = false; // fewer orientations than shards: (orientation, CTF) pairs are split instead
  bool useRccl = false;  // one GPU per shard: the merge runs over RCCL (bioem_hip_merge)

private:
  // one unit of the run: orientations [o0, o1) (and, with the CTF split, the run [u0, u1) of orientation-major
  // (orientation, CTF) pairs) on one device, with a private probability block
  struct Shard
  {
    bioem_hip_handle h = nullptr;
    int device = 0, o0 = 0, o1 = 0;
    long long u0 = 0, u1 = 0;
    void *prob = nullptr;
  };
  int readOptions(int argc, char **argv);
  void writeOutput();
  std::vector<Shard> shards;
};

} // namespace bioem_host
#endif
