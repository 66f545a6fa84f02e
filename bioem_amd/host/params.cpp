// params.cpp -- parameter file, orientation sets, CTF/PSF kernels and the grid volume element.
// Behaviour follows /root/reference/param.cpp (readParameters 64-627, CalculateGridsParam 988-1334,
// CalculateRefCTF 1336-1620) including its documented quirks; the code is
// written from that specification, not transcribed.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "bioem_host.h"

namespace bioem_host
{

void fatal(const char *fmt, ...)
{
  // same shape as the reference's myError (defs.h:18-26): banner, "Error - <text>", exit(1)
  printf("!!!!!!!!!!!!!!!!!!!!!!!!\nError - ");
  va_list ap;
  va_start(ap, fmt);
  vprintf(fmt, ap);
  va_end(ap);
  printf("\n!!!!!!!!!!!!!!!!!!!!!!!!\n");
  fflush(stdout);
  exit(1);
}

void warn(const char *fmt, ...)
{
  printf("Warning - ");
  va_list ap;
  va_start(ap, fmt);
  vprintf(fmt, ap);
  va_end(ap);
  printf("\n");
}

namespace
{
// the reference tokenises with strtok(line, " ") and converts with atof/atoi (param.cpp:124-133)
std::vector<std::string> split_spaces(const std::string &line)
{
  std::vector<std::string> t;
  size_t i = 0;
  while (i < line.size())
  {
    while (i < line.size() && line[i] == ' ')
      i++;
    size_t j = i;
    while (j < line.size() && line[j] != ' ')
      j++;
    if (j > i)
      t.push_back(line.substr(i, j - i));
    i = j;
  }
  return t;
}

float tokf(const std::vector<std::string> &t, size_t i, const char *key)
{
  if (i >= t.size())
    fatal("Missing value for keyword %s", key);
  return (float) atof(t[i].c_str());
}

int toki(const std::vector<std::string> &t, size_t i, const char *key)
{
  if (i >= t.size())
    fatal("Missing value for keyword %s", key);
  return atoi(t[i].c_str());
}

struct Range3
{
  float a, b;
  int n;
};

Range3 read_range(const std::vector<std::string> &t, const char *key, const char *what)
{
  Range3 r;
  r.a = tokf(t, 1, key);
  if (r.a < 0)
    fatal("Negative start %s", what);
  r.b = tokf(t, 2, key);
  if (r.b < 0)
    fatal("Negative end %s", what);
  r.n = toki(t, 3, key);
  if (r.n < 0)
    fatal("Negative number of grid points %s", what);
  if (r.a > r.b)
    fatal("Grid ill defined end > start");
  return r;
}
} // namespace

void InputParams::readParameters(const char *file)
{
  bool yesPixSi = false, yesNumPix = false, yesGPal = false, yesGPbe = false, yesMDC = false, yesBFact = false,
       yesDefocus = false, yesAMP = false, yesPSFenv = false, yesPSFpha = false, yesquatgrid = false;
  // defaults, param.cpp:84-106
  pd.tousepsf = 0;
  pd.sigmaPriorbctf = 100.f;
  pd.sigmaPriordefo = 2.0f;
  pd.Priordefcent = 3.0f;
  pd.sigmaPrioramp = 0.5f;
  pd.Priorampcent = 0.f;

  std::ifstream input(file);
  if (!input.good())
    fatal("Opening file: %s", file);
  std::cout << "\n +++++++++++++++++++++++++++++++++++++++++ \n";
  std::cout << "\n   READING BioEM PARAMETERS             \n\n";
  std::cout << " +++++++++++++++++++++++++++++++++++++++++ \n";
  std::string line;
  while (std::getline(input, line))
  {
    if (!line.empty() && line.back() == '\r')
      line.pop_back();
    if (line.empty() || line[0] == '#')
      continue;
    const std::vector<std::string> t = split_spaces(line);
    if (t.empty())
      continue;
    const std::string &k = t[0];
    if (k == "PIXEL_SIZE")
    {
      pixelSize = tokf(t, 1, "PIXEL_SIZE");
      if (pixelSize < 0)
        fatal("Negative pixel size");
      std::cout << "Pixel Size " << pixelSize << "\n";
      yesPixSi = true;
    }
    else if (k == "NUMBER_PIXELS")
    {
      N = toki(t, 1, "NUMBER_PIXELS");
      if (N < 0)
        fatal("Negative Number of Pixels");
      std::cout << "Number of Pixels " << N << "\n";
      yesNumPix = true;
    }
    else if (k == "GRIDPOINTS_ALPHA")
    {
      angleGridPointsAlpha = toki(t, 1, "GRIDPOINTS_ALPHA");
      if (angleGridPointsAlpha < 0)
        fatal("Negative GRIDPOINTS_ALPHA");
      std::cout << "Grid points alpha " << angleGridPointsAlpha << "\n";
      yesGPal = true;
    }
    else if (k == "GRIDPOINTS_BETA")
    {
      angleGridPointsBeta = toki(t, 1, "GRIDPOINTS_BETA");
      if (angleGridPointsBeta < 0)
        fatal("Negative GRIDPOINTS_BETA");
      std::cout << "Grid points in Cosine ( beta ) " << angleGridPointsBeta << "\n";
      yesGPbe = true;
    }
    else if (k == "USE_QUATERNIONS")
    {
      std::cout << "Orientations with Quaternions. \n";
      doquater = true;
    }
    else if (k == "GRIDPOINTS_QUATERNION")
    {
      if (notuniformangles)
        fatal("Inconsistent input: grid or list with quaternions?");
      GridPointsQuatern = toki(t, 1, "GRIDPOINTS_QUATERNION");
      std::cout << "Gridpoints Quaternions " << GridPointsQuatern << "\n";
      yesquatgrid = true;
      doquater = true;
    }
    else if (k == "CTF_B_ENV")
    {
      const Range3 r = read_range(t, "CTF_B_ENV", "B Env.");
      startBfactor = r.a;
      endBfactor = r.b;
      numberGridPointsEnvelop = r.n;
      std::cout << "Grid CTF B-ENV: " << r.a << " " << r.b << " " << r.n << "\n";
      yesBFact = true;
    }
    else if (k == "CTF_DEFOCUS")
    {
      const Range3 r = read_range(t, "CTF_DEFOCUS", "defocus");
      startDefocus = r.a;
      endDefocus = r.b;
      numberGridPointsCTF_phase = r.n;
      std::cout << "Grid CTF Defocus: " << r.a << " " << r.b << " " << r.n << "\n";
      if (endDefocus > 8.)
        fatal("Defocus beyond 8micro-m range is not allowed");
      yesDefocus = true;
    }
    else if (k == "CTF_AMPLITUDE" || k == "PSF_AMPLITUDE")
    {
      const Range3 r = read_range(t, k.c_str(), "amplitude");
      startGridCTF_amp = r.a;
      endGridCTF_amp = r.b;
      numberGridPointsCTF_amp = r.n;
      std::cout << "Grid Amplitude: " << r.a << " " << r.b << " " << r.n << "\n";
      yesAMP = true;
    }
    else if (k == "ELECTRON_WAVELENGTH")
    {
      elecwavel = tokf(t, 1, "ELECTRON_WAVELENGTH");
      if (elecwavel < 0.0150)
        fatal("Wrong electron wave length %lf. Has to be in Angstrom (A)", (double) elecwavel);
      std::cout << "Electron wave length in (A) is: " << elecwavel << "\n";
    }
    else if (k == "USE_PSF")
    {
      usepsf = true;
      pd.tousepsf = 1;
      std::cout << "Important: Using Point Spread Function. Thus, all parameters are in Real Space. \n";
    }
    else if (k == "PSF_ENVELOPE")
    {
      const Range3 r = read_range(t, "PSF_ENVELOPE", "PSF Env.");
      startGridEnvelop = r.a;
      endGridEnvelop = r.b;
      numberGridPointsEnvelop = r.n;
      std::cout << "Grid PSF Envelope: " << r.a << " " << r.b << " " << r.n << "\n";
      yesPSFenv = true;
    }
    else if (k == "PSF_PHASE")
    {
      const Range3 r = read_range(t, "PSF_PHASE", "PSF phase");
      startGridCTF_phase = r.a;
      endGridCTF_phase = r.b;
      numberGridPointsCTF_phase = r.n;
      std::cout << "Grid PSF phase: " << r.a << " " << r.b << " " << r.n << "\n";
      yesPSFpha = true;
    }
    else if (k == "DISPLACE_CENTER")
    {
      pd.maxDisplaceCenter = toki(t, 1, "DISPLACE_CENTER");
      if (pd.maxDisplaceCenter < 0)
        fatal("Negative MAX_D_CENTER");
      std::cout << "Maximum displacement Center " << pd.maxDisplaceCenter << "\n";
      pd.GridSpaceCenter = toki(t, 2, "DISPLACE_CENTER");
      if (pd.GridSpaceCenter < 0)
        fatal("Negative PIXEL_GRID_CENTER");
      std::cout << "Grid space displacement center " << pd.GridSpaceCenter << "\n";
      yesMDC = true;
    }
    else if (k == "WRITE_PROB_ANGLES")
    {
      pd.writeAngles = toki(t, 1, "WRITE_PROB_ANGLES");
      if (pd.writeAngles < 0)
        fatal("Negative WRITE_PROB_ANGLES");
      std::cout << "Writing " << pd.writeAngles << " Probabilies of each angle \n";
    }
    else if (k == "IGNORE_PDB")
    {
      ignorePDB = true;
      std::cout << "Ignoring PDB extension in model file \n";
    }
    else if (k == "NO_PROJECT_RADIUS")
    {
      doaaradius = false; // parsed, never used by the reference either
      std::cout << "Not Projecting corresponding radius \n";
    }
    else if (k == "WRITE_CTF_PARAM")
    {
      writeCTF = true;
      std::cout << "Writing CTF parameters from PSF parameters that maximize the posterior. \n";
    }
    else if (k == "NO_CENTEROFMASS")
    {
      nocentermass = true;
      std::cout << "BE CAREFUL CENTER OF MASS IS NOT REMOVED \n Calculated images might be out of range \n";
    }
    else if (k == "PRINT_ROTATED_MODELS")
    {
      printrotmod = true;
      std::cout << "PRINTING out rotatted models (best for debugging)\n";
    }
    else if (k == "NO_MAP_NORM")
    {
      notnormmap = true;
      std::cout << "NOT NORMALIZING MAP\n";
    }
    else if (k == "PRIOR_MODEL")
    {
      priorMod = tokf(t, 1, "PRIOR_MODEL");
      std::cout << "MODEL PRIOR Probability " << priorMod << "\n";
    }
    else if (k == "PRIOR_ANGLES")
    {
      yespriorAngles = true;
      std::cout << "READING Priors for Orientations in additonal orientation file\n";
    }
    else if (k == "SHIFT_X")
    {
      shiftX = toki(t, 1, "SHIFT_X");
      std::cout << "Shifting initial model X by " << shiftX << "\n";
    }
    else if (k == "SHIFT_Y")
    {
      shiftY = toki(t, 1, "SHIFT_Y");
      std::cout << "Shifting initial model Y by " << shiftY << "\n";
    }
    else if (k == "SIGMA_PRIOR_B_CTF")
    {
      pd.sigmaPriorbctf = tokf(t, 1, "SIGMA_PRIOR_B_CTF");
      std::cout << "Chainging  Gaussian width in Prior of Envelope b parameter: " << pd.sigmaPriorbctf << "\n";
    }
    else if (k == "SIGMA_PRIOR_DEFOCUS")
    {
      pd.sigmaPriordefo = tokf(t, 1, "SIGMA_PRIOR_DEFOCUS");
      std::cout << "Gaussian Width in Prior of defocus parameter: " << pd.sigmaPriordefo << "\n";
    }
    else if (k == "PRIOR_DEFOCUS_CENTER")
    {
      pd.Priordefcent = tokf(t, 1, "PRIOR_DEFOCUS_CENTER");
      std::cout << "Gaussian Center in Prior of defocus parameter: " << pd.Priordefcent << "\n";
    }
    else if (k == "SIGMA_PRIOR_AMP_CTF")
    {
      pd.sigmaPrioramp = tokf(t, 1, "SIGMA_PRIOR_AMP_CTF");
      std::cout << "Gaussian Width in Prior of amplitude parameter: " << pd.sigmaPrioramp << "\n";
    }
    else if (k == "PRIOR_AMP_CTF_CENTER")
    {
      pd.Priorampcent = tokf(t, 1, "PRIOR_AMP_CTF_CENTER");
      std::cout << "Gaussian Center in Prior of amplitude parameter: " << pd.Priorampcent << "\n";
    }
    // unknown keywords are ignored, as in the reference
  }
  input.close();

  // mandatory-input checks, param.cpp:532-599
  if (!yesPixSi)
    fatal("Input missing: please provide PIXEL_SIZE");
  if (!yesNumPix)
    fatal("Input missing: please provide NUMBER_PIXELS");
  if (!notuniformangles)
  {
    if (!doquater)
    {
      if (!yesGPal)
        fatal("Input missing: please provide GRIDPOINTS_ALPHA");
      if (!yesGPbe)
        fatal("Input missing: please provide GRIDPOINTS_BETA");
    }
    else if (!yesquatgrid)
      fatal("Input missing: please provide GRIDPOINTS_QUATERNION");
  }
  if (!yesMDC)
    fatal("Input missing: please provide grid displacement CENTER");
  std::cout << "To verify input of Priors:\n";
  std::cout << "Sigma Prior B-Env: " << pd.sigmaPriorbctf << "\n";
  std::cout << "Sigma Prior Defocus: " << pd.sigmaPriordefo << "\n";
  std::cout << "Center Prior Defocus: " << pd.Priordefcent << "\n";
  if (usepsf)
  {
    if (!yesPSFpha)
      fatal("Input missing: please provide grid PSF PHASE");
    if (!yesPSFenv)
      fatal("Input missing: please provide grid PSF ENVELOPE");
    if (!yesAMP)
      fatal("Input missing: please provide grid PSF AMPLITUD");
  }
  else
  {
    if (!yesBFact)
      fatal("Input missing: please provide grid CTF B Env.");
    if (!yesDefocus)
      fatal("Input missing: please provide grid CTF defocus");
    if (!yesAMP)
      fatal("Input missing: please provide grid CTF amplitude");
    // defocus [micro-m] -> phase, param.cpp:601-607 (double product stored to float)
    startGridCTF_phase = (float) (startDefocus * M_PI * 2.f * 10000 * elecwavel);
    endGridCTF_phase = (float) (endDefocus * M_PI * 2.f * 10000 * elecwavel);
    startGridEnvelop = startBfactor;
    endGridEnvelop = endBfactor;
    pd.Priordefcent = (float) (pd.Priordefcent * (M_PI * 2.f * 10000 * elecwavel));
    pd.sigmaPriordefo = (float) (pd.sigmaPriordefo * (M_PI * 2.f * 10000 * elecwavel));
  }
  pd.NumberPixels = N;
  pd.NumberFFTPixels1D = N / 2 + 1;
  if (writeCTF && !usepsf)
    fatal("Writing CTF is only valid when integrating over the PSF");
  std::cout << " +++++++++++++++++++++++++++++++++++++++++ \n";
}

namespace
{
// fixed-width (12 character) column of an orientation list line, param.cpp:1089-1096,1254-1264
float column12(const std::string &line, int col)
{
  const size_t off = (size_t) col * 12;
  if (off >= line.size())
    fatal("line parsed by sscanf has wrong argument");
  const std::string s = line.substr(off, 12);
  float v;
  if (sscanf(s.c_str(), "%f", &v) != 1)
    fatal("line parsed by sscanf has wrong argument");
  return v;
}
} // namespace

void InputParams::calculateGridsParam(const char *anglefile)
{
  angles.clear();
  angprior.clear();
  if (!doquater)
  {
    std::cout << "Analysis Using Default Euler Angles\n";
    if (!notuniformangles)
    {
      if (yespriorAngles)
        fatal("This option is not valid with prior for orientations."
              "Please provide separate file with orientations and priors");
      std::cout << "Calculating Grids in Euler Angles\n";
      // param.cpp:1015-1047
      const float grid_alpha = (float) (2.f * M_PI / (float) angleGridPointsAlpha);
      const float cos_grid_beta = 2.f / (float) angleGridPointsBeta;
      for (int ia = 0; ia < angleGridPointsAlpha; ia++)
        for (int ib = 0; ib < angleGridPointsBeta; ib++)
          for (int ig = 0; ig < angleGridPointsAlpha; ig++)
          {
            angles.push_back((float) ((float) ia * grid_alpha - M_PI + grid_alpha * 0.5f));
            angles.push_back(acosf((float) ib * cos_grid_beta - 1 + cos_grid_beta * 0.5f));
            angles.push_back((float) ((float) ig * grid_alpha - M_PI + grid_alpha * 0.5f));
            angles.push_back(0.f);
          }
      nTotGridAngles = (int) (angles.size() / 4);
      voluang = (float) (grid_alpha * grid_alpha * cos_grid_beta / (2.f * M_PI) / (2.f * M_PI) / 2.f * priorMod);
    }
    else
    {
      std::ifstream input(anglefile);
      if (!input.good())
        fatal("Euler angle file failed to open file %s", anglefile);
      std::string line;
      std::getline(input, line);
      int n = 0;
      if (sscanf(line.substr(0, 12).c_str(), "%d", &n) != 1)
        fatal("line parsed by sscanf has wrong argument");
      std::cout << "Number of Euler angles " << n << "\n";
      if (n < 1)
        fatal("Euler angles not defined in input file");
      int cnt = 0;
      while (std::getline(input, line))
      {
        angles.push_back(column12(line, 0));
        angles.push_back(column12(line, 1));
        angles.push_back(column12(line, 2));
        angles.push_back(0.f);
        if (yespriorAngles)
        {
          const float pp = column12(line, 3);
          if (pp < 0.0000001)
            std::cout << "Sure your input is correct? Very small prior.\n";
          angprior.push_back(pp);
        }
        cnt++;
        if (n < cnt)
          fatal("Not properly defined total Euler angles %d instead of %d", cnt, n);
      }
      if (n > cnt)
        fatal("Less quaternions than expected in header %d instead of %d", cnt, n);
      nTotGridAngles = n;
      voluang = (float) (1. / (float) n * priorMod);
    }
  }
  else
  {
    if (!notuniformangles)
    {
      std::cout << "Calculating Grids in Quaterions\n ";
      if (yespriorAngles)
        fatal("This option is not valid with prior for orientations. "
              "It is necessary to provide a separate file with the angles and the priors");
      if (GridPointsQuatern < 0)
        fatal("Missing gridpoints quaternions. After QUATERNIONS (int). (int)=Number of gridpoins per dimension");
      // param.cpp:1159-1209: grid points of the unit ball, each with +w and -w
      const float dgridq = 2.f / (float) (GridPointsQuatern + 1);
      for (int ia = 0; ia < GridPointsQuatern + 1; ia++)
      {
        const float q1 = (float) ((float) ia * dgridq - 1.f + 0.5 * dgridq);
        for (int ib = 0; ib < GridPointsQuatern + 1; ib++)
        {
          const float q2 = (float) ((float) ib * dgridq - 1.f + 0.5 * dgridq);
          for (int ig = 0; ig < GridPointsQuatern + 1; ig++)
          {
            const float q3 = (float) ((float) ig * dgridq - 1.f + 0.5 * dgridq);
            if (q1 * q1 + q2 * q2 + q3 * q3 <= 1.f)
            {
              const float w = sqrtf(1.f - q1 * q1 - q2 * q2 - q3 * q3);
              const float pos[2] = {w, -w};
              for (int s = 0; s < 2; s++)
              {
                angles.push_back(q1);
                angles.push_back(q2);
                angles.push_back(q3);
                angles.push_back(pos[s]);
              }
            }
          }
        }
      }
      nTotGridAngles = (int) (angles.size() / 4);
      voluang = dgridq * dgridq * dgridq * priorMod;
    }
    else
    {
      std::ifstream input(anglefile);
      if (!input.good())
        fatal("Quaterion list file %s", anglefile);
      std::string line;
      std::getline(input, line);
      int n = 0;
      if (sscanf(line.substr(0, 12).c_str(), "%d", &n) != 1)
        fatal("line parsed by sscanf has wrong argument");
      if (n < 1)
        fatal("Invalid number of quaternions %d", n);
      std::cout << "Number of quaternions " << n << "\n";
      int cnt = 0;
      while (std::getline(input, line))
      {
        float q[4];
        for (int c = 0; c < 4; c++)
        {
          q[c] = column12(line, c);
          if (q[c] < -1 || q[c] > 1)
            fatal("Reading quaterions from list. Value out of range %lf row %d", (double) q[c], cnt);
          angles.push_back(q[c]);
        }
        if (yespriorAngles)
        {
          const float pp = column12(line, 4);
          if (pp < 0.0000001)
            std::cout << "Sure your input is correct? Very small prior.\n";
          angprior.push_back(pp);
        }
        cnt++;
        if (n < cnt)
          fatal("More quaternions than expected in header %d instead of %d", cnt, n);
      }
      if (n > cnt)
        fatal("Less quaternions than expected in header %d instead of %d", cnt, n);
      nTotGridAngles = n;
      voluang = (float) (1. / (float) n * priorMod);
    }
    std::cout << "Analysis with Quaternions. Total number of quaternions " << nTotGridAngles << "\n";
  }
}

// CTF (Fourier space) or PSF (real space + r2c) kernels on the amp x phase x env grid.
// Quirks kept on purpose (they shift log P systematically):
//  * a one-point grid uses step := start;
//  * in CTF mode frequency index i is written to rows i AND N-1-i, for i < N/2+1.
int ctf_kernels(int N, float pixelSize, bool usepsf, float startAmp, float endAmp, int nAmp, float startPhase,
                float endPhase, int nPhase, float startEnv, float endEnv, int nEnv, float *refCTF, float *ctfParam,
                float *steps, void (*r2c)(void *, int, const float *, float *), void *ctx)
{
  const int H = N / 2 + 1;
  const size_t M = (size_t) N * H;
  float gAmp = (endAmp - startAmp) / (float) nAmp;
  float gPhase = (endPhase - startPhase) / (float) nPhase;
  float gEnv = (endEnv - startEnv) / (float) nEnv;
  if (nAmp == 1)
    gAmp = startAmp;
  else if ((endAmp - startAmp) < 0.)
    fatal("Interval of amplitude in CTF/PSF negative");
  if (nPhase == 1)
    gPhase = startPhase;
  else if ((endPhase - startPhase) < 0.)
    fatal("Interval of phase in CTF/PSF is negative");
  if (nEnv == 1)
    gEnv = startEnv;
  else if ((endEnv - startEnv) < 0.)
    fatal("Interval of envelope in CTF/PSF is negative");
  if (usepsf && sqrt(1. / ((float) nEnv * gEnv + startEnv)) > float(N) / 2.0)
    fatal("MAX standard deviation of envelope is larger than allowed KERNEL length");
  if (startAmp < 0 || endAmp > 1)
    fatal("PSF amplitude should be between 0 and 1. start: %lf end: %lf", (double) startAmp, (double) endAmp);
  if (steps)
  {
    steps[0] = gAmp;
    steps[1] = gPhase;
    steps[2] = gEnv;
  }
  std::vector<float> real;
  if (usepsf)
  {
    if (!r2c)
      fatal("PSF mode needs a forward transform");
    real.resize((size_t) N * N);
  }
  const int half = N / 2;
  int n = 0;
  for (int ia = 0; ia < nAmp; ia++)
  {
    const float amp = (float) ia * gAmp + startAmp;
    for (int ip = 0; ip < nPhase; ip++)
    {
      const float phase = (float) ip * gPhase + startPhase;
      for (int ie = 0; ie < nEnv; ie++)
      {
        const float env = (float) ie * gEnv + startEnv;
        float *cur = refCTF + 2 * M * (size_t) n;
        for (size_t e = 0; e < 2 * M; e++)
          cur[e] = 0.f;
        const float quad = sqrtf(1 - amp * amp); // float sqrt of a float argument
        if (usepsf)
        {
          float norm = 0.f;
          for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++)
            {
              const int ri = i <= half ? i : N - i;
              const int rj = j <= half ? j : N - j;
              const float radsq = (float) (ri * ri + rj * rj) * pixelSize * pixelSize;
              const float v = (float) (exp(-radsq * env / 2.0) *
                                       (-amp * cos(radsq * phase / 2.0) - quad * sin(radsq * phase / 2.0)));
              real[(size_t) i * N + j] = v;
              norm += v;
            }
          for (size_t e = 0; e < (size_t) N * N; e++)
            real[e] = real[e] / norm;
          r2c(ctx, N, real.data(), cur);
        }
        else
        {
          if (amp < 0.0000000001)
            fatal("CTF normalization AMP less than threshold < 10^-10");
          float norm = 0.f;
          for (int i = 0; i < H; i++)
            for (int j = 0; j < H; j++)
            {
              const float radsq = (float) (i * i + j * j) / N / N / pixelSize / pixelSize;
              const float v = (float) (exp(-env * radsq / 2.) *
                                       (-amp * cos(phase * radsq / 2.) - quad * sin(phase * radsq / 2.)));
              if (i == 0 && j == 0)
                norm = v;
              const float k = v / norm;
              cur[2 * ((size_t) i * H + j)] = k;
              cur[2 * ((size_t) (N - 1 - i) * H + j)] = k;
            }
        }
        ctfParam[3 * n + 0] = amp;
        ctfParam[3 * n + 1] = phase;
        ctfParam[3 * n + 2] = env;
        n++;
      }
    }
  }
  return n;
}

// param.cpp:1600-1607.  The expression is float up to the first double literal; the second displacement
// divisor is 2*(maxD+1), not (2*maxD+1) -- kept.
float volume_element(float voluang, int g, int maxD, float pixelSize, int nAmp, float gridEnvelop, float gridPhase,
                     float sigB, float sigDef, float sigAmp)
{
  const float head = voluang * (float) g * pixelSize * (float) g * pixelSize;
  double v = head / ((2.f * (float) maxD + 1.));
  v /= (2.f * (float) (maxD + 1.));
  v /= (float) nAmp;
  v *= gridEnvelop;
  v *= gridPhase;
  v /= 4.f;
  v /= M_PI;
  v /= sqrt(2.f * M_PI);
  v /= sigB;
  v /= sigDef;
  v /= sigAmp;
  return (float) v;
}

void InputParams::calculateRefCTF()
{
  nTotCTFs = numberGridPointsCTF_amp * numberGridPointsCTF_phase * numberGridPointsEnvelop;
  const size_t M = (size_t) N * (N / 2 + 1);
  refCTF.assign(2 * M * (size_t) nTotCTFs, 0.f);
  ctfParam.assign(3 * (size_t) nTotCTFs, 0.f);
  float steps[3];
  const int n = ctf_kernels(N, pixelSize, usepsf, startGridCTF_amp, endGridCTF_amp, numberGridPointsCTF_amp,
                            startGridCTF_phase, endGridCTF_phase, numberGridPointsCTF_phase, startGridEnvelop,
                            endGridEnvelop, numberGridPointsEnvelop, refCTF.data(), ctfParam.data(), steps, r2c,
                            r2c_ctx);
  if (n != nTotCTFs)
    fatal("Internal during CTF preparation");
  gridCTF_amp = steps[0];
  gridCTF_phase = steps[1];
  gridEnvelop = steps[2];
  pd.volu = volume_element(voluang, pd.GridSpaceCenter, pd.maxDisplaceCenter, pixelSize, numberGridPointsCTF_amp,
                           gridEnvelop, gridCTF_phase, pd.sigmaPriorbctf, pd.sigmaPriordefo, pd.sigmaPrioramp);
  pd.Ntotpi = (float) (N * N);
  pd.NxDisp = 2 * (int) (pd.maxDisplaceCenter / pd.GridSpaceCenter) + 1;
  pd.NtotDisp = pd.NxDisp * pd.NxDisp;
}

} // namespace bioem_host
