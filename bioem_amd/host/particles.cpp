// particles.cpp -- experimental particle image readers: text ("PARTICLE" blocks of fixed 33-byte
// records) and MRC mode-2 stacks (single file or a list of files).  Behaviour follows
// /root/reference/map.cpp:81-265 (MRC list handling), 268-414 (text), 663-936 + include/mrc.h (MRC):
// 1024-byte header + NSYMBT bytes, endianness guessed from header range violations, images stored
// transposed and z-score normalised with float accumulators unless NO_MAP_NORM.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "bioem_host.h"

namespace bioem_host
{

void ParticleStack::readTextMaps(const char *file)
{
  FILE *f = fopen(file, "rb");
  if (!f)
    fatal("Opening file: %s", file);
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  rewind(f);
  std::string buf((size_t) size, '\0');
  if (size > 0 && fread(&buf[0], 1, (size_t) size, f) != (size_t) size)
    fatal("Reading error");
  fclose(f);
  if (buf.compare(0, 8, "PARTICLE") != 0)
    fatal("Missing correct standard map format: PARTICLE HEADER");
  const int refMapSize = N * N;
  maps.clear();
  ntot = 0;
  size_t pos = 0;
  while (pos < buf.size())
  {
    // header line
    size_t k = buf.find('\n', pos);
    if (k == std::string::npos)
      break;
    k++;
    maps.resize((size_t) (ntot + 1) * refMapSize);
    float *m = &maps[(size_t) ntot * refMapSize];
    int i = 0, j = 0, count = 0;
    // records are exactly 33 bytes: %8d%8d%16.8f\n (map.cpp:372-374)
    while (k < buf.size() && buf[k] != 'P')
    {
      char rec[33] = {0};
      const size_t len = std::min<size_t>(32, buf.size() - k);
      memcpy(rec, buf.data() + k, len);
      k += 33;
      double z = 0.;
      if (sscanf(rec, "%d %d %lf", &i, &j, &z) != 3)
        fatal("line parsed by sscanf has wrong argument");
      if (i > -1 && i < N && j > -1 && j < N)
      {
        count++;
        m[i * N + j] = (float) z;
      }
      else
        fatal("Reading map (Map number %d, i %d, j %d)", ntot, i, j);
    }
    if (i != N - 1 || j != N - 1 || count != refMapSize)
      fatal("Inconsistent number of pixels in maps and inputfile ( %d, i %d, j %d)", count, i, j);
    ntot++;
    pos = k;
  }
  std::cout << ".Particle Maps read from Standard File: " << ntot << "\n";
}

namespace
{
unsigned int bswap32(unsigned int v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

int range_violations(const unsigned int *w, int swap)
{
  auto I = [&](int k) {
    unsigned int v = swap ? bswap32(w[k]) : w[k];
    int r;
    memcpy(&r, &v, 4);
    return r;
  };
  auto F = [&](int k) {
    unsigned int v = swap ? bswap32(w[k]) : w[k];
    float r;
    memcpy(&r, &v, 4);
    return r;
  };
  int n = 0;
  for (int k = 0; k < 3; k++) // nc nr ns
    n += (I(k) > 5000) + (I(k) < 0);
  for (int k = 4; k < 7; k++) // ncstart nrstart nsstart
    n += (I(k) > 5000) + (I(k) < -5000);
  for (int k = 7; k < 10; k++) // mx my mz
    n += (I(k) > 5000) + (I(k) < 0);
  for (int k = 13; k < 16; k++) // alpha beta gamma
    n += (F(k) > 360.0f) + (F(k) < -360.0f);
  return n;
}

} // namespace

MrcHeader mrc_read_header(const char *file)
{
  FILE *f = fopen(file, "rb");
  if (!f)
    fatal("Opening MRC: %s", file);
  unsigned int w[256];
  if (fread(w, 4, 256, f) != 256)
    fatal("Reading MRC header: %s", file);
  fclose(f);
  const int v0 = range_violations(w, 0), v1 = range_violations(w, 1);
  MrcHeader h;
  if (v0 < v1)
  {
    h.swap = 0;
    if (v0 > 0)
      warn("%i header field range violations detected in file %s", v0, file);
  }
  else
  {
    h.swap = 1;
    if (v1 > 0)
      warn("%i header field range violations detected in file %s", v1, file);
  }
  auto I = [&](int k) {
    unsigned int v = h.swap ? bswap32(w[k]) : w[k];
    int r;
    memcpy(&r, &v, 4);
    return r;
  };
  h.nc = I(0);
  h.nr = I(1);
  h.ns = I(2);
  h.mode = I(3);
  h.nsymbt = I(23);
  return h;
}

unsigned int mrc_bswap32(unsigned int v) { return bswap32(v); }

static void read_one_mrc(ParticleStack &S, const InputParams &p, const char *file)
{
  const MrcHeader h = mrc_read_header(file);
  printf("\n+++++++++++++++++++++++++++++++++++++++++++\n");
  printf("Reading Information from MRC: %s \n", file);
  printf("Number Columns  = %8d \n", h.nc);
  printf("Number Rows     = %8d \n", h.nr);
  printf("Number Sections = %8d \n", h.ns);
  printf("MODE = %4d (only data type mode 2: 32-bit)\n", h.mode);
  printf("NSYMBT = %4d (# bytes symmetry operators)\n", h.nsymbt);
  const int N = S.N;
  if (h.nr != N || h.nc != N)
    fatal("Inconsistent number of pixels in maps and inputfile ( %d, i %d, j %d)", N, h.nc, h.nr);
  if (h.mode != 2)
    fatal("MRC mode: %d. Currently mode 2 is the only one allowed", h.mode);
  FILE *f = fopen(file, "rb");
  if (!f)
    fatal("Opening MRC: %s", file);
  if (fseek(f, 1024 + (long) h.nsymbt, SEEK_SET) != 0)
    fatal("Converting Data: %s", file);
  const size_t mapsz = (size_t) N * N;
  std::vector<unsigned int> raw(mapsz);
  S.maps.resize((size_t) (S.ntot + h.ns) * mapsz);
  for (int s = 0; s < h.ns; s++)
  {
    if (fread(raw.data(), 4, mapsz, f) != mapsz)
      fatal("Converting Data: %s", file);
    float *m = &S.maps[(size_t) (S.ntot + s) * mapsz];
    float st = 0.0f, st2 = 0.0f;
    for (int j = 0; j < h.nr; j++)
      for (int i = 0; i < h.nc; i++)
      {
        unsigned int v = raw[(size_t) j * h.nc + i];
        if (h.swap)
          v = bswap32(v);
        float c;
        memcpy(&c, &v, 4);
        m[(size_t) i * N + j] = c; // transposed store, map.cpp:824
        st += c;
        st2 += c * c;
      }
    if (!p.notnormmap)
    {
      st /= float(h.nr * h.nc);
      st2 = sqrtf(st2 / float(h.nr * h.nc) - st * st);
      for (size_t e = 0; e < mapsz; e++)
        m[e] = m[e] / st2 - st / st2;
    }
  }
  fclose(f);
  S.ntot += h.ns;
}

void ParticleStack::readMRCMaps(const InputParams &p, const char *file)
{
  ntot = 0;
  maps.clear();
  if (readMultMRC)
  {
    std::cout << "Opening File with MRC list names: " << file << "\n";
    std::ifstream input(file);
    if (!input.good())
      fatal("Failed to open file contaning MRC names: %s", file);
    std::string line;
    while (std::getline(input, line))
    {
      if (line.empty())
        fatal("line parsed by sscanf has wrong argument"); // a blank line is fatal in the reference (map.cpp:118)
      if (line.find("mrc") > line.find_last_not_of(" \t"))
        warn("MRC extension NOT detected in file name: %s. Are you sure you want to read an MRC?", file);
      read_one_mrc(*this, p, line.c_str());
    }
    std::cout << "\n+++++++++++++++++++++++++++++++++++++++++++ \n";
    std::cout << "Particle Maps read from MULTIPLE MRC Files in: " << file << "\n";
  }
  else
  {
    const std::string name(file);
    if (name.find("mrc") > name.find_last_not_of(" \t"))
      warn("MRC extension NOT detected in file name: %s. Are you sure you want to read an MRC?", file);
    read_one_mrc(*this, p, file);
    std::cout << "\n++++++++++++++++++++++++++++++++++++++++++ \n";
    std::cout << "Particle Maps read from ONE MRC File: " << file << "\n";
  }
}

void ParticleStack::readRefMaps(const InputParams &p, const char *file)
{
  N = p.N;
  if (readMRC)
    readMRCMaps(p, file);
  else
    readTextMaps(file);
  if (getenv("BIOEM_DEBUG_NMAPS")) // map.cpp:545-548
  {
    const int n = atoi(getenv("BIOEM_DEBUG_NMAPS"));
    if (n < ntot)
    {
      ntot = n;
      maps.resize((size_t) ntot * N * N);
    }
  }
  std::cout << "Total Number of particles: " << ntot;
  std::cout << "\n+++++++++++++++++++++++++++++++++++++++++++ \n";
}

} // namespace bioem_host
