// model.cpp -- structural model readers (text x y z radius density; PDB C-alpha trace) and the
// centre-of-density-mass shift.  Behaviour follows /root/reference/model.cpp:85-329 (PDB),
// 419-601 (text), 604-672 (centre of mass), 738-844 (residue radius / electron tables).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "bioem_host.h"

namespace bioem_host
{

namespace
{
// residue -> (radius [A], number of electrons), reference model.cpp:738-844
struct Residue
{
  const char *name;
  float radius, electrons;
};
const Residue kResidues[] = {{"CYS", 2.75f, 64.0f}, {"PHE", 3.2f, 88.0f},  {"LEU", 3.1f, 72.0f},  {"TRP", 3.4f, 108.0f},
                             {"VAL", 2.95f, 64.0f}, {"ILE", 3.1f, 72.0f},  {"MET", 3.1f, 80.0f},  {"HIS", 3.05f, 82.0f},
                             {"TYR", 3.25f, 96.0f}, {"ALA", 2.5f, 48.0f},  {"GLY", 2.25f, 40.0f}, {"PRO", 2.8f, 62.0f},
                             {"ASN", 2.85f, 66.0f}, {"THR", 2.8f, 64.0f},  {"SER", 2.6f, 56.0f},  {"ARG", 3.3f, 93.0f},
                             {"GLN", 3.0f, 78.0f},  {"ASP", 2.8f, 59.0f},  {"LYS", 3.2f, 79.0f},  {"GLU", 2.95f, 53.0f}};

const Residue &residue(const char *name)
{
  for (const Residue &r : kResidues)
    if (strcmp(r.name, name) == 0)
      return r;
  fatal("Amino acid: %s", name);
}

std::string slurp(const char *file)
{
  FILE *f = fopen(file, "rb");
  if (!f)
    fatal("Opening file: %s", file);
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  rewind(f);
  std::string s((size_t) n, '\0');
  if (n > 0 && fread(&s[0], 1, (size_t) n, f) != (size_t) n)
    fatal("Reading error");
  fclose(f);
  return s;
}
} // namespace

void Model::readTextFile(const InputParams &p, const char *file)
{
  std::cout << "Note: Reading model in simple text format\n";
  std::cout << "----  x   y   z  radius  density ------- \n";
  const std::string name(file);
  const size_t found = name.find(".pdb"), endpos = name.find_last_not_of(" \t");
  if (found < endpos)
  {
    warn("PDB detected in file name: %s. Are you sure you do not need --ReadPDB? "
         "If so then you must include the keyword IGNORE_PDB in inputfile",
         file);
    if (!p.ignorePDB)
      fatal("PDB is not ignored");
  }
  const std::string buf = slurp(file);
  points.clear();
  NormDen = 0.f;
  // one record per line; a newline in the last byte does not open a record (model.cpp:497-505)
  size_t pos = 0;
  while (pos < buf.size())
  {
    size_t eol = buf.find('\n', pos);
    if (eol == std::string::npos)
      eol = buf.size();
    const std::string line = buf.substr(pos, eol - pos);
    double v[5];
    if (sscanf(line.c_str(), "%lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], &v[4]) != 5)
      fatal("line parsed by sscanf has wrong argument");
    if (v[3] < 0)
      fatal("Radius must be positive");
    bioem_hip_model_point q{};
    q.pos[0] = (float) v[0];
    q.pos[1] = (float) v[1];
    q.pos[2] = (float) v[2];
    q.radius = (float) v[3];
    q.density = (float) v[4];
    NormDen += q.density;
    points.push_back(q);
    pos = eol + 1;
    if (pos == buf.size())
      break;
  }
  std::cout << "Protein structure read from Standard File\n";
}

void Model::readPDBFile(const char *file)
{
  const std::string name(file);
  const size_t found = name.find(".pdb"), endpos = name.find_last_not_of(" \t");
  if (found > endpos)
    warn("PDB extension NOT detected in file name: %s. Are you sure you want to read a PDB?", file);
  const std::string buf = slurp(file);
  points.clear();
  NormDen = 0.f;
  size_t pos = 0;
  while (pos < buf.size())
  {
    size_t eol = buf.find('\n', pos);
    if (eol == std::string::npos)
      eol = buf.size();
    const std::string line = buf.substr(pos, eol - pos);
    pos = eol + 1;
    // columns (1-based): 1-6 record, 13-16 atom name, 18-20 residue, 31-54 x y z
    if (line.size() < 54)
      continue;
    char type[8] = {0}, atom[8] = {0}, res[8] = {0};
    if (sscanf(line.substr(0, 6).c_str(), "%6s", type) != 1)
      continue;
    if (strcmp(type, "ATOM") != 0)
      continue;
    if (sscanf(line.substr(12, 4).c_str(), "%4s", atom) != 1 || strcmp(atom, "CA") != 0)
      continue;
    if (sscanf(line.substr(17, 3).c_str(), "%3s", res) != 1)
      fatal("line parsed by sscanf has wrong argument");
    double x, y, z;
    if (sscanf(line.substr(30, 24).c_str(), "%lf %lf %lf", &x, &y, &z) != 3)
      fatal("line parsed by sscanf has wrong argument");
    const Residue &r = residue(res);
    bioem_hip_model_point q{};
    q.pos[0] = (float) x;
    q.pos[1] = (float) y;
    q.pos[2] = (float) z;
    q.radius = r.radius;
    q.density = r.electrons;
    NormDen += q.density;
    points.push_back(q);
  }
  std::cout << "Protein structure read from PDB\n";
}

// density map as a model (model.cpp:332-416): every voxel becomes a point of radius 2*pixelSize whose
// "density" is the voxel value; voxel (i,j,k), counted from 1 in file order i slowest, sits at
// ((i - nx/2)*px, (j - ny/2)*px, (k - nz/2)*px).
void Model::readMRCFile(const InputParams &p, const char *file)
{
  const std::string name(file);
  if (name.find(".mrc") > name.find_last_not_of(" \t"))
    warn("MRC extension NOT detected in file name: %s. Are you sure you want to read an MRC?", file);
  const MrcHeader h = mrc_read_header(file);
  if (h.mode != 2)
    fatal("MRC mode: %d. Currently mode 2 is the only one allowed", h.mode);
  const int nx = h.nc, ny = h.nr, nz = h.ns;
  const size_t total = (size_t) nx * ny * nz;
  FILE *f = fopen(file, "rb");
  if (!f)
    fatal("Opening MRC: %s", file);
  if (fseek(f, 1024 + (long) h.nsymbt, SEEK_SET) != 0)
    fatal("Converting Data: %s", file);
  std::vector<unsigned int> raw(total);
  if (fread(raw.data(), 4, total, f) != total)
    fatal("Converting Data: %s", file);
  fclose(f);
  points.clear();
  points.reserve(total);
  NormDen = 0.f;
  size_t e = 0;
  for (int i = 1; i <= nx; i++)
    for (int j = 1; j <= ny; j++)
      for (int k = 1; k <= nz; k++, e++)
      {
        unsigned int v = raw[e];
        if (h.swap)
          v = mrc_bswap32(v);
        float c;
        memcpy(&c, &v, 4);
        bioem_hip_model_point q{};
        q.pos[0] = (float) ((i - nx / 2.0) * p.pixelSize);
        q.pos[1] = (float) ((j - ny / 2.0) * p.pixelSize);
        q.pos[2] = (float) ((k - nz / 2.0) * p.pixelSize);
        q.radius = (float) (2.0 * p.pixelSize);
        q.density = c;
        NormDen += q.density;
        points.push_back(q);
      }
  std::cout << "Protein structure read from MRC\n";
}

void Model::centerDensityMass()
{
  float r[3] = {0.f, 0.f, 0.f};
  for (const auto &p : points)
    for (int k = 0; k < 3; k++)
      r[k] += p.pos[k] * p.density;
  for (int k = 0; k < 3; k++)
    r[k] /= NormDen;
  for (auto &p : points)
    for (int k = 0; k < 3; k++)
      p.pos[k] -= r[k];
}

void Model::readModel(const InputParams &p, const char *file)
{
  if (readPDB)
    readPDBFile(file);
  else if (readModelMRC)
    readMRCFile(p, file);
  else
    readTextFile(p, file);
  std::cout << "Total Number of Voxels " << points.size();
  std::cout << "\nTotal Number of Electrons " << NormDen;
  std::cout << "\n+++++++++++++++++++++++++++++++++++++++++ \n";
  if (!p.nocentermass)
    centerDensityMass();
}

} // namespace bioem_host
