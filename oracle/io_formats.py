"""Writers/parsers for the reference's on-disk formats (TEST INFRASTRUCTURE ONLY).

Used by oracle/make_golden.py (to feed the reference binary) and by tests/ (to
feed the product CLI and to parse outputs).  Nothing in the product path may
import this module.

Formats restated from the reference readers (not copied):
  * parameter file ........ /root/reference/param.cpp:121-527 (keyword per line, space separated)
  * text model ............ /root/reference/model.cpp:419-601 ("%lf %lf %lf %lf %lf", no blank lines)
  * text particles ........ /root/reference/map.cpp:268-414 ("PARTICLE" header, 33-byte records %8d%8d%16.8f)
  * PDB model ............. /root/reference/model.cpp:85-329 (ATOM records with atom name CA; columns 18-20 residue,
                            31-54 x y z; radius / electrons per residue type model.cpp:738-844)
  * MRC density model ..... /root/reference/model.cpp:332-416 (mode-2 volume; one point per voxel, radius 2 px)
  * orientation list ...... /root/reference/param.cpp:1213-1327 (count line, then 12-char columns)
  * Output_Probabilities .. /root/reference/bioem.cpp:1077-1222
  * ANG_PROB .............. /root/reference/bioem.cpp:1050-1075,1245-1365
"""
import re
import numpy as np


def write_param_file(path, keywords):
    """keywords: list of (KEY, [values...]) in order; values may be empty."""
    with open(path, "w") as f:
        for key, vals in keywords:
            if vals:
                f.write(key + " " + " ".join(str(v) for v in vals) + "\n")
            else:
                f.write(key + "\n")


def write_text_model(path, pts):
    """pts: (n,5) array x y z radius density."""
    with open(path, "w") as f:
        lines = ["%.6f %.6f %.6f %.6f %.6f" % tuple(float(v) for v in p) for p in pts]
        f.write("\n".join(lines) + "\n")


# residue -> (radius [A], electrons), restating /root/reference/model.cpp:738-844
RESIDUES = {"CYS": (2.75, 64.0), "PHE": (3.2, 88.0), "LEU": (3.1, 72.0), "TRP": (3.4, 108.0), "VAL": (2.95, 64.0),
            "ILE": (3.1, 72.0), "MET": (3.1, 80.0), "HIS": (3.05, 82.0), "TYR": (3.25, 96.0), "ALA": (2.5, 48.0),
            "GLY": (2.25, 40.0), "PRO": (2.8, 62.0), "ASN": (2.85, 66.0), "THR": (2.8, 64.0), "SER": (2.6, 56.0),
            "ARG": (3.3, 93.0), "GLN": (3.0, 78.0), "ASP": (2.8, 59.0), "LYS": (3.2, 79.0), "GLU": (2.95, 53.0)}


def write_pdb_model(path, pts, resnames):
    """C-alpha trace as fixed-column PDB records (80 columns, so that the reference's column reads never leave a
    line), interleaved with records its reader must skip: backbone N/C atoms, HETATM, TER, REMARK, END."""
    def rec(kind, serial, atom, res, seq, x, y, z, elem):
        ln = "%-6s%5d  %-3s %3s A%4d    %8.3f%8.3f%8.3f  1.00  0.00          %2s" % (kind, serial, atom, res, seq, x, y,
                                                                                   z, elem)
        return ln.ljust(80) + "\n"
    with open(path, "w") as f:
        f.write("REMARK   1 GENERATED TEST MODEL (C-ALPHA TRACE)".ljust(80) + "\n")
        serial = 1
        for i, (p, res) in enumerate(zip(pts, resnames)):
            f.write(rec("ATOM", serial, "N", res, i + 1, p[0] - 1.2, p[1] + 0.4, p[2], "N"))
            f.write(rec("ATOM", serial + 1, "CA", res, i + 1, p[0], p[1], p[2], "C"))
            f.write(rec("ATOM", serial + 2, "C", res, i + 1, p[0] + 1.1, p[1] - 0.7, p[2] + 0.5, "C"))
            serial += 3
        f.write("TER".ljust(80) + "\n")
        f.write(rec("HETATM", serial, "CA", "CA", len(pts) + 1, 1.0, 2.0, 3.0, "CA"))
        f.write("END".ljust(80) + "\n")


def mrc_volume_points(vol, px):
    """Points the reference's MRC model reader builds from a volume vol[i][j][k] stored in file order (first index
    slowest): position ((i+1) - n/2) * px per axis, radius 2 px, density = voxel value (model.cpp:380-399)."""
    nx, ny, nz = vol.shape
    out = []
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            for k in range(1, nz + 1):
                out.append([(i - nx / 2.0) * px, (j - ny / 2.0) * px, (k - nz / 2.0) * px, 2.0 * px,
                            float(vol[i - 1, j - 1, k - 1])])
    return np.array(out, dtype=np.float64)


def write_mrc_volume(path, vol):
    """mode-2 little-endian MRC volume; header words 1-3 = (nx, ny, nz) in the order the reference's model reader
    loops over them (first slowest), cell 100 A / 90 degrees."""
    import struct
    nx, ny, nz = vol.shape
    hdr = np.zeros(256, dtype="<i4")
    hdr[0:4] = [nx, ny, nz, 2]
    hdr[7:10] = [nx, ny, nz]
    raw = hdr.tobytes()
    raw = raw[:40] + struct.pack("<6f", 100., 100., 100., 90., 90., 90.) + raw[64:]
    with open(path, "wb") as f:
        f.write(raw + np.ascontiguousarray(vol, dtype="<f4").tobytes())


def read_text_model(path):
    return np.loadtxt(path, dtype=np.float64).reshape(-1, 5)


def write_text_particles(path, maps):
    """maps: (nP,N,N) float array.  i slow, j fast; record = %8d%8d%16.8f\\n (33 bytes)."""
    maps = np.asarray(maps)
    nP, N, _ = maps.shape
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    ii = ii.ravel()
    jj = jj.ravel()
    with open(path, "w") as f:
        for p in range(nP):
            f.write("PARTICLE %d\n" % (p + 1))
            vals = maps[p].ravel()
            f.write("".join("%8d%8d%16.8f\n" % (i, j, v) for i, j, v in zip(ii, jj, vals)))


def write_orientation_list(path, rows, priors=None):
    """rows: (n,4) quaternions or (n,3) Euler angles; 12-char columns."""
    rows = np.asarray(rows, dtype=np.float64)
    with open(path, "w") as f:
        f.write("%d\n" % len(rows))
        for k, r in enumerate(rows):
            s = "".join("%12.8f" % v for v in r)
            if priors is not None:
                s += "%12.8f" % priors[k]
            f.write(s + "\n")


def read_orientation_list(path):
    with open(path) as f:
        lines = f.read().split("\n")
    n = int(lines[0][:12])
    out = []
    for ln in lines[1:1 + n]:
        cols = [float(ln[c:c + 12]) for c in range(0, 48, 12) if ln[c:c + 12].strip()]
        out.append(cols)
    return np.array(out, dtype=np.float32)


_num = r"[-+]?(?:\d+\.\d*|\.\d+|\d+|nan|inf)(?:[eE][-+]?\d+)?"


def parse_output_probabilities(path_or_text):
    """Returns list of dicts per RefMap: logp, constant, maxlogp, angles[3|4], ctf[3], cx, cy, norm, mu."""
    if "\n" in path_or_text:
        text = path_or_text
    else:
        with open(path_or_text) as f:
            text = f.read()
    res = {}
    for ln in text.split("\n"):
        m = re.match(r"RefMap: (\d+) LogProb:\s+(" + _num + r") Constant: (" + _num + ")", ln)
        if m:
            d = res.setdefault(int(m.group(1)), {})
            d["logp"] = float(m.group(2))
            d["constant"] = float(m.group(3))
            continue
        m = re.match(r"RefMap: (\d+) Maximizing Param: (.*)$", ln)
        if m:
            d = res.setdefault(int(m.group(1)), {})
            toks = m.group(2).split()
            nums = []
            for t in toks:
                try:
                    nums.append(float(t))
                except ValueError:
                    pass
            # maxlogp, angles (3 or 4), amp, defocus, env, cx, cy, norm, mu
            d["maxlogp"] = nums[0]
            nang = len(nums) - 8
            d["angles"] = nums[1:1 + nang]
            d["ctf"] = nums[1 + nang:4 + nang]
            d["cx"] = int(nums[4 + nang])
            d["cy"] = int(nums[5 + nang])
            d["norm"] = nums[6 + nang]
            d["mu"] = nums[7 + nang]
    return [res[k] for k in sorted(res)]


def parse_ang_prob(path_or_text):
    """Returns dict map -> list of (angles..., logp, logsum, const, numconst[, prior])."""
    if "\n" in path_or_text:
        text = path_or_text
    else:
        with open(path_or_text) as f:
            text = f.read()
    out = {}
    for ln in text.split("\n"):
        if "Separated:" not in ln:
            continue
        left, right = ln.split("Separated:")
        l = left.split()
        r = right.split()
        out.setdefault(int(l[0]), []).append(
            dict(angles=[float(v) for v in l[1:-1]], logp=float(l[-1]),
                 sep=[float(v) for v in r]))
    return out
