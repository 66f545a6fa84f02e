"""Writers/parsers for the reference's on-disk formats (TEST INFRASTRUCTURE ONLY).

Used by oracle/make_golden.py (to feed the reference binary) and by tests/ (to
feed the product CLI and to parse outputs).  Nothing in the product path may
import this module.

Formats restated from the reference readers (not copied):
  * parameter file ........ /root/reference/param.cpp:121-527 (keyword per line, space separated)
  * text model ............ /root/reference/model.cpp:419-601 ("%lf %lf %lf %lf %lf", no blank lines)
  * text particles ........ /root/reference/map.cpp:268-414 ("PARTICLE" header, 33-byte records %8d%8d%16.8f)
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
