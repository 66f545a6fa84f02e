#!/usr/bin/env python3
"""Reads the gfx950 code object out of bioem_amd/lib/libbioem_hip.so (no GPU needed) and prints one row per kernel:
registers, LDS, scratch, spills.  Exit code 1 when a shipped kernel spills vector registers or uses scratch memory
(`.vgpr_spill_count` / `.private_segment_fixed_size` in the code-object notes), or when a memory instruction of any
kernel sits in a waterfall loop (disassembly) -- `make -C bioem_amd/csrc check` and __graft_entry__.build() run it, so
such a kernel fails the build.

usage: scripts/check_code_object.py [--json out.json] [--allow REGEX] [--so path] [--quiet]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


class ToolMissing(Exception):
    pass


def tool(name):
    path = os.path.join(LLVM, name)
    if not os.path.exists(path):
        raise ToolMissing(path)
    return path


def extract_code_objects(so, workdir):
    """The library is linked from one object per kernel family: .hip_fatbin holds one offload bundle per translation
    unit (magic __CLANG_OFFLOAD_BUNDLE__), each with its own gfx950 code object."""
    fat = os.path.join(workdir, "fatbin")
    subprocess.check_call([tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = []
    at = blob.find(magic)
    while at >= 0:
        starts.append(at)
        at = blob.find(magic, at + 1)
    cos = []
    for k, st in enumerate(starts):
        part = os.path.join(workdir, "bundle%d" % k)
        with open(part, "wb") as f:
            f.write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, "gfx950_%d.co" % k)
        subprocess.check_call([tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        cos.append(co)
    return cos


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return p.stdout.split("\n")[:len(names)]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(CompareArgs\)|\(\(anonymous namespace\)::CompareArgs\)", "", name)
    return re.sub(r"\(.*\)$", "", name)


def kernels_of(co):
    notes = subprocess.check_output([tool("llvm-readelf"), "--notes", co], text=True)
    # the per-kernel maps list their keys alphabetically; a kernel's block ends at `.wavefront_size`
    ks, cur = [], {}
    for ln in notes.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in ("agpr_count", "group_segment_fixed_size", "max_flat_workgroup_size", "private_segment_fixed_size",
                 "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "wavefront_size"):
            cur[k] = int(v)
        elif k == "symbol":
            cur["symbol"] = v.strip("'")
        elif k == "name" and v.startswith(("_Z", "'_Z")):
            cur["name"] = v.strip("'")
        if k == "wavefront_size":
            ks.append(cur)
            cur = {}
    return ks


def waterfall_kernels(co):
    """Kernels in which a memory instruction sits in a waterfall loop (its buffer descriptor or address ended up in
    vector registers: 4 v_readfirstlane + 2 v_cmp + exec juggling around every load)."""
    dis = subprocess.check_output([tool("llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    out, name, body = [], None, []

    def flush():
        if name is None:
            return
        n = 0
        for t, ln in enumerate(body):
            if "buffer_load" in ln or "buffer_store" in ln or "global_load" in ln or "global_store" in ln:
                ctx = " ".join(body[max(0, t - 6):t])
                if "v_readfirstlane_b32" in ctx and "s_and_saveexec_b64" in ctx:
                    n += 1
        if n:
            out.append((name, n))

    for ln in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            flush()
            name, body = m.group(1), []
        else:
            body.append(ln)
    flush()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "bioem_amd", "lib", "libbioem_hip.so"))
    ap.add_argument("--json")
    ap.add_argument("--allow", default=None, help="regex of kernel names that may use scratch (diagnostic builds)")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    # a missing tool (or a tool that no longer understands the file) is not a spilling kernel: say so and leave with a
    # code of its own (3), which `make check` / __graft_entry__.build() report without failing the product build
    try:
        if subprocess.run(["c++filt", "--version"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode:
            raise ToolMissing("c++filt")
        with tempfile.TemporaryDirectory() as d:
            ks, wf = [], []
            for co in extract_code_objects(a.so, d):
                ks += kernels_of(co)
                wf += waterfall_kernels(co)
        names = demangle([k.get("name", k.get("symbol", "?")) for k in ks])
    except (ToolMissing, FileNotFoundError, subprocess.CalledProcessError) as e:
        print("check_code_object: code-object tools unavailable or failed (%s): check SKIPPED" % e, file=sys.stderr)
        sys.exit(3)
    if not ks:
        print("check_code_object: no kernels found in %s (format of the code-object notes changed?): check SKIPPED" % a.so,
              file=sys.stderr)
        sys.exit(3)
    rows = []
    for k, n in zip(ks, names):
        alloc = -(-max(1, k.get("vgpr_count", 0) + k.get("agpr_count", 0)) // 8) * 8
        rows.append({"kernel": short(n), "vgpr": k.get("vgpr_count", 0), "agpr": k.get("agpr_count", 0),
                     "waves_per_simd_by_registers": min(8, 512 // alloc), "sgpr": k.get("sgpr_count", 0),
                     "lds_static": k.get("group_segment_fixed_size", 0), "scratch": k.get("private_segment_fixed_size", 0),
                     "vgpr_spills": k.get("vgpr_spill_count", 0), "sgpr_spills": k.get("sgpr_spill_count", 0),
                     "threads": k.get("max_flat_workgroup_size", 0)})
    rows.sort(key=lambda r: r["kernel"])
    allow = re.compile(a.allow) if a.allow else None
    bad = [r for r in rows if (r["scratch"] or r["vgpr_spills"]) and not (allow and allow.search(r["kernel"]))]
    if not a.quiet:
        print("%-64s %5s %5s %5s %7s %7s %6s %6s" % ("kernel", "vgpr", "w/EU", "sgpr", "scratch", "vspill", "sspill", "thr"))
        for r in rows:
            print("%-64s %5d %5d %5d %7d %7d %6d %6d" % (r["kernel"][:64], r["vgpr"], r["waves_per_simd_by_registers"],
                                                        r["sgpr"], r["scratch"], r["vgpr_spills"], r["sgpr_spills"],
                                                        r["threads"]))
    print("%d kernels, %d with vector spills or scratch, %d with SGPR spills (to VGPR lanes)"
          % (len(rows), len([r for r in rows if r["scratch"] or r["vgpr_spills"]]),
             len([r for r in rows if r["sgpr_spills"]])))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"kernels": rows, "spilling": [r["kernel"] for r in bad]}, f, indent=1)
    if wf:
        print("FAIL: memory instructions inside waterfall loops (non-uniform descriptor / address):", file=sys.stderr)
        for n, c in zip(demangle([w[0] for w in wf]), [w[1] for w in wf]):
            print("  %s: %d" % (short(n), c), file=sys.stderr)
        sys.exit(1)
    if bad:
        print("FAIL: kernels with vector-register spills / scratch:", file=sys.stderr)
        for r in bad:
            print("  %s: %d spilled VGPRs, %d B scratch" % (r["kernel"], r["vgpr_spills"], r["scratch"]), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
