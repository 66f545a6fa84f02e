"""Helpers shared by the parity tests: load a golden case and set the oracle up on it."""
import os

import numpy as np

import oracle as orc          # oracle/oracle.py  (test infrastructure)
import io_formats as iof      # oracle/io_formats.py

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(d for d in os.listdir(GOLDEN) if os.path.isdir(os.path.join(GOLDEN, d)))


def load_case(name):
    d = os.path.join(GOLDEN, name)
    inp = np.load(os.path.join(d, "inputs.npz"))
    P = orc.parse_param_file(os.path.join(d, "param.txt"))
    ol = [str(x) for x in inp["orient_lines"]]
    fmt = str(inp["particles"]) if "particles" in inp.files else "text"
    env = dict(zip([str(k) for k in inp["env_keys"]], [str(v) for v in inp["env_vals"]])) if "env_keys" in inp.files \
        else {}
    raw = inp["maps"]
    # what the particle reader hands to the run: text files carry the maps as they are, MRC stacks are transposed
    # and z-scored by the reader (unless NO_MAP_NORM)
    maps = orc.mrc_reader_maps(raw, P["notnormmap"]) if fmt in ("mrc", "multimrc") else raw
    # model file format the reference was run with (text | pdb | mrc); inputs["model"] always holds the points the
    # reader must deliver (x y z radius density), model.pdb / model.mrc the file itself
    mfmt = str(inp["model_format"]) if "model_format" in inp.files else "text"
    stacks = [int(v) for v in inp["stacks"]] if "stacks" in inp.files else []
    return dict(dir=d, P=P, model=inp["model"], maps=maps, raw_maps=raw, particles=fmt, env=env, model_format=mfmt,
                stacks=stacks, orient_lines=ol if ol else None, algos=[int(a) for a in inp["algos"]])


def oracle_setup(case):
    dbg = int(case["env"]["BIOEM_DEBUG_BREAK"]) if "BIOEM_DEBUG_BREAK" in case.get("env", {}) else None
    return orc.Setup(case["P"], case["model"], case["maps"], case["orient_lines"], debug_break=dbg)


def golden_output(case, algo, plugin=False):
    """Output_Probabilities of the reference CPU path, or (plugin=True) of the reference driving libbioem_hip.so
    through its compareRefMaps virtual (oracle/ref_plugin, GPU=1)."""
    with open(os.path.join(case["dir"], "Output_Probabilities_%salgo%d" % ("plugin_" if plugin else "", algo))) as f:
        return f.read()


def write_mrc_stack(path, data):
    """mode-2 little-endian MRC stack (sections, rows, columns), 1024-byte header, no symmetry bytes."""
    import struct
    ns, nr, nc = data.shape
    hdr = np.zeros(256, dtype="<i4")
    hdr[0:4] = [nc, nr, ns, 2]
    hdr[7:10] = [nc, nr, ns]
    raw = hdr.tobytes()
    raw = raw[:40] + struct.pack("<6f", 100., 100., 100., 90., 90., 90.) + raw[64:]
    with open(path, "wb") as f:
        f.write(raw + data.astype("<f4").tobytes())


def write_case_inputs(case, d):
    """Writes the model / particle / orientation files of a golden case into directory d exactly as the reference was
    fed (oracle/make_golden.py) and returns the command-line options that select them."""
    d = str(d)
    if case["model_format"] == "pdb":
        margs = ["--Modelfile", os.path.join(case["dir"], "model.pdb"), "--ReadPDB"]
    elif case["model_format"] == "mrc":
        margs = ["--Modelfile", os.path.join(case["dir"], "model.mrc"), "--ReadModelMRC"]
    else:
        iof.write_text_model(os.path.join(d, "model.txt"), case["model"])
        margs = ["--Modelfile", "model.txt"]
    if case["particles"] == "multimrc":
        lo = 0
        with open(os.path.join(d, "list.txt"), "w") as f:
            for k, cnt in enumerate(case["stacks"]):
                write_mrc_stack(os.path.join(d, "stack%d.mrc" % k), case["raw_maps"][lo:lo + cnt])
                f.write("stack%d.mrc\n" % k)  # relative to the run directory (the reader's name buffer is short)
                lo += cnt
        pargs = ["--Particlesfile", "list.txt", "--ReadMRC", "--ReadMultipleMRC"]
    elif case["particles"] == "mrc":
        write_mrc_stack(os.path.join(d, "particles.mrc"), case["raw_maps"])
        pargs = ["--Particlesfile", "particles.mrc", "--ReadMRC"]
    else:
        iof.write_text_particles(os.path.join(d, "particles.txt"), case["maps"])
        pargs = ["--Particlesfile", "particles.txt"]
    oargs = []
    if case["orient_lines"]:
        with open(os.path.join(d, "orient.txt"), "w") as f:
            f.write("%d\n" % len(case["orient_lines"]) + "\n".join(case["orient_lines"]) + "\n")
        oargs = ["--ReadOrientation", "orient.txt"]
    return margs + pargs + oargs
