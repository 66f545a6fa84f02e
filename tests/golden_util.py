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
    maps = orc.mrc_reader_maps(raw, P["notnormmap"]) if fmt == "mrc" else raw
    return dict(dir=d, P=P, model=inp["model"], maps=maps, raw_maps=raw, particles=fmt, env=env,
                orient_lines=ol if ol else None, algos=[int(a) for a in inp["algos"]])


def oracle_setup(case):
    dbg = int(case["env"]["BIOEM_DEBUG_BREAK"]) if "BIOEM_DEBUG_BREAK" in case.get("env", {}) else None
    return orc.Setup(case["P"], case["model"], case["maps"], case["orient_lines"], debug_break=dbg)


def golden_output(case, algo):
    with open(os.path.join(case["dir"], "Output_Probabilities_algo%d" % algo)) as f:
        return f.read()
