"""CPU tests: the C-ABI library loads and exports every symbol include/bioem_hip.h declares (no compute calls
without a GPU), and the C++ host layer (parameter file, orientation sets, CTF kernels, volume element,
model / particle readers) agrees with the oracle's restatement of the reference on the golden inputs."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import io_formats as iof
import oracle as orc
from golden_util import CASES, load_case, oracle_setup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bioem_amd import engine
    L = engine.load_library()
    with open(os.path.join(ROOT, "include", "bioem_hip.h")) as f:
        hdr = f.read()
    declared = sorted(set(re.findall(r"\b(bioem_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libbioem_hip.so does not export %s" % name
    assert sorted(engine.EXPORTS) == declared


def test_struct_sizes_match_reference_layout():
    from bioem_amd import engine
    assert C.sizeof(engine.ParamDevice) == 60          # bioem_param_device
    assert engine.PROB_MAP_DTYPE.itemsize == 40         # bioem_Probability_map
    assert engine.PROB_ANGLE_DTYPE.itemsize == 16
    assert engine.PARAM5_DTYPE.itemsize == 20           # myparam5_t
    assert engine.POINT_DTYPE.itemsize == 24            # bioem_model_point
    L = engine.load_library()
    assert L.bioem_hip_prob_size(10, 7, 0) == 400
    assert L.bioem_hip_prob_size(10, 7, 3) == 400 + 10 * 7 * 16


def test_missing_library_fails_loudly(monkeypatch):
    from bioem_amd import engine
    monkeypatch.setattr(engine, "_lib", None)
    monkeypatch.setattr(engine, "lib_path", lambda: "/nonexistent/libbioem_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.load_library()


@pytest.mark.parametrize("name", [c for c in CASES if "psf" not in c])
def test_host_setup_equals_oracle(name, tmp_path):
    """readParameters + CalculateGridsParam + CalculateRefCTF of the C++ host layer vs the oracle: bitwise."""
    from bioem_amd import hostlib
    case = load_case(name)
    # set-up as configure() builds it before BIOEM_DEBUG_BREAK truncates the counts (bioem.cpp:518-525)
    S = orc.Setup(case["P"], case["model"], case["maps"], case["orient_lines"])
    of = None
    if case["orient_lines"]:
        of = str(tmp_path / "orient.txt")
        with open(of, "w") as f:
            f.write("%d\n" % len(case["orient_lines"]) + "\n".join(case["orient_lines"]) + "\n")
    H, ang, ref, par = hostlib.setup_from_files(os.path.join(case["dir"], "param.txt"), of)
    assert H.nAngles == S.nAngles and H.nCTF == S.nCTF and bool(H.isQuat) == bool(S.isQuat)
    assert np.array_equal(ang, S.angles)
    assert np.array_equal(ref, S.refCTF)
    assert np.array_equal(par, S.ctfParam)
    for f, _ in orc.ParamDevice._fields_:
        assert getattr(H.pd, f) == getattr(S.pd, f), f
    assert np.float32(H.voluang) == S.voluang
    assert (H.shiftX, H.shiftY) == (S.P["shiftX"], S.P["shiftY"])


def test_ctf_kernel_index_quirk():
    """for N=8 rows 0..7 at column j hold frequency indices 0,1,2,4,4,2,1,0."""
    from bioem_amd import hostlib
    ref, par, steps = hostlib.ctf_kernels(8, np.float32(1.5), (np.float32(0.1), np.float32(0.1), 1),
                                          (np.float32(3.0), np.float32(3.0), 1), (np.float32(50.), np.float32(50.), 1))
    col = ref[0, :, 1, 0]
    assert col[0] == col[7] and col[1] == col[6] and col[2] == col[5] and col[3] == col[4]
    assert ref[0, 0, 0, 0] == 1.0 and np.all(ref[..., 1] == 0)
    assert steps[0] == np.float32(0.1) and steps[1] == np.float32(3.0)   # one-point grid: step := start


def test_model_reader_text_and_pdb(tmp_path):
    from bioem_amd import hostlib
    case = load_case("g1_n48")
    p = str(tmp_path / "model.txt")
    iof.write_text_model(p, case["model"])
    pts, nd = hostlib.read_model(p)
    opts, ond = orc.model_from_array(case["model"])
    assert nd == ond and len(pts) == len(opts)
    assert np.array_equal(pts["pos"], opts["pos"]) and np.array_equal(pts["radius"], opts["radius"])
    # PDB: only ATOM/CA records count; residue tables give radius / electrons (model.cpp:738-844)
    pdb = str(tmp_path / "m.pdb")
    with open(pdb, "w") as f:
        f.write("HEADER    TEST\n")
        f.write("ATOM      1  N   GLY A   1      11.104   6.134  -6.504  1.00  0.00           N\n")
        f.write("ATOM      2  CA  GLY A   1      11.639   6.071  -5.147  1.00  0.00           C\n")
        f.write("ATOM      3  CA  TRP A   2      -1.500   2.250   3.125  1.00  0.00           C\n")
        f.write("HETATM    4  CA  CA  A   3       0.000   0.000   0.000  1.00  0.00          CA\n")
        f.write("END\n")
    pts, nd = hostlib.read_model(pdb, isPDB=True, nocentermass=True)
    assert len(pts) == 2 and nd == np.float32(148.0)
    assert pts["radius"][0] == np.float32(2.25) and pts["density"][1] == np.float32(108.0)
    assert np.allclose(pts["pos"][1], [-1.5, 2.25, 3.125])


def _write_mrc_volume(path, vol, big_endian=False, nsymbt=0):
    e = ">" if big_endian else "<"
    ns, nr, nc = vol.shape
    hdr = np.zeros(256, dtype=e + "i4")
    hdr[0:4] = [nc, nr, ns, 2]
    hdr[7:10] = [nc, nr, ns]
    hdr[23] = nsymbt
    raw = hdr.tobytes()
    raw = raw[:40] + struct.pack(e + "6f", 10., 10., 10., 90., 90., 90.) + raw[64:]
    with open(path, "wb") as f:
        f.write(raw + b"\0" * nsymbt + vol.astype(e + "f4").tobytes())


@pytest.mark.parametrize("big_endian", [False, True])
def test_model_reader_mrc_density_map(tmp_path, big_endian):
    """--ReadModelMRC (reference model.cpp:332-416): every voxel a point of radius 2*px at
    ((i - nx/2) px, (j - ny/2) px, (k - nz/2) px), i,j,k from 1, file order with i slowest."""
    from bioem_amd import hostlib
    rng = np.random.default_rng(3)
    nx, ny, nz = 4, 5, 6                       # header nc, nr, ns
    vals = rng.uniform(0.0, 2.0, size=nx * ny * nz).astype(np.float32)
    path = str(tmp_path / "vol.mrc")
    _write_mrc_volume(path, vals.reshape(nz, ny, nx), big_endian, nsymbt=16)   # raw order == file order
    px = np.float32(1.5)
    pts, nd = hostlib.read_model(path, isMRC=True, nocentermass=True, pixelSize=px)
    assert len(pts) == nx * ny * nz
    e = 0
    acc = np.float32(0)
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            for k in range(1, nz + 1):
                exp = np.array([(i - nx / 2.0) * float(px), (j - ny / 2.0) * float(px), (k - nz / 2.0) * float(px)],
                               dtype=np.float32)
                assert np.array_equal(pts["pos"][e], exp)
                assert pts["density"][e] == vals[e] and pts["radius"][e] == np.float32(2.0 * float(px))
                acc = np.float32(acc + vals[e])
                e += 1
    assert nd == acc


def test_particle_reader_text(tmp_path):
    from bioem_amd import hostlib
    case = load_case("g3_n32_trace")
    p = str(tmp_path / "particles.txt")
    iof.write_text_particles(p, case["maps"])
    maps = hostlib.read_particles(p, 32)
    assert maps.shape == case["maps"].shape and np.array_equal(maps, case["maps"])


@pytest.mark.parametrize("big_endian", [False, True])
def test_particle_reader_mrc(tmp_path, big_endian):
    """MRC mode 2: 1024-byte header + NSYMBT, endianness guess, transposed store, float z-score
    (reference map.cpp:663-845, include/mrc.h)."""
    from bioem_amd import hostlib
    N, ns, nsymbt = 16, 3, 80
    rng = np.random.default_rng(5)
    data = rng.normal(2.0, 3.0, size=(ns, N, N)).astype(np.float32)
    e = ">" if big_endian else "<"
    hdr = np.zeros(256, dtype=e + "i4")
    hdr[0:4] = [N, N, ns, 2]
    hdr[7:10] = [N, N, ns]
    hdr[23] = nsymbt
    hdr = hdr.tobytes()
    hdr = hdr[:40] + struct.pack(e + "6f", 10., 10., 10., 90., 90., 90.) + hdr[64:]
    path = str(tmp_path / "stack.mrc")
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * nsymbt + data.astype(e + "f4").tobytes())
    maps = hostlib.read_particles(path, N, mode=1)
    assert maps.shape == (ns, N, N)
    for s in range(ns):
        st = np.float32(0)
        st2 = np.float32(0)
        for v in data[s].ravel():
            st = np.float32(st + v)
            st2 = np.float32(st2 + v * v)
        st = np.float32(st / np.float32(N * N))
        sd = np.float32(np.sqrt(np.float32(st2 / np.float32(N * N) - st * st)))
        exp = (data[s].T / sd - st / sd).astype(np.float32)
        assert np.allclose(maps[s], exp, rtol=0, atol=1e-6)
    raw = hostlib.read_particles(path, N, mode=1, notnormmap=True)
    assert np.array_equal(raw[1], data[1].T)
    lst = str(tmp_path / "list.txt")
    with open(lst, "w") as f:
        f.write(path + "\n" + path + "\n")
    both = hostlib.read_particles(lst, N, mode=2)
    assert both.shape == (2 * ns, N, N) and np.array_equal(both[:ns], both[ns:])


def test_merge_host_equals_oracle_merge():
    from bioem_amd import engine
    case = load_case("g10_n64")
    S = oracle_setup(case)
    shards = []
    for (a, b) in [(0, 20), (20, 41), (41, 64)]:
        pm, _ = S.run(1, a, b)
        shards.append(pm)
    full, _ = S.run(1)
    om = orc.merge(shards)
    blocks = [np.frombuffer(s.tobytes(), dtype=np.uint8).copy() for s in shards]
    mine = engine.merge_host(blocks, S.nMaps, S.nAngles, 0).view(engine.PROB_MAP_DTYPE)
    for a, b, c in zip(mine, om, full):
        assert a["Total"] == b["Total"] and a["Constoadd"] == b["Constoadd"]
        assert (a["cent_x"], a["cent_y"], a["orient"], a["conv"]) == (b["cent_x"], b["cent_y"], b["orient"], b["conv"])
        # merged shards == unsharded run (associativity of the log-sum-exp fold)
        assert abs(S.final_logp(a) - S.final_logp(c)) < 1e-9 * abs(S.final_logp(c))
        assert (a["orient"], a["conv"], a["cent_x"], a["cent_y"]) == (c["orient"], c["conv"], c["cent_x"], c["cent_y"])


# ------------------------------------------------------------------------------------------------------
# readers against the fixtures the REFERENCE was run on (tests/golden/g24-g26: the model / particle files fed to
# oracle/_ref/bioEM_ref with --ReadPDB, --ReadModelMRC, --ReadMRC --ReadMultipleMRC; its Output_Probabilities are
# checked against the oracle on these points in test_oracle_golden.py and against the CLI in the GPU suite)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kind", [("g24_n32_pdb", dict(isPDB=True)), ("g25_n32_modelmrc", dict(isMRC=True))])
def test_model_readers_deliver_the_golden_points(name, kind):
    from bioem_amd import hostlib
    case = load_case(name)
    mfile = os.path.join(case["dir"], "model.pdb" if "isPDB" in kind else "model.mrc")
    pts, nd = hostlib.read_model(mfile, nocentermass=True, pixelSize=case["P"]["pixelSize"], **kind)
    want = case["model"]
    assert len(pts) == len(want)
    assert np.array_equal(pts["pos"], want[:, :3].astype(np.float32))
    assert np.array_equal(pts["radius"], want[:, 3].astype(np.float32))
    assert np.array_equal(pts["density"], want[:, 4].astype(np.float32))
    assert abs(float(nd) - want[:, 4].sum()) <= 1e-4 * want[:, 4].sum()


def test_multiple_mrc_reader_delivers_the_golden_maps(tmp_path):
    from bioem_amd import hostlib
    from golden_util import write_case_inputs
    case = load_case("g26_n32_multimrc")
    assert case["particles"] == "multimrc" and len(case["stacks"]) == 2
    write_case_inputs(case, tmp_path)
    cwd = os.getcwd()
    os.chdir(tmp_path)          # the list names its stacks relative to the run directory
    try:
        maps = hostlib.read_particles("list.txt", case["P"]["N"], mode=2)
    finally:
        os.chdir(cwd)
    assert maps.shape == case["maps"].shape
    assert np.allclose(maps, case["maps"], rtol=0, atol=2e-6)


def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent (which never touches a GPU) must
    start two child ranks with the launcher contract's variables and hand the failing rank's exit code through.  Here,
    without a GPU, both ranks stop at the device check -- which proves they were started as ranks 0 and 1 of a world of
    2; each rank's stderr is kept in its own file (bench_rank<k>.err)."""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["BIOEM_BENCH_LOGDIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU suite runs bench.py itself)")
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "exited with code 2" in r.stderr and "bench.py needs a GPU" in r.stderr
    # the rank that failed first wrote the message; the other either wrote it too or was terminated before it got there
    logs = [open(tmp_path / ("bench_rank%d.err" % k)).read() for k in (0, 1)]
    assert any("bench.py needs a GPU" in t for t in logs)
    assert all("WORLD_SIZE" not in t for t in logs)            # no rank saw a world size other than --gpus


def test_cpu_share_detection():
    """The CPU baseline runs one thread per CPU the cgroup grants (cpu.max quota / period), not per visible core."""
    lim = orc.cgroup_cpu_limit()
    assert lim is None or lim >= 1
    n = orc.usable_cpus(cap=1 << 20)
    assert 1 <= n <= (os.cpu_count() or 1) and (lim is None or n <= lim)
