"""The N > 1 path with real processes: two ranks, one HIP engine each, ONE all-gather + fold (reference: MPI merge,
bioem.cpp:909-1044; block partition :748-753).

* backend nccl (= RCCL over xGMI): needs two GPUs -- skipped on the one-GPU boxes of this pool, runs wherever a box
  has two (torch.cuda.device_count() does not initialise the GPU);
* backend gloo: the same worker with both ranks on GPU 0 and the merge tensors on the CPU -- runs on every GPU box, so
  the worker itself is exercised even where RCCL cannot take two ranks;
* bioem_hip_merge (C ABI, one process, n handles on n GPUs: ncclCommInitAll + ncclAllGather + k_merge_shards)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc
from golden_util import load_case, oracle_setup

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_gpus():
    import torch
    return torch.cuda.device_count()


def run_ranks(backend, name, outdir, world=2):
    """Starts the ranks and watches ALL of them (as bench.self_launch does): a rank that dies -- say at handle creation --
    would otherwise leave the survivor in the rendezvous or the all-gather until its own time-out before the failure is
    reported.  Every rank's output goes to a file (a pipe nobody drains while polling could fill up)."""
    import tempfile
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, logs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        logs.append(tempfile.NamedTemporaryFile(prefix="rccl_rank%d_" % r, suffix=".log", delete=False))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), backend, name,
                                       str(outdir)], env=env, stdout=logs[-1], stderr=subprocess.STDOUT))
    deadline = time.time() + 300
    failed = None
    while failed is None and time.time() < deadline:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
        elif all(c == 0 for c in codes):
            break
        else:
            time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            p.terminate()          # the exact children this test started, by handle
    for p in procs:
        try:
            p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    outs = []
    for f in logs:
        f.close()
        with open(f.name, errors="replace") as g:
            outs.append(g.read())
        os.unlink(f.name)
    assert failed is None, "rank %d failed (the other ranks were terminated):\n%s" % (failed, outs[failed][-3000:])
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d (timed out?):\n%s" % (r, outs[r][-3000:])


def check_against_unsharded(name, outdir, world=2):
    S = oracle_setup(load_case(name))
    merged = [np.load(os.path.join(outdir, "merged_%d.npy" % r)) for r in range(world)]
    for m in merged[1:]:
        assert m.tobytes() == merged[0].tobytes()              # every rank holds the same merged posterior
    full, pang = S.run(1)
    for a, c in zip(merged[0], full):
        la, lc = S.final_logp(a), S.final_logp(c)
        assert abs(la - lc) <= 1e-4 * abs(lc) and abs(la - lc) <= 2e-2
        assert (a["orient"], a["conv"], a["cent_x"], a["cent_y"]) == (c["orient"], c["conv"], c["cent_x"], c["cent_y"])
    if S.pd.writeAngles:
        c0 = np.load(os.path.join(outdir, "cands_0.npy"))
        assert c0.tobytes() == np.load(os.path.join(outdir, "cands_1.npy")).tobytes()
        rows = orc.ang_prob_rows(S, full, pang)
        for m in range(S.nMaps):
            assert [r["orient"] for r in rows[m]] == [int(v) for v in c0[m]["orient"]]
            for r, c in zip(rows[m], c0[m]):
                assert abs(r["logp"] - c["logp"]) <= 5e-3


@pytest.mark.parametrize("name", ["g10_n64", "g4_n32_angles"])
def test_two_process_ranks_share_one_gpu_over_gloo(name, tmp_path):
    run_ranks("gloo", name, tmp_path)
    check_against_unsharded(name, tmp_path)


@pytest.mark.skipif(n_gpus() < 2, reason="two real RCCL ranks need two GPUs")
@pytest.mark.parametrize("name", ["g10_n64", "g4_n32_angles"])
def test_two_rccl_ranks_over_xgmi(name, tmp_path):
    run_ranks("nccl", name, tmp_path)
    check_against_unsharded(name, tmp_path)


@pytest.mark.skipif(n_gpus() < 2, reason="bioem_hip_merge over RCCL needs one GPU per handle")
@pytest.mark.parametrize("name", ["g10_n64", "g4_n32_angles"])
def test_c_abi_merge_with_two_handles_on_two_gpus(name):
    import bioem_amd.engine as eng
    S = oracle_setup(load_case(name))
    K = int(S.pd.writeAngles)
    pd = eng.ParamDevice()
    for f, _ in eng.ParamDevice._fields_:
        setattr(pd, f, getattr(S.pd, f))
    engines = []
    for g in range(2):
        o0, o1 = g * S.nAngles // 2, (g + 1) * S.nAngles // 2
        E = eng.Engine(pd, S.nMaps, S.nAngles, S.nCTF, algo=1, device=g, shard=(o0, o1))
        E.upload_particle_maps(S.maps)
        E.upload_ctf(S.refCTF, S.ctfParam)
        E.upload_model(S.points, S.NormDen, S.px, S.P["shiftX"], S.P["shiftY"])
        E.upload_orientations(S.angles, S.isQuat)
        raw, _, _ = eng.new_prob_block(S.nMaps, 0, 0)
        E.start_run(raw)
        E.project_convolve_compare(o0, o1)
        E.finish_run(raw)
        engines.append(E)
    numconst = orc.logp_constant(S.pd)
    for _ in range(2):                                     # second call: the cached communicator
        merged, cand = eng.merge_rccl(engines, K, numconst)
        full, pang = S.run(1)
        for a, c in zip(merged, full):
            la, lc = S.final_logp(a), S.final_logp(c)
            assert abs(la - lc) <= 1e-4 * abs(lc) and abs(la - lc) <= 2e-2
            assert (a["orient"], a["conv"], a["cent_x"], a["cent_y"]) == (c["orient"], c["conv"], c["cent_x"], c["cent_y"])
        if K:
            rows = orc.ang_prob_rows(S, full, pang)
            for m in range(S.nMaps):
                assert [r["orient"] for r in rows[m]] == [int(v) for v in cand[m]["orient"]]
    for E in engines:
        E.close()
