"""world_size-2 gloo tests (CPU) of the N>1 path: orientation blocks per rank + one all-gather + log-sum-exp merge over
torch.distributed, against the oracle's unsharded run -- the map entries, the arg-max orientation of ranks that number
their block from 0 (orient_offset), and with WRITE_PROB_ANGLES the K best orientations per particle."""
import heapq
import math
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    from bioem_amd.dist_merge import merge_prob_maps
    from golden_util import load_case, oracle_setup
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = load_case("g10_n64")
    S = oracle_setup(case)
    nA = S.nAngles
    o0 = rank * nA // world            # the reference's block partition, bioem.cpp:748-753
    o1 = (rank + 1) * nA // world
    pm, _ = S.run(1, o0, o1)           # this rank's shard (stands in for the HIP engine on CPU)
    # the rank numbers its own block from 0 (as an engine that only knows its own orientations would)
    local = pm.copy()
    local["orient"] -= o0
    merged = merge_prob_maps(local, torch.device("cpu"), orient_offset=o0)
    np.save(os.path.join(outdir, "merged_%d.npy" % rank), merged)
    dist.barrier()
    dist.destroy_process_group()


def _block_candidates(pang, o0, o1, K, numconst):
    """What bioem_hip_topk_angles returns for the block [o0, o1): the writer's heap rule (bioem.cpp:1251-1286)."""
    from bioem_amd.engine import CANDIDATE_DTYPE
    nMaps = pang.shape[1]
    out = np.zeros((nMaps, K), dtype=CANDIDATE_DTYPE)
    out["orient"] = -1
    out["logp"] = -np.inf
    for m in range(nMaps):
        q = []
        for io in range(o0, o1):
            pa = pang[io, m]
            logp = (math.log(pa["forAngles"]) if pa["forAngles"] > 0 else -math.inf) + pa["ConstAngle"] + numconst
            if len(q) < K:
                heapq.heappush(q, (logp, io))
            elif q[0][0] < logp:
                heapq.heapreplace(q, (logp, io))
        for k, (lp, io) in enumerate(sorted(q, reverse=True)):
            out[m, k] = (pang[io, m]["forAngles"], pang[io, m]["ConstAngle"], lp, io, 0)
    return out


def _worker_angles(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import oracle as orc
    import torch
    import torch.distributed as dist
    from bioem_amd.dist_merge import merge_prob_maps
    from golden_util import load_case, oracle_setup
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = oracle_setup(load_case("g4_n32_angles"))
    nA, K = S.nAngles, S.pd.writeAngles
    o0, o1 = rank * nA // world, (rank + 1) * nA // world
    pm, pang = S.run(1, o0, o1)
    cands = _block_candidates(pang, o0, o1, K, orc.logp_constant(S.pd))
    merged, mc = merge_prob_maps(pm, torch.device("cpu"), cands=cands)
    np.save(os.path.join(outdir, "merged_%d.npy" % rank), merged)
    np.save(os.path.join(outdir, "cands_%d.npy" % rank), mc)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_matches_unsharded(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from golden_util import load_case, oracle_setup
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0 = np.load(tmp_path / "merged_0.npy")
    m1 = np.load(tmp_path / "merged_1.npy")
    assert m0.tobytes() == m1.tobytes()          # every rank holds the same merged posterior
    S = oracle_setup(load_case("g10_n64"))
    full, _ = S.run(1)
    for a, c in zip(m0, full):
        la, lc = S.final_logp(a), S.final_logp(c)
        assert abs(la - lc) <= 1e-9 * abs(lc)
        assert (a["orient"], a["conv"], a["cent_x"], a["cent_y"]) == (c["orient"], c["conv"], c["cent_x"], c["cent_y"])
        assert a["norm"] == c["norm"] and a["mu"] == c["mu"]


def test_two_rank_merge_of_the_k_best_orientations(tmp_path):
    """WRITE_PROB_ANGLES across ranks: every orientation has one owner, so each rank ships its K best per particle and
    the K best of the union must be the K best of the unsharded table (ANG_PROB of the reference, g4)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    from golden_util import load_case, oracle_setup
    port = _free_port()
    mp.spawn(_worker_angles, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0 = np.load(tmp_path / "cands_0.npy")
    c1 = np.load(tmp_path / "cands_1.npy")
    assert c0.tobytes() == c1.tobytes()
    assert np.load(tmp_path / "merged_0.npy").tobytes() == np.load(tmp_path / "merged_1.npy").tobytes()
    S = oracle_setup(load_case("g4_n32_angles"))
    pm, pang = S.run(1)
    rows = orc.ang_prob_rows(S, pm, pang)
    for m in range(S.nMaps):
        assert [r["orient"] for r in rows[m]] == [int(v) for v in c0[m]["orient"]]
        for r, c in zip(rows[m], c0[m]):
            assert abs(r["logp"] - c["logp"]) <= 1e-9 * abs(r["logp"])
            assert abs(r["logsum"] - math.log(c["forAngles"])) <= 1e-9 and r["const"] == c["ConstAngle"]
