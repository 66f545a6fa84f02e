"""world_size-2 gloo test (CPU) of the N>1 path: orientation blocks per rank + log-sum-exp merge over
torch.distributed, against the oracle's unsharded run."""
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
    merged = merge_prob_maps(pm, torch.device("cpu"))
    np.save(os.path.join(outdir, "merged_%d.npy" % rank), merged)
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
