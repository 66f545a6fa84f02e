"""One rank of tests/test_rccl_two_ranks.py (started as a fresh process per rank: nothing forks or execs from a
GPU-initialised process).  Rank r runs the HIP engine on ITS orientation block of a golden case on GPU `device`, then
the path's single exchange step through bioem_amd.dist_merge.merge_prob_maps (backend nccl = RCCL over xGMI; gloo for
the one-GPU rehearsal, where both ranks share device 0), and writes the merged block."""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    backend, name, outdir = sys.argv[1:4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import numpy as np
    import torch
    import torch.distributed as dist
    import bioem_amd.engine as eng
    import oracle as orc
    from bioem_amd.dist_merge import merge_prob_maps
    from golden_util import load_case, oracle_setup
    device = rank if backend == "nccl" else 0
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
    kw = dict(device_id=dev) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120), **kw)
    S = oracle_setup(load_case(name))
    K = int(S.pd.writeAngles)
    o0, o1 = rank * S.nAngles // world, (rank + 1) * S.nAngles // world        # bioem.cpp:748-753
    pd = eng.ParamDevice()
    for f, _ in eng.ParamDevice._fields_:
        setattr(pd, f, getattr(S.pd, f))
    E = eng.Engine(pd, S.nMaps, S.nAngles, S.nCTF, algo=1, device=device, shard=(o0, o1))
    E.upload_particle_maps(S.maps)
    E.upload_ctf(S.refCTF, S.ctfParam)
    E.upload_model(S.points, S.NormDen, S.px, S.P["shiftX"], S.P["shiftY"])
    E.upload_orientations(S.angles, S.isQuat)
    raw, pmap, _ = eng.new_prob_block(S.nMaps, 0, 0)
    E.start_run(raw)
    E.project_convolve_compare(o0, o1)
    E.finish_run(raw)
    cands = E.topk_angles(K, orc.logp_constant(S.pd)) if K else None
    merged = merge_prob_maps(pmap, dev, cands=cands)
    if K:
        merged, mc = merged
        np.save(os.path.join(outdir, "cands_%d.npy" % rank), mc)
    np.save(os.path.join(outdir, "merged_%d.npy" % rank), merged)
    dist.barrier()
    dist.destroy_process_group()
    E.close()


if __name__ == "__main__":
    main()
