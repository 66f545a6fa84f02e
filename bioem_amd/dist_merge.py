"""Shard merge over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Orientation blocks are independent; the only exchange step of the path is the log-sum-exp merge of the
per-particle posteriors at the end of a run (reference: MPI merge, bioem.cpp:909-1044):
    C*      = all_reduce_max(Constoadd)
    Total*  = all_reduce_sum(Total * exp(Constoadd - C*))
    arg-max = the entry of the LOWEST rank holding C* (lowest orientation block = serial first-maximum
              semantics; the reference's MPI path takes the highest rank, bioem.cpp:946-949)
Three small collectives (8 B, 8 B and 32 B per particle): latency-bound, no data-path collective.
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import PROB_MAP_DTYPE


def merge_prob_maps(pmap, device, group=None):
    """pmap: this rank's numpy PROB_MAP_DTYPE array.  Returns the merged array (identical on every rank)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(pmap)
    const = torch.from_numpy(np.ascontiguousarray(pmap["Constoadd"])).to(device)
    total = torch.from_numpy(np.ascontiguousarray(pmap["Total"])).to(device)
    cmax = const.clone()
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX, group=group)
    tsum = total * torch.exp(const - cmax)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=group)
    owner = torch.where(const >= cmax, torch.full((n,), rank, dtype=torch.int64, device=device),
                        torch.full((n,), world, dtype=torch.int64, device=device))
    dist.all_reduce(owner, op=dist.ReduceOp.MIN, group=group)
    # ship the 24-byte arg-max records of the owners: masked sum (exactly one contributor per particle)
    rec = np.zeros((n, 6), dtype=np.int32)
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = pmap["cent_x"], pmap["cent_y"], pmap["orient"], pmap["conv"]
    rec[:, 4] = pmap["norm"].view(np.int32)
    rec[:, 5] = pmap["mu"].view(np.int32)
    trec = torch.from_numpy(rec).to(device).to(torch.int64)
    mine = (owner == rank).unsqueeze(1)
    trec = torch.where(mine, trec, torch.zeros_like(trec))
    dist.all_reduce(trec, op=dist.ReduceOp.SUM, group=group)
    out = np.zeros(n, dtype=PROB_MAP_DTYPE)
    out["Total"] = tsum.cpu().numpy()
    out["Constoadd"] = cmax.cpu().numpy()
    r = trec.cpu().numpy().astype(np.int32)
    out["cent_x"], out["cent_y"], out["orient"], out["conv"] = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    out["norm"] = r[:, 4].copy().view(np.float32)
    out["mu"] = r[:, 5].copy().view(np.float32)
    return out
