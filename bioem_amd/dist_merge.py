"""Shard merge over torch.distributed for one-process-per-GPU runs (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The in-process counterpart (n GPUs, one process) is bioem_hip_merge in the C ABI.

Orientation blocks are independent; the only exchange step of the path is the merge of the per-particle posteriors at
the end of a run (reference: MPI merge, bioem.cpp:909-1044).  Every rank contributes its nMaps 40-byte map entries --
and, with WRITE_PROB_ANGLES, its K best orientations per particle (32 bytes each, selected on the device by
bioem_hip_topk_angles: each orientation has one owner, so the angle table itself needs no reduction) -- to ONE
all-gather; every rank then folds the gathered shards with the same rule as bioem_hip_merge_host:
    C*      = max_s Constoadd_s
    Total*  = sum_s Total_s * exp(Constoadd_s - C*)            (rank order)
    arg-max = the record of the LOWEST rank holding C* (lowest orientation block = serial first-maximum semantics;
              the reference's MPI path takes the highest rank, bioem.cpp:946-949)
    angles  = the K best of the union of the ranks' candidates (the writer's min-heap rule, bioem.cpp:1251-1286)
400 KB (+ 3.2 MB of candidates at K = 10) per rank at 10 000 particles: latency-bound, xGMI bandwidth irrelevant.
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import CANDIDATE_DTYPE, PROB_MAP_DTYPE, merge_host, merge_topk_host


def merge_prob_maps(pmap, device, orient_offset=0, cands=None, group=None):
    """pmap: this rank's numpy PROB_MAP_DTYPE array; orient_offset is added to its arg-max orientation indices (for
    engines that number their own block from 0; engines created as shards of the global list pass 0); cands: this
    rank's [nMaps, K] CANDIDATE_DTYPE array or None.  Returns the merged pmap (and the merged candidates if cands is
    given), identical on every rank."""
    world = dist.get_world_size(group)
    n = len(pmap)
    mine = np.array(pmap, dtype=PROB_MAP_DTYPE, copy=True)
    if orient_offset:
        mine["orient"] += np.int32(orient_offset)
    parts = [mine.view(np.uint8).reshape(-1)]
    K = 0
    if cands is not None:
        c = np.array(cands, dtype=CANDIDATE_DTYPE, copy=True)
        assert c.shape[0] == n
        K = c.shape[1]
        if orient_offset:
            c["orient"] = np.where(c["orient"] >= 0, c["orient"] + np.int32(orient_offset), c["orient"])
        parts.append(c.view(np.uint8).reshape(-1))
    payload = np.concatenate(parts)
    send = torch.from_numpy(payload).to(device)
    recv = torch.empty(world * payload.size, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)          # the exchange step
    got = recv.cpu().numpy().reshape(world, payload.size)
    map_bytes = n * PROB_MAP_DTYPE.itemsize
    blocks = [np.ascontiguousarray(got[r, :map_bytes]) for r in range(world)]
    merged = merge_host(blocks, n, 0, 0).view(PROB_MAP_DTYPE)
    if cands is None:
        return merged
    lists = [np.ascontiguousarray(got[r, map_bytes:]).view(CANDIDATE_DTYPE).reshape(n, K) for r in range(world)]
    return merged, merge_topk_host(lists)
