#!/usr/bin/env python3
"""bench.py -- image-comparisons/s of the BioEM compare path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (projection -> CTF convolution -> FFT cross-correlation over the
+-10 px displacement window -> log-sum-exp posterior) over this rank's orientation block against all
particles, followed by the shard merge.  Workload at N GPUs (weak scaling, config 2 per GPU):
  224^2 maps, 1 000 synthetic particles (replicated), 4 608 orientations PER GPU, 5 CTF envelopes,
  441 displacements; rank r owns orientation block r (the reference's MPI sharding, bioem.cpp:748-753),
  log-sum-exp merge of the per-particle posteriors by RCCL all-reduce (max, then sum).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events recorded on the engine's
stream around every comparison-kernel launch inside the timed region.  `cpu_baseline` (rank 0, N=1)
times the CPU oracle port on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(W, n_orient, n_threads):
    """CPU oracle port (oracle/bioem_oracle.c: full c2r FFT cross-correlation + calProb loop, OpenMP over
    particles like bioem.cpp:1392) on `n_orient` orientations x all CTFs x all particles of this workload."""
    import ctypes as C
    import oracle as orc
    L = orc.lib()
    L.orc_set_num_threads(n_threads)
    N, H = W.N, W.N // 2 + 1
    nP = W.nP
    pd = orc.ParamDevice()
    for f, _ in orc.ParamDevice._fields_:
        setattr(pd, f, getattr(W.pd, f))
    refFFT, sumRef, sumsqRef = W.engine.debug_particles()
    pts = np.zeros(len(W.points), dtype=orc.POINT_DTYPE)
    for k in ("pos", "radius", "density"):
        pts[k] = W.points[k]
    pmap = np.zeros(nP, dtype=orc.PROB_MAP_DTYPE)
    L.orc_init_prob(nP, W.nOrient, 0, pmap.ctypes.data, None)
    t0 = time.time()
    L.orc_run(C.byref(pd), 1, pts.ctypes.data, len(pts), W.NormDen, W.angles.ctypes.data, W.nOrient, 1, W.px, 0, 0,
              W.nCTF, W.refCTF.ctypes.data, W.ctfParam.ctypes.data, nP, refFFT.ctypes.data, sumRef.ctypes.data,
              sumsqRef.ctypes.data, 0, n_orient, pmap.ctypes.data, None)
    dt = time.time() - t0
    return n_orient * W.nCTF * nP / dt, dt, pmap, orc.logp_constant(pd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=1000)
    ap.add_argument("--orientations", type=int, default=4608, help="orientations per GPU")
    ap.add_argument("--pixels", type=int, default=224)
    ap.add_argument("--envelopes", type=int, default=5, help="CTF_B_ENV grid points")
    ap.add_argument("--defocus", type=int, default=1, help="CTF_DEFOCUS grid points (config 3: 2 x 5 envelopes = 10 CTFs)")
    ap.add_argument("--max-displacement", type=int, default=10, help="DISPLACE_CENTER half width (pixels)")
    ap.add_argument("--grid", type=int, default=1, help="DISPLACE_CENTER grid spacing")
    ap.add_argument("--write-angles", action="store_true", help="WRITE_PROB_ANGLES: keep the per-orientation table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-orientations", type=int, default=64, help="orientations of the CPU-baseline sample (~12 s)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the engine has no CPU path", file=sys.stderr)
        sys.exit(2)
    # rehearsal knobs (not used by the driver): BIOEM_BENCH_BACKEND=gloo runs the N>1 control flow with every
    # rank on ONE GPU (merge tensors on the CPU) -- RCCL itself refuses two ranks on the same device.
    backend = os.environ.get("BIOEM_BENCH_BACKEND", "nccl")
    gpu_index = local_rank if backend == "nccl" else int(os.environ.get("BIOEM_BENCH_DEVICE", "0"))
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from bioem_amd.dist_merge import merge_prob_maps
    from bioem_amd.engine import new_prob_block
    from bioem_amd.synthetic import Workload

    # every rank renders the same particle stack (same seeds); orientation seed differs per rank so that the
    # global list is the concatenation of `world` distinct blocks of `orientations` each.
    W = Workload(N=args.pixels, nP=args.particles, nOrient=args.orientations, device=gpu_index,
                 orient_seed=20260103 + rank, nEnv=args.envelopes, nDefocus=args.defocus, maxD=args.max_displacement, grid=args.grid,
                 write_angles=args.write_angles)
    E = W.engine
    nMaps = W.nP

    def one_step():
        raw, pmap, _ = new_prob_block(nMaps, W.nOrient, int(args.write_angles))
        E.start_run(raw)
        E.project_convolve_compare(0, W.nOrient)
        E.finish_run(raw)
        if world > 1:
            # the path's single exchange step: log-sum-exp merge of bioem.cpp:909-994 over RCCL
            return merge_prob_maps(pmap, dev)
        return pmap

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    E.reset_kernel_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    dt = time.perf_counter() - t0
    kms, launches, ncomp = E.kernel_stats()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # BASELINE.json's metric string (value = the comparisons/s half; the log P half is the `parity` object)
    metric_name = "image-comparisons/sec (orient x CTF x particle) at %d^2" % W.N
    try:
        if W.N == 224:
            with open(os.path.join(ROOT, "BASELINE.json")) as f:
                metric_name = json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        pass
    shape = (W.N, W.nP, W.nOrient, W.nCTF, args.max_displacement, args.grid)
    workload_name = {(224, 1000, 4608, 5, 10, 1): "BASELINE config 2 per GPU",
                     (224, 10000, 4608, 10, 10, 1): "BASELINE config 3, one GPU's share of 8"}.get(shape, "custom")
    total_comparisons = world * W.comparisons_per_pass * args.steps
    value = total_comparisons / dt
    b_alg = 8 * W.N * (W.N // 2 + 1)  # bytes: one read of the particle half-spectrum per comparison
    achieved = (ncomp * b_alg / 1e9) / (kms / 1e3) if kms > 0 else 0.0

    out = {
        "metric": metric_name,
        "value": value,
        "unit": "comparisons/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: %d^2 maps, %d particles, %d orientations/GPU, %d CTFs, +-%d px grid %d (%d "
                               "displacements)" % (workload_name, W.N, W.nP, W.nOrient, W.nCTF,
                                                   args.max_displacement, args.grid, int(W.pd.NtotDisp)),
                   "pixels": W.N, "particles": W.nP, "orientations_per_gpu": W.nOrient, "ctf": W.nCTF,
                   "displacements": int(W.pd.NtotDisp), "orientation_list": "seeded uniform random quaternions",
                   "fast_path": bool(E.fast_path), "parallelism": "orientation blocks x%d, RCCL log-sum-exp merge"
                   % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": E.kernel_name, "launches": int(launches),
                     "avg_launch_ms": (kms / launches) if launches else None,
                     "alg_bytes_per_comparison": b_alg,
                     "comparisons_per_launch": (ncomp / launches) if launches else None},
    }
    # HBM traffic of the same kernel on the same per-launch workload from the committed rocprofv3 PMC passes
    # (profiles/pmc_traffic.json, collected in separate --pmc runs; cannot be sampled inside this process)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if launches and abs(ncomp / launches - pmc["comparisons_per_launch"]) < 1 and W.N == 224:
            out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"] / 1e9 / (kms / launches / 1e3)
            out["roofline"]["traffic_bytes_per_launch"] = pmc["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/%s_pmc_summary.json" % pmc["tag"]
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        nthreads = orc.usable_cpus(cap=16)  # the GPU box's CPU share for one GPU
        nco = min(args.cpu_orientations, W.nOrient)
        v, secs, want, const = cpu_baseline(W, nco, nthreads)
        out["cpu_baseline"] = {"value": v, "unit": "comparisons/s", "cores": nthreads, "kind": "port",
                               "sample": "%d orientations x %d CTF x %d particles of the same workload (%.1f s)"
                               % (nco, W.nCTF, W.nP, secs)}
        # the checker's second job: the HIP path on the SAME sample against the oracle (untimed)
        raw, got, _ = new_prob_block(nMaps, W.nOrient, int(args.write_angles))
        E.start_run(raw)
        E.project_convolve_compare(0, nco)
        E.finish_run(raw)
        la = np.log(got["Total"]) + got["Constoadd"] + const
        lb = np.log(want["Total"]) + want["Constoadd"] + const
        same = ((got["orient"] == want["orient"]) & (got["conv"] == want["conv"]) &
                (got["cent_x"] == want["cent_x"]) & (got["cent_y"] == want["cent_y"]))
        out["parity"] = {"against": "CPU oracle on the cpu_baseline sample", "particles": int(nMaps),
                         "max_abs_dlogp": float(np.abs(la - lb).max()),
                         "max_rel_dlogp": float((np.abs(la - lb) / np.abs(lb)).max()),
                         "argmax_mismatches": int((~same).sum()), "tolerance_rel": 1e-4}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
