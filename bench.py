#!/usr/bin/env python3
"""bench.py -- image-comparisons/s of the BioEM compare path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (projection -> CTF convolution -> FFT cross-correlation over the
+-10 px displacement window -> log-sum-exp posterior) over this rank's orientation block against all
particles, followed by the shard merge.  Workload (weak scaling: 4 608 orientations PER GPU, rank r owns orientation
block r of the global list = the reference's MPI sharding, bioem.cpp:748-753; the merge of the per-particle posteriors
is ONE all-gather over RCCL + a local fold):
  --config 2 (default below 8 GPUs)  BASELINE config 2 per GPU: 224^2, 1 000 particles, 5 CTF envelopes, 441 displacements
  --config 3 (default at 8 GPUs)     BASELINE config 3: 224^2, 10 000 particles, 2 x 5 CTFs, 8 x 4 608 = 36 864 orientations
Explicit --particles/--envelopes/... override either.

Launch: `python bench.py --gpus N` starts N ranks itself when no launcher set WORLD_SIZE (the parent makes no GPU call
and only forwards rank 0's line); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the
ranks come from the launcher.

Prints ONE JSON line (rank 0).  `roofline` names the resource that bounds the comparison kernel -- VALU issue -- and is
measured live: HIP events recorded on the engine's stream around every comparison-kernel launch of the timed region
give the launch duration; the wave-instructions per comparison of exactly this kernel instantiation come from the
committed rocprofv3 counter passes (profiles/pmc_current.json, scripts/profile_round.sh + pmc_summary.py), refused when
kernel or shape differ.  `hbm` holds the memory side: the north-star's algorithmic bytes, the compulsory bytes and the
counter-measured fabric traffic.  `cpu_baseline` (rank 0, N=1) times the CPU oracle port on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VECTOR_PEAK_TFLOPS = 157.3     # same guide: 256 CUs x 4 SIMDs x 64 lanes x 2 flop (FMA) x 2.4 GHz
POSTERIOR_FLOPS_PER_DISPLACEMENT = 230   # ~40 flop + 2 log + 1-2 exp in FP64 per displacement (~0.1 MFLOP / 441)
VALU_PEAK_GINSTR = 1024 * 2.4 / 2   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz (same guide)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("auto", "2", "3"), default="auto",
                    help="BASELINE.json configuration: 2 = 1 000 particles x 5 CTFs per GPU, 3 = 10 000 particles x 10 "
                         "CTFs (the 8-GPU configuration); auto = 3 at --gpus 8, else 2")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--orientations", type=int, default=4608, help="orientations per GPU")
    ap.add_argument("--pixels", type=int, default=224)
    ap.add_argument("--envelopes", type=int, default=5, help="CTF_B_ENV grid points")
    ap.add_argument("--defocus", type=int, default=None, help="CTF_DEFOCUS grid points (config 3: 2 x 5 envelopes = 10 CTFs)")
    ap.add_argument("--max-displacement", type=int, default=10, help="DISPLACE_CENTER half width (pixels)")
    ap.add_argument("--grid", type=int, default=1, help="DISPLACE_CENTER grid spacing")
    ap.add_argument("--algo", type=int, choices=(1, 2), default=1,
                    help="BIOEM_ALGO: 1 = doRefMapFFT (bioem_algorithm.h:144-198, the reference's default), 2 = "
                         "doRefMap_CPU_Parallel / _Reduce (bioem.cpp:1461-1602); engine and CPU oracle alike")
    ap.add_argument("--write-angles", type=int, nargs="?", const=10, default=0, metavar="K",
                    help="WRITE_PROB_ANGLES K: keep the per-orientation table (on the device, sharded) and select the K "
                         "best orientations per particle")
    ap.add_argument("--direct", action="store_true",
                    help="BASELINE config 4: cross-correlation as a sliding window in real space (BIOEM_CC_DIRECT=1; "
                         "images up to 160 pixels, e.g. --pixels 128) instead of the transform path")
    ap.add_argument("--max-seconds", type=float, default=480.0,
                    help="wall-clock budget of the whole run (set-up + warm-up + timed steps + merge): after the first "
                         "warm-up step the warm-up and step counts are trimmed, on every rank alike, so that the run "
                         "ends inside it; the line reports steps_requested / steps (= run) and warmup_requested / warmup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-orientations", type=int, default=64, help="orientations of the CPU-baseline sample")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    args = ap.parse_args()
    cfg = args.config if args.config != "auto" else ("3" if args.gpus == 8 else "2")
    if args.particles is None:
        args.particles = 10000 if cfg == "3" else 1000
    if args.defocus is None:
        args.defocus = 2 if cfg == "3" else 1
    return args


def log_dir():
    d = os.environ.get("BIOEM_BENCH_LOGDIR") or os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        d = os.getcwd()
    return d


def self_launch(n, child_argv, poll_s=0.2, out=None):
    """No launcher in the environment: start the n ranks as fresh child processes (this parent never touches the GPU;
    nothing is exec'ed from a GPU-initialised process), watch ALL of them, and hand rank 0's output through.  Every
    rank's stderr goes to <log dir>/bench_rank<k>.err.  On the first non-zero exit the other ranks are terminated (they
    would otherwise sit in the rendezvous or in the all-gather until the backend's timeout), the failed rank's
    message is printed and its exit code returned."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = log_dir()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   BIOEM_BENCH_LOGDIR=d, BIOEM_BENCH_SELF_LAUNCHED="1")
        errs.append(open(os.path.join(d, "bench_rank%d.err" % r), "wb"))
        procs.append(subprocess.Popen(child_argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=errs[-1]))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        for r, c in enumerate(codes):
            if c not in (None, 0):
                failed = (r, c)
                break
        if failed is None and all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()            # the exact child this parent started, by handle
                p.wait()
    reader.join(timeout=5.0)
    for f in errs:
        f.close()
    out = out or sys.stdout
    out.write(b"".join(c for c in chunks if c).decode(errors="replace"))
    out.flush()
    if failed is not None:
        r, c = failed
        path = os.path.join(d, "bench_rank%d.err" % r)
        try:
            with open(path, "rb") as f:
                tail = f.read()[-4000:].decode(errors="replace")
        except OSError:
            tail = ""
        sys.stderr.write("bench.py: rank %d of %d exited with code %d; the other ranks were terminated.  Its stderr "
                         "(%s):\n%s\n" % (r, n, c, path, tail))
        return c
    return 0


def plan_steps(steps, warmup_left, step_s, elapsed_s, max_seconds, tail_s):
    """Time budget (--max-seconds): how many of the remaining `warmup_left` warm-up steps and of the `steps` timed steps
    still fit, given that one step took `step_s` and `elapsed_s` of the budget are gone; `tail_s` is kept for what
    follows the timed region (merge, CPU baseline, parity leg).  Warm-up goes first, then timed steps; at least one timed
    step always runs.  Pure function: every rank calls it with the same (max-reduced) numbers."""
    if max_seconds is None or max_seconds <= 0 or step_s <= 0:
        return warmup_left, steps
    fit = int(max(0.0, max_seconds - elapsed_s - tail_s) / step_s)
    fit = max(1, fit)
    w = min(warmup_left, max(0, fit - steps))
    return w, min(steps, max(1, fit - w))


def shared_stack_path(shape):
    """One particle stack per NODE for the N > 1 runs: rank 0 renders and leaves it in /dev/shm, the other ranks map it
    (config 3: 2 GB once instead of 2 GB and a rendering pass per rank)."""
    tag = "_".join(str(int(v)) for v in shape)
    return "/dev/shm/bioem_bench_stack_%s_%s.npy" % (os.environ.get("MASTER_PORT", "0"), tag)


def wait_for_file(path, timeout_s, poll_s=0.2):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
            return False
        time.sleep(poll_s)
    return True


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(W, n_orient, n_threads, repeats, algo=1):
    """CPU oracle port (oracle/bioem_oracle.c: full c2r FFT cross-correlation + calProb loop, OpenMP over
    particles like bioem.cpp:1392) on `n_orient` orientations x all CTFs x all particles of this workload;
    median of `repeats` runs."""
    import ctypes as C
    import numpy as np
    import oracle as orc
    L = orc.lib()
    L.orc_set_num_threads(n_threads)
    nP = W.nP
    pd = orc.ParamDevice()
    for f, _ in orc.ParamDevice._fields_:
        setattr(pd, f, getattr(W.pd, f))
    # the oracle's OWN particle inputs (r2c + sequential float sums of the real-space stack, map.cpp:557-630,
    # bioem.cpp:2087-2107): the device particle transform must not sit on both sides of the parity object
    refFFT = np.stack([orc.fft2_r2c(m) for m in W.maps])
    sums = [orc.map_sums(m) for m in W.maps]
    sumRef = np.array([a for a, _ in sums], dtype=np.float32)
    sumsqRef = np.array([b for _, b in sums], dtype=np.float32)
    dspec, dsum, dsumsq = W.engine.debug_particles()
    particle_check = {"sums_bitwise_equal": bool(np.array_equal(dsum, sumRef) and np.array_equal(dsumsq, sumsqRef)),
                      "spectra_max_rel_diff": float(np.abs(dspec - refFFT).max() / np.abs(refFFT).max())}
    del dspec
    pts = np.zeros(len(W.points), dtype=orc.POINT_DTYPE)
    for k in ("pos", "radius", "density"):
        pts[k] = W.points[k]
    times = []
    for _ in range(repeats):
        pmap = np.zeros(nP, dtype=orc.PROB_MAP_DTYPE)
        L.orc_init_prob(nP, len(W.angles), 0, pmap.ctypes.data, None)
        t0 = time.time()
        L.orc_run(C.byref(pd), algo, pts.ctypes.data, len(pts), W.NormDen, W.angles.ctypes.data, len(W.angles), 1, W.px, 0,
                  0, W.nCTF, W.refCTF.ctypes.data, W.ctfParam.ctypes.data, nP, refFFT.ctypes.data, sumRef.ctypes.data,
                  sumsqRef.ctypes.data, 0, n_orient, pmap.ctypes.data, None)
        times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    return n_orient * W.nCTF * nP / dt, times, pmap, orc.logp_constant(pd), particle_check


def main():
    args = parse_args()
    if args.direct:
        os.environ["BIOEM_CC_DIRECT"] = "1"  # read by bioem_hip_create; inherited by self-launched ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    t_start = time.perf_counter()
    if os.environ.get("BIOEM_BENCH_INJECT_FAILURE") == os.environ.get("RANK", "0"):
        raise RuntimeError("injected failure on rank %s (BIOEM_BENCH_INJECT_FAILURE)" % os.environ.get("RANK", "0"))
    import numpy as np
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
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that never arrives must not hold the others for the backend's default half hour
        rdv = datetime.timedelta(seconds=float(os.environ.get("BIOEM_BENCH_RENDEZVOUS_S", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=rdv)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=rdv)

    from bioem_amd.dist_merge import merge_prob_maps
    from bioem_amd.engine import new_prob_block
    from bioem_amd.synthetic import Workload

    # the global orientation list is `world` blocks of `orientations`; rank r owns block r; every rank renders the
    # same particle stack (from block 0, same seeds), i.e. the N = 1 workload's stack
    K = int(args.write_angles)
    wl = dict(N=args.pixels, nP=args.particles, nOrient=args.orientations, device=gpu_index, nEnv=args.envelopes,
              nDefocus=args.defocus, maxD=args.max_displacement, grid=args.grid, write_angles=K, blocks=world, block=rank,
              algo=args.algo)
    stack_file = None
    if world > 1 and os.access("/dev/shm", os.W_OK) and not os.environ.get("BIOEM_BENCH_RENDER_PER_RANK"):
        stack_file = shared_stack_path((args.pixels, args.particles, args.orientations, args.envelopes, args.defocus))
    if stack_file is None:
        W = Workload(**wl)
        stack_source = "rendered by this rank"
    elif rank == 0:
        W = Workload(**wl)
        np.save(stack_file + ".tmp.npy", W.maps)
        os.replace(stack_file + ".tmp.npy", stack_file)     # atomic: a waiting rank never sees a partial file
        stack_source = "rendered by rank 0, shared through %s" % stack_file
    else:
        W = Workload(render=False, **wl)
        if not wait_for_file(stack_file, float(os.environ.get("BIOEM_BENCH_RENDER_WAIT_S", "600"))):
            raise RuntimeError("rank %d: the particle stack %s of rank 0 did not appear" % (rank, stack_file))
        W.maps = np.load(stack_file, mmap_mode="r")
        W.engine.upload_particle_maps(W.maps)
        stack_source = "mapped from %s" % stack_file
    E = W.engine
    nMaps = W.nP
    numconst = 0.0
    if K:
        import math
        numconst = 0.5 * math.log(math.pi) + (1 - float(W.pd.Ntotpi) * 0.5) * (math.log(2 * math.pi) + 1) + math.log(
            float(W.pd.volu))
    shard_engine = E.shard is not None

    def one_step():
        raw, pmap, _ = new_prob_block(nMaps, 0 if shard_engine else W.nOrient, 0)
        E.start_run(raw)
        E.project_convolve_compare(W.o0, W.o1)
        E.finish_run(raw)
        cands = E.topk_angles(K, numconst) if K else None      # K x nMaps x 32 B instead of the table
        if world > 1:
            # the path's single exchange step (reference: MPI merge, bioem.cpp:909-1044): one all-gather + fold
            tm = time.perf_counter()
            merged = merge_prob_maps(pmap, dev, cands=cands)
            merge_s[0] += time.perf_counter() - tm
            return merged
        return (pmap, cands) if K else pmap

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    merge_s = [0.0]
    comm_setup_s = 0.0
    if world > 1:
        # communicator creation (RCCL builds its rings on the first collective) stays outside the timed region even
        # with --warmup 0: one tiny all-gather + barrier here
        tc = time.perf_counter()
        probe = torch.zeros(8, dtype=torch.uint8, device=dev)
        sink = torch.empty(8 * world, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(sink, probe)
        dist.barrier()
        if backend == "nccl":
            torch.cuda.synchronize()
        comm_setup_s = time.perf_counter() - tc
    def rank_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # time budget (--max-seconds): the first executed step prices a step; warm-up, then timed steps are trimmed to fit
    tail_s = 90.0 if (world == 1 and not args.no_cpu_baseline) else 10.0
    warmup_run, steps_run = args.warmup, args.steps
    if args.warmup > 0:
        tp = time.perf_counter()
        one_step()
        sync()
        step_s = rank_max(time.perf_counter() - tp)
        w_extra, steps_run = plan_steps(args.steps, args.warmup - 1, step_s, rank_max(time.perf_counter() - t_start),
                                        args.max_seconds, tail_s)
        for _ in range(w_extra):
            one_step()
        warmup_run = 1 + w_extra
    sync()
    E.reset_kernel_stats()
    merge_s[0] = 0.0
    setup_s = time.perf_counter() - t_start
    t0 = time.perf_counter()
    done = 0
    while done < steps_run:
        last = one_step()
        done += 1
        if args.warmup == 0 and done == 1 and steps_run > 1:
            sync()          # no warm-up step priced a step: the first timed one does (one extra barrier in the region)
            _, more = plan_steps(args.steps - 1, 0, rank_max(time.perf_counter() - t0),
                                 rank_max(time.perf_counter() - t_start), args.max_seconds, tail_s)
            steps_run = 1 + more
    sync()
    dt = time.perf_counter() - t0
    steps_requested, warmup_requested = args.steps, args.warmup
    args.steps, args.warmup = steps_run, warmup_run
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
                     (224, 10000, 4608, 10, 10, 1): "BASELINE config 3 (10 000 particles, 36 864 orientations, 10 CTFs)"
                     if world == 8 else "BASELINE config 3, one GPU's share of 8"}.get(shape, "custom")
    total_comparisons = world * W.comparisons_per_pass * args.steps
    value = total_comparisons / dt
    b_alg = 8 * W.N * (W.N // 2 + 1)  # bytes: one read of the particle half-spectrum per comparison
    avg_ms = (kms / launches) if launches else None
    cpl = (ncomp / launches) if launches else None
    sig = E.kernel_signature

    out = {
        "metric": metric_name,
        "value": value,
        "unit": "comparisons/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        # --max-seconds: what was asked for; "steps" / "warmup" above are what ran (equal unless the budget trimmed them)
        "steps_requested": steps_requested,
        "warmup_requested": warmup_requested,
        "max_seconds": args.max_seconds,
        "ms_per_step": 1e3 * dt / args.steps,
        # inside ms_per_step: this rank's share of it spent in the shard merge (D2H of the 40-byte entries, ONE
        # all-gather, host fold); outside: everything before the timed region (imports, particle rendering, uploads,
        # communicator creation, warm-up)
        "merge_ms": (1e3 * merge_s[0] / args.steps) if world > 1 else 0.0,
        "setup_s": setup_s,
        "comm_setup_s": comm_setup_s,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: %d^2 maps, %d particles, %d orientations/GPU, %d CTFs, +-%d px grid %d (%d "
                               "displacements)%s" % (workload_name, W.N, W.nP, W.nOrient, W.nCTF,
                                                      args.max_displacement, args.grid, int(W.pd.NtotDisp),
                                                      (", WRITE_PROB_ANGLES %d" % K) if K else ""),
                   "pixels": W.N, "particles": W.nP, "orientations_per_gpu": W.nOrient, "ctf": W.nCTF,
                   "displacements": int(W.pd.NtotDisp), "algo": args.algo, "orientation_list": "seeded uniform random quaternions",
                   "fast_path": bool(E.fast_path), "cross_correlation": "direct (real-space sliding window)" if args.direct else "transform", "parallelism": "orientation blocks x%d, one RCCL all-gather + "
                   "log-sum-exp fold" % world},
    }
    # ---- roofline of the dominant kernel: VALU issue (the kernel is instruction-issue bound, not HBM bound) ----
    rl = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": None,
          "traffic": None, "kernel": sig, "launches": int(launches), "avg_launch_ms": avg_ms,
          "comparisons_per_launch": cpl,
          "peak_definition": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)"}
    # work-based bound (secondary): what the REFERENCE algorithm spends per comparison -- spectrum product
    # 6 M + full unpruned 2-D c2r 2.5 N^2 log2(N^2) + posterior per displacement -- at the measured rate, against the
    # FP32 vector peak.  Independent of how many instructions this kernel issues for that work (the pruned transforms
    # need ~0.85 M lane-operations for it): a kernel that issues twice the instructions does NOT score the same here.
    import math
    M = W.N * (W.N // 2 + 1)
    alg_flops = 6.0 * M + 2.5 * W.N * W.N * math.log2(W.N * W.N) + int(W.pd.NtotDisp) * POSTERIOR_FLOPS_PER_DISPLACEMENT
    rl["alg_flops_per_comparison"] = alg_flops
    rl["alg_flops_definition"] = ("6 M + 2.5 N^2 log2(N^2) + NtotDisp x %d (SURVEY.md 8d: the reference's product, full "
                                  "c2r and posterior)" % POSTERIOR_FLOPS_PER_DISPLACEMENT)
    rl["alg_TFLOPs_whole_job"] = value * alg_flops / 1e12 / world          # per GPU
    rl["alg_TFLOPs_kernel"] = (ncomp * alg_flops / 1e12) / (kms / 1e3) if kms > 0 else None
    rl["fp32_vector_peak_TFLOPs"] = FP32_VECTOR_PEAK_TFLOPS
    rl["frac_of_fp32_vector_peak"] = (rl["alg_TFLOPs_kernel"] / FP32_VECTOR_PEAK_TFLOPS) if rl["alg_TFLOPs_kernel"] else None
    # north_star's HBM model as a fraction, stated so that no reader has to derive it: it EXCEEDS 1 because the four waves
    # of a block share the particle rows through L1 and the blocks of an XCD through L2 -- the model is not a bound
    if kms > 0:
        rl["alg_bytes_frac"] = (ncomp * b_alg / 1e9) / (kms / 1e3) / HBM_PEAK_GBS
        rl["alg_bytes_note"] = ("algorithmic bytes (8 N (N/2+1) per comparison, SURVEY.md 8d) / launch time / 8 TB/s; "
                                "above 1 by cache reuse (L1 69 %, L2 90 % hit rate), counter traffic is hbm.frac_of_peak")
    hbm = {"peak_GBps": HBM_PEAK_GBS, "alg_bytes_per_comparison": b_alg,
           "alg_GBps": (ncomp * b_alg / 1e9) / (kms / 1e3) if kms > 0 else None,
           "alg_note": "north-star model (one particle half-spectrum per comparison); L1/L2 reuse makes it exceed the "
                       "real traffic, it is not a fraction of anything",
           "compulsory_bytes_per_launch": (W.nP + cpl / W.nP) * b_alg if cpl else None,
           "counter_bytes_per_launch": None, "counter_GBps": None, "frac_of_peak": None}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            pmc = json.load(f)
        same = (pmc.get("kernel") and sig.replace(" ", "") in pmc["kernel"].replace(" ", "") and cpl and
                abs(cpl - pmc["comparisons_per_launch"]) < 1 and pmc["config"]["pixels"] == W.N and
                pmc["config"]["particles"] == W.nP and pmc["config"]["ctf"] == W.nCTF and
                pmc["config"]["displacements"] == int(W.pd.NtotDisp))
        from bioem_amd.buildinfo import source_blobs
        # (the sources of the measured kernel's own translation unit: bioem_amd/buildinfo.py)
        stale = [k for k, v in source_blobs(sig).items() if pmc.get("source_blobs", {}).get(k) != v]
        if same and stale:
            same = False
            rl["counters_refused"] = ("profiles/pmc_current.json was measured on other device sources (git blob hash "
                                      "differs: %s); re-run scripts/profile_round.sh" % ", ".join(stale))
        if same and avg_ms:
            d = pmc["derived"]
            ipc = d["valu_wave_instr_per_comparison"]
            rl["valu_wave_instr_per_comparison"] = ipc
            rl["achieved"] = ipc * cpl / (avg_ms / 1e3) / 1e9
            rl["frac"] = rl["achieved"] / VALU_PEAK_GINSTR
            rl["clock_mhz_under_pmc"] = d.get("clock_mhz_under_pmc")
            rl["frac_at_sustained_clock"] = d.get("valu_issue_frac_at_sustained_clock")
            rl["utilisation"] = "frac = VALU issue utilisation: counter-measured wave-instructions of this build at the live launch time"
            rl["counters_source"] = ("profiles/%s_pmc_summary.json (same kernel instantiation, launch shape and git blob "
                                     "hashes of bioem_amd/csrc/*)" % pmc["tag"])
            rl["traffic"] = d.get("traffic_bytes_per_launch")
            hbm["counter_bytes_per_launch"] = d.get("traffic_bytes_per_launch")
            if d.get("traffic_bytes_per_launch"):
                hbm["counter_GBps"] = d["traffic_bytes_per_launch"] / (avg_ms / 1e3) / 1e9
                hbm["frac_of_peak"] = hbm["counter_GBps"] / HBM_PEAK_GBS
            hbm["l2_hit_rate"] = d.get("l2_hit_rate")
        else:
            rl["counters_source"] = "none: profiles/pmc_current.json is for another kernel, launch shape or source tree"
    except (OSError, KeyError, ValueError, TypeError):
        rl["counters_source"] = "none: profiles/pmc_current.json missing"
    out["roofline"] = rl
    out["hbm"] = hbm

    if world == 1:
        # the TIMED pass checks itself: its last step's block must be finite, must select the planted orientation
        # (particle p was rendered from orientation (7919 p) mod orientations) for most particles, and one more
        # (untimed) step on the same inputs must reproduce it bit for bit
        a = last[0] if K else last
        rerun = one_step()
        b = rerun[0] if K else rerun
        planted = (7919 * np.arange(nMaps)) % W.nOrient
        rec = float((a["orient"] == planted).mean())
        fin = bool(np.isfinite(a["Constoadd"]).all() and (a["Total"] > 0).all() and np.isfinite(a["Total"]).all())
        bit = bool(a.tobytes() == b.tobytes())
        out["result_check"] = {"of": "the last timed step", "finite": fin, "planted_orientation_recovered": rec,
                               "bitwise_equal_to_untimed_rerun": bit, "ok": bool(fin and bit and rec >= 0.9)}
    out["particle_stack"] = stack_source
    if world > 1 and rank == 0:
        # the merged block, checked: arg-max orientations are global indices, and the planted truth (particle p was
        # rendered from orientation (7919 p) mod orientations of block 0) is what most particles must select
        merged = last[0] if K else last
        planted = (7919 * np.arange(nMaps)) % W.nOrient
        out["merge_check"] = {"orient_in_range": bool(((merged["orient"] >= 0) & (merged["orient"] < world * W.nOrient)).all()),
                              "planted_orientation_recovered": float((merged["orient"] == planted).mean()),
                              "finite": bool(np.isfinite(merged["Constoadd"]).all() and (merged["Total"] > 0).all())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        # every host core this process may use: affinity mask, cut to the cgroup's CPU quota if there is one; on a box
        # where neither tells the share (all 256 host cores visible, quota unlimited) BIOEM_CPU_THREADS sets it
        nthreads = orc.usable_cpus(cap=int(os.environ.get("BIOEM_CPU_THREADS", 1 << 20)))
        nco = min(args.cpu_orientations, W.nOrient)
        v, times, want, const, particle_check = cpu_baseline(W, nco, nthreads, max(1, args.cpu_repeats), args.algo)
        import ctypes.util
        out["cpu_baseline"] = {"value": v, "unit": "comparisons/s", "cores": nthreads, "kind": "port",
                               "cpu_model": cpu_model(), "host_cores_total": os.cpu_count(),
                               "cgroup_cpu_limit": orc.cgroup_cpu_limit(),
                               "repeat_seconds": [round(t, 2) for t in times], "statistic": "median of repeats",
                               "fftw3f_on_box": bool(ctypes.util.find_library("fftw3f")),
                               "sample": "%d orientations x %d CTF x %d particles of the same workload, %d repeats"
                               % (nco, W.nCTF, W.nP, len(times))}
        # the checker's second job: the HIP path on the SAME sample against the oracle (untimed)
        raw, got, _ = new_prob_block(nMaps, 0 if shard_engine else W.nOrient, 0)
        E.start_run(raw)
        E.project_convolve_compare(0, nco)
        E.finish_run(raw)
        la = np.log(got["Total"]) + got["Constoadd"] + const
        lb = np.log(want["Total"]) + want["Constoadd"] + const
        same = ((got["orient"] == want["orient"]) & (got["conv"] == want["conv"]) &
                (got["cent_x"] == want["cent_x"]) & (got["cent_y"] == want["cent_y"]))
        out["parity"] = {"against": "CPU oracle on the cpu_baseline sample, particle spectra and sums computed by the "
                                    "oracle from the real-space stack", "particles": int(nMaps),
                         "device_particle_transform_vs_oracle": particle_check,
                         "max_abs_dlogp": float(np.abs(la - lb).max()),
                         "max_rel_dlogp": float((np.abs(la - lb) / np.abs(lb)).max()),
                         "argmax_mismatches": int((~same).sum()), "tolerance_rel": 1e-4}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if stack_file and rank == 0:
            try:
                os.unlink(stack_file)
            except OSError:
                pass
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # under a launcher the ranks' stderr is interleaved (or lost): every rank keeps its own traceback in a file
        import traceback
        if not os.environ.get("BIOEM_BENCH_SELF_LAUNCHED"):        # (self-launched ranks: stderr IS that file)
            try:
                with open(os.path.join(log_dir(), "bench_rank%s.err" % os.environ.get("RANK", "0")), "a") as f:
                    traceback.print_exc(file=f)
            except OSError:
                pass
        traceback.print_exc()
        sys.exit(3)
