"""bench.py's own launcher (python bench.py --gpus N without torchrun): a rank that dies before or inside the
rendezvous must not leave the others waiting for the backend's timeout -- the parent watches every child, terminates
the rest on the first non-zero exit, keeps each rank's stderr in bench_rank<k>.err and returns the failed rank's code."""
import io
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, time
rank = int(os.environ["RANK"])
assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
if rank == 1:
    time.sleep(0.5)
    sys.stderr.write("rank 1: hipErrorNoDevice (stand-in failure)\n")
    sys.exit(3)
if rank == 0:
    print('{"partial": true}', flush=True)
time.sleep(120)          # stands for a rank waiting in the rendezvous / all-gather for the one that died
'''


def test_watchdog_terminates_the_other_ranks_and_reports_the_failed_one(tmp_path, monkeypatch, capsys):
    import bench
    monkeypatch.setenv("BIOEM_BENCH_LOGDIR", str(tmp_path))
    out = io.StringIO()
    t0 = time.time()
    rc = bench.self_launch(3, [sys.executable, "-c", CHILD], out=out)
    dt = time.time() - t0
    assert rc == 3 and dt < 10.0, (rc, dt)
    assert '"partial"' in out.getvalue()                        # rank 0's output is still handed through
    assert "hipErrorNoDevice" in open(tmp_path / "bench_rank1.err").read()
    err = capsys.readouterr().err
    assert "rank 1 of 3 exited with code 3" in err and "hipErrorNoDevice" in err
    for r in (0, 1, 2):
        assert os.path.exists(tmp_path / ("bench_rank%d.err" % r))


def test_launcher_returns_zero_and_rank0_output_when_all_ranks_succeed(tmp_path, monkeypatch):
    import bench
    monkeypatch.setenv("BIOEM_BENCH_LOGDIR", str(tmp_path))
    out = io.StringIO()
    rc = bench.self_launch(2, [sys.executable, "-c", "import os; print('line from rank', os.environ['RANK'])"], out=out)
    assert rc == 0 and out.getvalue() == "line from rank 0\n"


def test_bench_with_an_injected_rank_failure_returns_quickly(tmp_path):
    """The real script: rank 1 raises before the rendezvous (BIOEM_BENCH_INJECT_FAILURE=1).  On a box without a GPU
    rank 0 stops at its own "needs a GPU" check, so either rank may be reported first -- what is asserted is that the
    launch returns within seconds, non-zero, with the injected message in rank 1's file."""
    env = dict(os.environ, BIOEM_BENCH_LOGDIR=str(tmp_path), BIOEM_BENCH_INJECT_FAILURE="1", BIOEM_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60
    assert "injected failure on rank 1" in open(tmp_path / "bench_rank1.err").read()
    assert "exited with code" in r.stderr


def test_config_selection_follows_baseline_json():
    import bench
    argv = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8"]
        a = bench.parse_args()
        assert (a.particles, a.defocus, a.envelopes, a.orientations, a.pixels) == (10000, 2, 5, 4608, 224)   # config 3
        sys.argv = ["bench.py", "--gpus", "4"]
        a = bench.parse_args()
        assert (a.particles, a.defocus, a.envelopes, a.orientations) == (1000, 1, 5, 4608)                   # config 2
        sys.argv = ["bench.py", "--config", "3"]
        a = bench.parse_args()
        assert (a.gpus, a.particles, a.defocus) == (1, 10000, 2)
        sys.argv = ["bench.py", "--gpus", "8", "--config", "2", "--particles", "64"]
        a = bench.parse_args()
        assert (a.particles, a.defocus) == (64, 1)
    finally:
        sys.argv = argv


def test_time_budget_trims_warmup_first_then_steps():
    """--max-seconds: plan_steps is the pure function every rank evaluates with the same max-reduced numbers."""
    import bench
    # everything fits: nothing is trimmed
    assert bench.plan_steps(20, 4, 1.0, 30.0, 480.0, 10.0) == (4, 20)
    # config 3 on eight GPUs: 8.6 s per step, 60 s of set-up gone, 480 s budget, 10 s tail -> 47 steps fit: untouched
    assert bench.plan_steps(20, 4, 8.6, 60.0, 480.0, 10.0) == (4, 20)
    # a slower start: 200 s gone -> 31 fit -> 20 + 4 still fit
    assert bench.plan_steps(20, 4, 8.6, 200.0, 480.0, 10.0) == (4, 20)
    # 120 s budget: (120 - 40 - 10) / 8.6 = 8 steps fit: no further warm-up, eight timed steps
    assert bench.plan_steps(20, 4, 8.6, 40.0, 120.0, 10.0) == (0, 8)
    # warm-up is what goes first: 22 fit -> 2 of the 4 remaining warm-up steps, all 20 timed ones
    assert bench.plan_steps(20, 4, 1.0, 448.0, 480.0, 10.0) == (2, 20)
    # nothing fits any more: one timed step always runs (the line must exist), no warm-up
    assert bench.plan_steps(20, 4, 8.6, 500.0, 480.0, 10.0) == (0, 1)
    # no budget / no estimate: untouched
    assert bench.plan_steps(20, 4, 8.6, 500.0, 0.0, 10.0) == (4, 20)
    assert bench.plan_steps(20, 4, 0.0, 10.0, 480.0, 10.0) == (4, 20)


def test_shared_stack_path_is_per_launch_and_per_shape(monkeypatch):
    import bench
    monkeypatch.setenv("MASTER_PORT", "29511")
    a = bench.shared_stack_path((224, 10000, 4608, 5, 2))
    b = bench.shared_stack_path((224, 1000, 4608, 5, 1))
    assert a != b and a.startswith("/dev/shm/") and "29511" in a
    monkeypatch.setenv("MASTER_PORT", "29512")
    assert bench.shared_stack_path((224, 10000, 4608, 5, 2)) != a
