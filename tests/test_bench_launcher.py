"""bench.py --gpus N without an external launcher: the parent starts N rank processes with the
torch.distributed.run environment and relays rank 0's line (BASELINE configs 4/5 scale the way the reference
does with its Pool(8), rl/train.py:19).  The rank program here is a stand-in that does what a bench rank does
with torch.distributed on gloo — rendezvous from the env, a max-reduce of its time, a sum of its units —
so the spawn path is covered without a GPU."""
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RANK_PROGRAM = textwrap.dedent("""
    import json, os, sys
    sys.path[:0] = [%r, %r]
    import torch.distributed as dist
    from ebcsim import shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    dist.init_process_group("gloo")
    elapsed, units = shard.job_rate(1.0 + rank, 40960.0)
    if len(sys.argv) > 1 and int(sys.argv[1]) == rank:
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "elapsed": elapsed, "units": units}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, os.path.join(ROOT, "eb-cadrl_amd"))


def test_launch_ranks_relays_rank0_and_reduces():
    import bench
    rc, out = bench.launch_ranks(2, [sys.executable, "-c", RANK_PROGRAM], timeout=300)
    assert rc == 0, out
    line = json.loads(out.strip().splitlines()[-1])
    assert line == {"n_gpus": 2, "elapsed": 2.0, "units": 81920.0}  # max over ranks, sum of units


def test_launch_ranks_reports_a_failed_rank():
    import bench
    rc, out = bench.launch_ranks(2, [sys.executable, "-c", RANK_PROGRAM, "1"], timeout=300)
    assert rc != 0


HANG_PROGRAM = textwrap.dedent("""
    import os, sys, time
    if int(os.environ["RANK"]) == int(sys.argv[1]):
        sys.exit(7)          # e.g. a HIP error or a reported mailbox fault
    time.sleep(600)          # the survivors: in a barrier that never completes
""")


@pytest.mark.parametrize("dead", [0, 1, 2])
def test_a_dead_rank_ends_the_job_promptly(dead):
    """Whichever rank dies first, the parent returns its code at once and leaves no survivor behind."""
    import time
    import bench
    t0 = time.monotonic()
    rc, out = bench.launch_ranks(3, [sys.executable, "-c", HANG_PROGRAM, str(dead)], timeout=300)
    assert rc == 7 and time.monotonic() - t0 < 30


def test_job_timeout_is_finite():
    import time
    import bench
    t0 = time.monotonic()
    rc, out = bench.launch_ranks(2, [sys.executable, "-c", HANG_PROGRAM, "-1"], timeout=3)
    assert rc == 124 and time.monotonic() - t0 < 30


def test_parent_does_not_touch_the_gpu_before_spawning():
    """The self-launch branch sits before any torch import in main(): the parent must stay free of HIP state
    (a process that initialised the GPU must not start replacing itself, and need not hold a context)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("import torch")
    assert "os.exec" not in src


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.gpu
def test_bench_rank_through_rccl_on_one_rank():
    """The rank program itself with its process group on RCCL: one rank on the one GPU there is, the barriers, the
    MAX all-reduce of the block times and job_rate's reductions forced through the backend
    (EBCSIM_FORCE_COLLECTIVES=1) — the calls an N-GPU run makes, on device tensors."""
    import subprocess
    env = dict(os.environ, EBCSIM_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "5",
                        "--no-also", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 50 and line["value"] > 1e8
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5", "--no-also",
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    ref = json.loads(plain.stdout.strip().splitlines()[-1])
    assert abs(line["value"] / ref["value"] - 1) < 0.2  # the same job with and without the group


REHEARSAL_RANKS = int(os.environ.get("EBCSIM_REHEARSAL_RANKS", "3"))


@pytest.mark.gpu
def test_multi_rank_rehearsal_of_config4_on_one_gpu():
    """BASELINE config 4 in its N-rank FORM on the one GPU there is: `bench.py --gpus N --workload cfg4` starts N rank
    processes (gloo process group), every rank builds ITS env slice of 16384 x 5 from its own seeds and the job prints
    one line.  A one-GPU box admits at most 6 processes on its card (the box's process guard counts this test runner
    too), so by default 3 ranks run: two share the device, the third is a host-only rehearsal rank (its slice on the host
    and every collective, no GPU).  No scaling number can come out of ranks that share a GPU — the line says so
    ("timing": "rehearsal"); the measured 1-2-4-8 curve is the driver's, on an 8-GPU node."""
    import subprocess
    n = REHEARSAL_RANKS
    env = dict(os.environ, EBCSIM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "cfg4", "--steps", "5",
                        "--warmup", "2", "--no-cpu-baseline", "--no-also", "--rehearsal-gpu-ranks", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == n and line["config"]["envs_per_gpu"] == 16384 and line["timing"] == "rehearsal"
    assert line["ranks_on_gpu"] == 2 and line["roofline"]["achieved"] > 0 and line["scaling"] == "weak"
    assert line["ranks_with_gpu_open"] == 2  # the host-only rank really stayed off the card
    assert line["units_per_step"] == n * 16384 * 5
