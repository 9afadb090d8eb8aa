"""bench.py's N > 1 code path end to end without GPUs: `bench.py --gpus 2` launched the way the driver launches it (one process per
rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), on the emulated kernel library (tests/emu/emu_plugin.py) with
NT_BENCH_DRY_SINGLE_GPU=1 (gloo instead of RCCL, every rank on "cuda:0").  Checks the plumbing the first real 8-GPU run depends on:
per-rank world build (each rank builds only its own worlds of the global scene), barrier + MAX-over-ranks timing, exactly ONE JSON
line from rank 0 and none from the other rank, whole-job aggregate `value`, weak scaling, the contract's keys.  (The collectives on
the real backend: tests/test_gpu_nccl_single.py; shard equivalence: tests/test_shard_equivalence.py, test_sharding_gloo.py.)"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")

CHILD = r'''
import os, runpy, sys
sys.path[:0] = [os.environ["NT_EMU"], os.environ["NT_ROOT"], os.path.join(os.environ["NT_ROOT"], "tests")]
import emu_plugin  # noqa: F401  ("cuda" tensors -> host memory, product loader -> the emulated library)
sys.argv = ["bench.py"] + os.environ["NT_BENCH_ARGS"].split()
runpy.run_path(os.path.join(os.environ["NT_ROOT"], "bench.py"), run_name="__main__")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, bench_args, timeout=900):
    port = str(_free_port())
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   NT_BENCH_DRY_SINGLE_GPU="1", NT_EMU=EMU, NT_ROOT=ROOT, NT_BENCH_ARGS=bench_args)
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=timeout)
        outs.append((p.returncode, o, e))
    return outs


def _json_lines(text):
    return [json.loads(line) for line in text.splitlines() if line.startswith("{")]


def test_bench_two_ranks_dry_run(oracle_lib):
    args = "--gpus 2 --steps 2 --warmup 1 --envs-per-gpu 16 --settle-frames 0 --no-cpu-baseline"
    outs = _launch(2, args)
    for rank, (rc, o, e) in enumerate(outs):
        assert rc == 0, (rank, o[-2000:], e[-4000:])
    lines0, lines1 = _json_lines(outs[0][1]), _json_lines(outs[1][1])
    assert len(lines0) == 1 and len(lines1) == 0, (outs[0][1][-2000:], outs[1][1][-2000:])
    d = lines0[0]
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "env-steps/s" and d["dtype"] == "f32"
    assert d["metric"].startswith("env-steps/sec at 4096 batched envs (Anymal, XPBD)") and d["vs_baseline"] is None
    assert d["config"]["envs_per_gpu"] == 16 and d["config"]["parallelism"] == "env-shard x2"
    # whole-job aggregate: both ranks' env-steps over the MAX-over-ranks wall time
    total = 2 * 16 * d["config"]["substeps_per_step"] * d["steps"]
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and d["valid_state"] in (True, False)
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_bench_rejects_a_world_size_that_does_not_match(oracle_lib):
    outs = _launch(1, "--gpus 2 --steps 1 --warmup 0 --envs-per-gpu 16 --settle-frames 0 --no-cpu-baseline", timeout=300)
    rc, o, e = outs[0]
    assert rc != 0 and "WORLD_SIZE=1" in (o + e)


def test_bench_spawns_its_own_ranks_without_a_launcher(oracle_lib):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (no torch.distributed.run around it): bench.py starts the two
    ranks itself (one process per GPU, 127.0.0.1 rendezvous) and the command prints exactly ONE JSON line -- rank 0's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NT_BENCH_DRY_SINGLE_GPU="1", NT_EMU=EMU, NT_ROOT=ROOT,
               PYTHONPATH=os.pathsep.join([os.path.join(EMU, "site"), env.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "16",
                        "--settle-frames", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["config"]["parallelism"] == "env-shard x2" and d["scaling"] == "weak"
    total = 2 * 16 * d["config"]["substeps_per_step"] * d["steps"]
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]


def test_a_failing_rank_ends_the_self_spawned_job(oracle_lib):
    """A rank that dies takes the job down with a non-zero exit code instead of leaving the other rank in the barrier."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NT_BENCH_DRY_SINGLE_GPU="1", NT_EMU=EMU, NT_ROOT=ROOT, NT_BENCH_FAIL_RANK="1",
               PYTHONPATH=os.pathsep.join([os.path.join(EMU, "site"), env.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--envs-per-gpu", "16",
                        "--settle-frames", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not _json_lines(r.stdout)
