"""CPU suite: the torch-free rank start-up and control plane of N > 1 runs (pde_multigrid_amd/launch.py) with world
size 2 and 3: rendezvous through the job file, broadcast of a 128-byte id, all_gather, max, barrier; a failing rank
takes the job down with a non-zero code instead of leaving the others waiting."""
import importlib.util
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch():
    spec = importlib.util.spec_from_file_location("mgx_launch", os.path.join(ROOT, "pde_multigrid_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


WORKER = textwrap.dedent("""
    import importlib.util, json, os, sys
    spec = importlib.util.spec_from_file_location("mgx_launch", os.path.join(%r, "pde_multigrid_amd", "launch.py"))
    L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("FAIL_RANK") == str(rank):
        sys.exit(7)
    r = L.Rendezvous(rank, world, timeout=30)
    uid = r.broadcast_bytes(bytes(range(128)) if rank == 0 else b"")
    assert uid == bytes(range(128))
    parts = r.all_gather({"rank": rank, "sq": rank * rank})
    assert [p["rank"] for p in parts] == list(range(world))
    assert r.max(1.5 + rank) == 1.5 + world - 1
    assert r.broadcast("x" if rank == 1 else None, src=1) == "x"
    for _ in range(20):
        r.barrier()
    r.close()
    assert "torch" not in sys.modules
    with open(os.path.join(sys.argv[1], "ok%%d" %% rank), "w") as fh:
        fh.write("ok")
""") % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_spawn_rendezvous_collectives(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    rc = _launch().spawn(str(script), [str(tmp_path)], world, timeout=60, need_gpus=False)
    assert rc == 0
    assert sorted(p.name for p in tmp_path.glob("ok*")) == ["ok%d" % r for r in range(world)]


def test_torchrun_style_environment(tmp_path):
    """ranks started by somebody else's launcher: only RANK / WORLD_SIZE / MASTER_* in the environment"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29%03d" % (os.getpid() % 1000))
        env.pop("MGX_RDZV_KEY", None)
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env))
    assert [p.wait(timeout=60) for p in procs] == [0, 0]


def test_failing_rank_takes_the_job_down(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    rc = _launch().spawn(str(script), [str(tmp_path)], 2, timeout=60, env_extra={"FAIL_RANK": "1"}, need_gpus=False)
    assert rc == 7
    assert not list(tmp_path.glob("ok*"))


def test_bench_parent_refuses_without_gpus():
    """`bench.py --gpus 2` on a box with fewer GPUs: exit code 2 and a message, no child is started (here: no GPU at all)"""
    if _launch().count_gpus() >= 2:
        pytest.skip("this box has two GPUs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert p.returncode == 2 and "needs 2 GPUs" in p.stderr
