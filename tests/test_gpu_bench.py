"""GPU suite: bench.py itself.  The slab-decomposed branch with one rank (MGX_BENCH_FORCE_DIST=1: rendezvous, RCCL
communicator, slab hierarchy) must pass its result check WITHOUT PyTorch in the process, report what the communicator
says about itself, and the default N = 1 line must carry the contract's fields, the batch statistics and the secondary
configurations with their own result checks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MGX_RDZV_KEY")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_forced_slab_branch_one_rank_without_torch():
    out = _bench(["--gpus", "1", "--size", "257", "--steps", "3", "--warmup", "1", "--batches", "2", "--no-cpu-baseline"],
                 {"MGX_BENCH_FORCE_DIST": "1"})
    assert out["result_check"]["status"] == "ok"
    comm = out["config"]["communicator"]
    assert comm["ranks_seen"] == 1 and comm["rccl_version"] > 20000 and comm["torch_imported"] is False
    assert out["n_gpus"] == 1 and out["steps"] == 3 and len(out["batches"]["ms_per_step"]) == 2


@pytest.mark.timeout(900)
def test_default_line_contract_and_secondary_configs():
    out = _bench(["--steps", "5", "--warmup", "2", "--batches", "3", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in out
    assert out["result_check"]["status"] == "ok"
    assert out["batches"]["min"] <= out["batches"]["median"] == out["ms_per_step"]
    assert 0.3 < out["roofline"]["frac"] < 1.0 and out["roofline"]["bound"] == "hbm"
    down = out["roofline_down"]  # the fused last black pass + residual + restrict launch of the finest level
    assert down["kernel"].startswith("relax_rr3d_xs_kernel<double") and 0.2 < down["frac"] < 1.0 and down["avg_launch_us"] > 100
    sec = out["secondary"]
    assert len(sec) == 5 and all(c["result_check"] == "ok" for c in sec.values()), sec  # configs[1], [2], fp32, the two published workloads
    pub = [c for k, c in sec.items() if k.startswith("published workload")][0]
    assert 0.05 < pub["seconds"] < 5.0 and pub["known_answer"] == "3d_n129_fmg_2_3000_3000_f32"
