"""-m gpu: the N > 1 launch path of bench.py as the driver starts it -- `python bench.py --gpus N ...` from a FRESH process that
starts the ranks itself (bench.py's self-launch: a child `python -m torch.distributed.run`, nothing is exec'ed from a process that
holds the GPU) -- on the one-GPU box: two ranks share the device, `--backend gloo` (RCCL refuses two ranks on one device; with
`--backend nccl` the same control flow runs over the library's RCCL communicator). What is pinned: the job starts, rendezvous and the
start-up self-check pass, the JSON line parses, and the decisions equal the one-rank run of the same command.
The N > 1 logic itself is covered on the CPU (tests/test_shard_gloo.py, tests/test_host_control_flow.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    """the ONE stdout line must parse and stay under 4 KB (what the driver reads); the per-step decisions are in the detail file"""
    import tempfile

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("WORLD_SIZE", None)
    with tempfile.TemporaryDirectory() as d:
        env["SCLENS_BENCH_DETAIL"] = os.path.join(d, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                           timeout=timeout, cwd=ROOT)
        assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) <= 4096, r.stdout[-2000:]
        line = json.loads(lines[0])
        full = json.load(open(env["SCLENS_BENCH_DETAIL"]))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "dtype"):
        assert line[key] == full[key], key
    assert all(full["config"][k] == v for k, v in line["config"].items())  # the line's config is a subset of the full record's
    return full


def _decisions(out):
    d = out["observed"]["decisions_per_step"]
    assert len(d) == out["steps"] >= 1
    return [(q["signals"], q["robust_signals"], q["search_iters"], q["p_"]) for q in d]


def test_bench_two_ranks_on_one_gpu_equal_the_one_rank_run():
    """ensemble t mod G + speculative search rounds + spread first phase at cfg2, two ranks (gloo) against one"""
    common = ["--config", "cfg2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--strict-fp32", "off",
              "--budget-s", "800"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "2", "--backend", "gloo", *common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["comm"]["world"] == 2 and "gloo" in two["config"]["comm"]["transport"]
    assert _decisions(one) == _decisions(two), (_decisions(one), _decisions(two))
    assert one["observed"]["signals"] > 0


def test_bench_row_sharded_two_ranks_on_one_gpu():
    """cells row-sharded at 20 000 x 6 000: every rank holds half of the cells and its own candidates; the signal
    count is the unsharded run's, the selected sparsity within two steps of it (the candidate LIST is the concatenation of the
    ranks' lists -- another order than the unsharded draw, so the samples differ: tests/test_gpu_atlas.py replays that exactly)"""
    common = ["--config", "rs20k", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--strict-fp32", "off",
              "--budget-s", "800", "--n-perturb", "6"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "2", "--backend", "gloo", "--row-shard", *common)
    assert two["n_gpus"] == 2 and "row-sharded" in two["config"]["parallelism"]
    (k1, r1, s1, p1), (k2, r2, s2, p2) = _decisions(one)[0], _decisions(two)[0]
    assert k1 == k2 > 0 and r1 == r2
    assert abs(p1 - p2) <= 0.002 + 1e-12 and abs(s1 - s2) <= 2


def test_bench_eight_ranks_on_one_gpu_equal_the_one_rank_run():
    """the driver's N = 8 launch (`python bench.py --gpus 8`, one rank per GPU of the target node) rehearsed on the one-GPU box: eight gloo
    ranks share the device at 20 000 x 6 000 (a matrix with signals: the search, the ensemble and the scoring all run) -- search rounds of
    8 x streams evaluations, 16 ensemble members t mod 8, first decompositions on ranks 0 / 1 / 2, the status agreements and the final
    gather all at the target world size; decisions equal the one-rank run"""
    common = ["--config", "rs20k", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--strict-fp32", "off",
              "--budget-s", "1200", "--n-perturb", "16"]
    one = _bench("--gpus", "1", *common)
    eight = _bench("--gpus", "8", "--backend", "gloo", *common, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["config"]["comm"]["world"] == 8
    assert _decisions(one) == _decisions(eight), (_decisions(one), _decisions(eight))
    assert one["observed"]["signals"] > 0
