"""bench.py's launch logic, checked without a GPU: `--gpus N` must either BE an N-rank run (ranks started by bench.py
itself, or by a launcher whose WORLD_SIZE agrees) or exit non-zero -- never print an n_gpus=1 line under an N label."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(kw)
    return e


def test_import_has_no_side_effects():
    """Scripts import bench for W/H/F and synth_chunk: that must not redirect stdout or import torch."""
    out = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, %r); import bench; print('torch' in sys.modules, bench.W, bench.H, bench.F)" % ROOT],
                         capture_output=True, text=True, env=_env(), timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["False", "1920", "1080", "64"]


def test_plan_ranks_is_one_process_per_gpu():
    sys.path.insert(0, ROOT)
    import bench
    plan = bench.plan_ranks(4, ["--gpus", "4", "--steps", "2", "--warmup", "1"], {"PATH": "x", "WORLD_SIZE": "stale"}, 12345, python="py")
    assert len(plan) == 4
    ports = set()
    for r, (cmd, env) in enumerate(plan):
        assert cmd == ["py", BENCH, "--gpus", "4", "--steps", "2", "--warmup", "1"]
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "x"
        ports.add(env["MASTER_PORT"])
    assert ports == {"12345"}


def test_check_launch_cases():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.check_launch(1, {}, None) == "rank"
    assert bench.check_launch(8, {"WORLD_SIZE": "8"}, None) == "rank"          # under torchrun
    assert bench.check_launch(8, {}, 8) == "spawn"                              # bare `python bench.py --gpus 8`
    assert bench.check_launch(2, {}, 1).startswith("error")                     # a 1-GPU box
    assert bench.check_launch(2, {"WORLD_SIZE": "4"}, None).startswith("error")
    assert bench.check_launch(1, {"WORLD_SIZE": "2"}, None).startswith("error")
    assert bench.check_launch(0, {}, None).startswith("error")


def test_gpus2_without_two_gpus_exits_nonzero():
    """This container has no GPU: `python bench.py --gpus 2` must fail loudly and print no result line."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=_env(), timeout=120)
    assert out.returncode != 0
    assert out.stdout.strip() == ""
    assert "--gpus 2" in out.stderr and "visible" in out.stderr


def test_world_size_mismatch_exits_nonzero():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True,
                         env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "WORLD_SIZE=4" in out.stderr


def test_spawn_dry_run_shows_two_children():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                         env=_env(ALICE_BENCH_SPAWN_DRYRUN="1"), timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)
    assert [p["env"]["RANK"] for p in plan] == ["0", "1"]
    assert all(p["env"]["WORLD_SIZE"] == "2" and p["cmd"][1:] == [BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"] for p in plan)
    assert len({p["env"]["MASTER_PORT"] for p in plan}) == 1
