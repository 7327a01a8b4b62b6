"""bench.py --gpus N without a launcher around it: the process must start N fresh ranks itself (before anything
touches the GPU) with distinct RANK / LOCAL_RANK and one rendezvous address, and fail when a rank fails."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}


def test_dry_launch_starts_one_rank_per_gpu():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=_clean_env(), capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ranks = json.loads(out.stdout.strip().splitlines()[-1])["dry_launch"]
    assert len(ranks) == 4
    assert sorted(r["local_rank"] for r in ranks) == [0, 1, 2, 3]
    assert sorted(r["rank"] for r in ranks) == [0, 1, 2, 3]
    assert {r["world_size"] for r in ranks} == {4}
    assert len({r["master"] for r in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    assert len({r["pid"] for r in ranks}) == 4  # separate processes


def test_under_a_launcher_the_script_is_a_rank_not_a_launcher():
    env = dict(_clean_env(), RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout)["rank"] == 1


def test_launcher_does_not_load_torch_and_reports_a_failed_rank():
    # the launcher itself must stay clear of torch (and so of the GPU): run launch() with torch imports poisoned
    code = ("import sys, json; sys.modules['torch'] = None; sys.argv = ['bench.py', '--gpus', '2', '--dry-launch'];"
            "import runpy; runpy.run_path(%r, run_name='__main__')" % BENCH)
    out = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert len(json.loads(out.stdout.strip().splitlines()[-1])["dry_launch"]) == 2
    # ranks that cannot run (no GPU here, or a bad flag) make the launcher exit non-zero
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--tokens", "64",
                          "--backend", "no-such-backend"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
