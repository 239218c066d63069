"""bench.py as its own launcher (`python bench.py --gpus N` with WORLD_SIZE unset): the per-rank environment it builds, and
that on a box with fewer than N devices it says so on stdout as one JSON line with a non-zero exit code (not "needs
torch.distributed.run").  CPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_env():
    import bench
    base = {"PATH": "/usr/bin", "RANK": "7"}
    e = bench.rank_env(base, 3, 8, 29511)
    assert e["RANK"] == "3" and e["LOCAL_RANK"] == "3" and e["WORLD_SIZE"] == "8" and e["LOCAL_WORLD_SIZE"] == "8"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" and e["PATH"] == "/usr/bin"
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert base["RANK"] == "7"  # the caller's environment is not touched
    assert bench.rank_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 0, 2, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_too_few_devices_is_reported_as_json():
    import torch
    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode not in (0, 2), (p.returncode, p.stderr[-500:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["error"] == "needs %d devices, found %d" % (n, n - 2) and j["n_gpus"] == n
    assert "torch.distributed.run" not in p.stdout + p.stderr


def _fake_rank_script(tmp_path, body):
    p = tmp_path / "fake_rank.py"
    p.write_text("import os, sys, time\nrank = int(os.environ['RANK'])\n" + body)
    return str(p)


def test_a_dead_rank_ends_the_run(tmp_path, capsys):
    """rank 1 dies at start-up, rank 0 would wait (here: sleeps 600 s) -- the launcher must end it and return rank 1's code"""
    import time
    import bench
    script = _fake_rank_script(tmp_path, "assert os.environ['ZH_BENCH_RDZV_FILE']\n"
                                         "if rank == 1:\n    sys.exit(7)\nprint('{\"from_rank\": 0}', flush=True)\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rc = bench.launch_ranks(2, [], timeout_s=120, count=lambda: 2, script=script)
    assert rc == 7 and time.monotonic() - t0 < 60
    out = capsys.readouterr().out.strip().splitlines()
    assert out[0] == '{"from_rank": 0}'                      # rank 0's stdout is relayed
    last = json.loads(out[-1])
    assert "ranks failed" in last["error"] and last["n_gpus"] == 2  # and the LAST line says the run failed


def test_hung_ranks_time_out(tmp_path, capsys):
    import time
    import bench
    script = _fake_rank_script(tmp_path, "time.sleep(600)\n")
    t0 = time.monotonic()
    rc = bench.launch_ranks(2, [], timeout_s=2, count=lambda: 2, script=script)
    assert rc == 124 and time.monotonic() - t0 < 60
    assert "timeout" in json.loads(capsys.readouterr().out.strip().splitlines()[-1])["error"]


def test_all_ranks_ok(tmp_path, capsys):
    import bench
    script = _fake_rank_script(tmp_path, "print('x' * 200000 if rank == 0 else '')\nprint('{\"ok\": %d}' % rank)\n")  # (more than a pipe holds)
    assert bench.launch_ranks(3, [], timeout_s=120, count=lambda: 3, script=script) == 0
    assert capsys.readouterr().out.strip().splitlines()[-1] == '{"ok": 0}'
