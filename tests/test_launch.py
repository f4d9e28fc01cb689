"""graphembeddings_amd/launch.py: `bench.py --gpus N` / `train.py --gpus G` start their own ranks (no torchrun).
CPU tests: the child environment, failure propagation, and a real world-2 gloo rendezvous of two spawned children."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from graphembeddings_amd import launch as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_child_env_is_what_env_rendezvous_reads():
    base = {"PATH": "/bin", "RANK": "7", "FOO": "bar"}
    envs = [L.child_env(r, 3, 29999, base) for r in range(3)]
    for r, e in enumerate(envs):
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "3"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999"   # never the container hostname
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                             # dmabuf IPC, as RCCL needs on this pool
        assert e["FOO"] == "bar" and e["PATH"] == "/bin"
    assert base["RANK"] == "7"                                                   # the parent's environment is not edited
    assert L.launched_by_a_launcher(envs[0]) and not L.launched_by_a_launcher({"PATH": "/bin"})
    with pytest.raises(ValueError):
        L.child_env(3, 3, 1, base)


def test_world_larger_than_the_visible_devices_is_refused(monkeypatch):
    monkeypatch.setattr(L, "visible_devices", lambda: 1)
    with pytest.raises(RuntimeError):
        L.check_world(2, env={})
    L.check_world(2, env={"GE_SINGLE_DEVICE": "1"})     # the one-GPU rehearsal: every rank on cuda:0
    L.check_world(1, env={})
    with pytest.raises(ValueError):
        L.check_world(0, env={})


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawned_ranks_rendezvous_over_gloo_and_rank0_prints_the_line(tmp_path, monkeypatch):
    monkeypatch.setenv("GE_SINGLE_DEVICE", "1")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    out = tmp_path / "out.json"
    script = _script(tmp_path, f"""
        import json, os, sys
        import torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        seen = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(seen)
        if dist.get_rank() == 0:
            open({str(out)!r}, "w").write(json.dumps({{"ranks_seen": int(seen), "argv": sys.argv[1:]}}))
        dist.barrier()
        dist.destroy_process_group()
    """)
    rc = L.spawn_ranks(2, ["--steps", "3"], script=script)
    assert rc == 0
    got = json.loads(out.read_text())
    assert got == {"ranks_seen": 2, "argv": ["--steps", "3"]}


def test_a_failing_rank_stops_the_others_and_fails_the_parent(tmp_path, monkeypatch):
    monkeypatch.setenv("GE_SINGLE_DEVICE", "1")
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)          # rank 0 would wait for ever: the parent has to stop it
    """)
    rc = L.spawn_ranks(2, [], script=script, grace_s=5.0)
    assert rc == 3


def test_bench_refuses_more_ranks_than_devices_without_touching_a_gpu():
    # no GPU in the build container: --gpus 2 must fail with a diagnosable line, not run one rank and print n_gpus: 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "GE_SINGLE_DEVICE")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 0 and "device" in line["error"]
