"""CPU: `python bench.py --gpus N` without a launcher starts N child ranks itself (before any GPU
call) and only relays their output and exit code (bench.launch_ranks)."""
import json
import os
import sys
import textwrap

import bench


def test_launch_ranks_starts_n_children_and_relays_rank0(tmp_path, capfd):
    script = tmp_path / "fake_bench.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        import torch
        t = torch.ones(1) * (dist.get_rank() + 1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": t.item(),
                              "argv": sys.argv[1:], "master": os.environ["MASTER_ADDR"]}))
        dist.destroy_process_group()
        sys.exit(0)
    """))
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "3"], script=str(script))
    out = capfd.readouterr().out
    assert rc == 0
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2", "--steps", "3"],
                    "master": "127.0.0.1"}


def test_launch_ranks_returns_the_childrens_failure(tmp_path):
    script = tmp_path / "failing.py"
    script.write_text("import sys; sys.exit(3)\n")
    assert bench.launch_ranks(2, [], script=str(script)) != 0


def test_main_becomes_the_launcher_only_without_world_size(monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, script=None: calls.append(
        (n, list(argv))) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LCF_REHEARSE", raising=False)
    # the launcher makes no GPU call of any kind (not even a device count): a process that has
    # initialised HIP must not start other GPU programs on this pool
    import torch

    def forbidden(*a, **k):
        raise AssertionError("the launcher touched torch.cuda")
    monkeypatch.setattr(torch.cuda, "device_count", forbidden)
    monkeypatch.setattr(torch.cuda, "is_available", forbidden)
    assert bench.main(["--gpus", "2", "--steps", "5", "--warmup", "1"]) == 0
    assert calls == [(2, ["--gpus", "2", "--steps", "5", "--warmup", "1"])]


def test_a_rank_without_a_device_refuses_with_exit_code_2(monkeypatch, capsys):
    """The visible-GPU check lives in the child ranks: LOCAL_RANK >= device count -> exit code 2
    and a message, before any process group is set up."""
    import torch
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", str(torch.cuda.device_count() + 3))
    monkeypatch.delenv("LCF_REHEARSE", raising=False)
    assert bench.main(["--gpus", "1", "--steps", "1", "--warmup", "0"]) == 2
    assert "has no GPU" in capsys.readouterr().err
