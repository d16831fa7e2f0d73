"""CPU: `python bench.py --gpus N` is all a caller needs for N > 1 -- without a launcher around it the process starts the N ranks as a
CHILD under torch.distributed.run (never an exec, no GPU call in the parent) and returns the child's exit code."""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_dry_launch_prints_the_child_command():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.strip().splitlines()[-1].split()
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]           # the caller's own flags, minus --dry-launch


def test_self_launch_starts_a_child_and_relays_its_exit_code(monkeypatch):
    import bench
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(os, "execv", lambda *a: (_ for _ in ()).throw(AssertionError("never exec")))
    monkeypatch.setattr(os, "execvp", lambda *a: (_ for _ in ()).throw(AssertionError("never exec")))
    import torch
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a: (_ for _ in ()).throw(AssertionError("the parent makes no GPU call")))
    args = types.SimpleNamespace(gpus=8, dry_launch=False)
    rc = bench.self_launch(args, ["--gpus", "8", "--steps", "20", "--warmup", "3"])
    assert rc == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[0] == sys.executable and "--nproc-per-node=8" in cmd and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    import bench
    seen = []
    monkeypatch.setattr(bench, "self_launch", lambda args, argv: seen.append(argv) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    try:
        bench.main()
    except SystemExit as exc:
        assert exc.code == 0
    assert seen == [["--gpus", "4"]]
    # under a launcher with a mismatching world size the rank refuses instead of launching again
    monkeypatch.setenv("WORLD_SIZE", "2")
    try:
        bench.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as exc:
        assert "WORLD_SIZE=2" in str(exc.code)
    assert len(seen) == 1


# ---- transport agreement of bench.pick_transport under two gloo ranks (no GPU): whatever fails on whichever rank, both ranks leave with the
#      SAME answer and have issued the same torch.distributed collectives (a rank that bailed out early would leave the other one waiting) ----

def _transport_worker(rank, world, port, mode, q):
    import os
    import types
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from scoreperformer_amd import comm
    if mode == "unavailable_on_rank1" and rank == 1:
        def boom():
            raise RuntimeError("no RCCL here")
        comm.available = boom
    elif mode == "id_fails_on_rank0":   # stage 1 passes everywhere, then rank 0 cannot make the RCCL id: the id broadcast must still happen
        comm.available = lambda: None

        def no_id():
            raise RuntimeError("ncclGetUniqueId failed")
        if rank == 0:
            comm.unique_id = no_id
    torch.cuda.synchronize = lambda *a, **k: None            # (the check synchronises the device it does not have here)
    args = types.SimpleNamespace(dp_transport="auto")
    transport, note = bench.pick_transport(args, dist, torch.device("cpu"))
    # a collective AFTER the choice: both ranks must still be in step
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    q.put((rank, transport, note, float(t)))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("mode", ["unavailable_on_rank1", "id_fails_on_rank0"])
def test_transport_choice_is_agreed_by_both_ranks_whatever_fails(mode):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        if p.exitcode is None:
            p.kill()
        assert p.exitcode == 0, mode
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == res[1][1] == "torch", res          # no GPU here: the native transport cannot pass its check on either rank
    assert res[0][3] == res[1][3] == 3.0
