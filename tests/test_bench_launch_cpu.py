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
    # the launcher picks its own free port (no check-then-use race in the parent); every address is 127.0.0.1
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]           # the caller's own flags, minus --dry-launch


def test_self_launch_starts_a_child_and_relays_its_exit_code(monkeypatch):
    import bench
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 7

    monkeypatch.setattr(bench, "run_child", fake_call)
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: (_ for _ in ()).throw(AssertionError("the child runs through run_child")))
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


def test_explicit_port_is_passed_through():
    import bench
    cmd = bench.launch_command(["--gpus", "2"], 2, port=29511)
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511" and "--standalone" not in cmd


def test_run_child_relays_the_exit_code_and_a_sigterm_ends_the_whole_group(tmp_path):
    """The launcher runs in its own process group; a SIGTERM of the parent reaches the launcher AND its ranks (here: a child that starts a
    grandchild), and the child's exit code is what the parent returns."""
    import signal
    import time
    import bench
    assert bench.run_child([sys.executable, "-c", "import sys; sys.exit(5)"], dict(os.environ)) == 5
    marker = tmp_path / "grandchild.pid"
    prog = ("import bench, sys, os\n"
            "code = 'import subprocess, sys, time; p = subprocess.Popen([sys.executable, \"-c\", \"import time; time.sleep(600)\"]); "
            "open(sys.argv[1], \"w\").write(str(p.pid)); time.sleep(600)'\n"
            "sys.exit(bench.run_child([sys.executable, '-c', code, sys.argv[1]], dict(os.environ)))\n")
    parent = subprocess.Popen([sys.executable, "-c", prog, str(marker)], env=_clean_env(), cwd=ROOT)
    for _ in range(300):
        if marker.exists() and marker.read_text().strip():
            break
        time.sleep(0.1)
    grandchild = int(marker.read_text())
    parent.send_signal(signal.SIGTERM)
    rc = parent.wait(60)
    assert rc != 0                                  # the child died of the relayed signal
    for _ in range(100):                            # and so did the grandchild (same process group)
        try:
            os.kill(grandchild, 0)
        except ProcessLookupError:
            break
        # a zombie until its (dead) parent is reaped by init: look at the state instead
        try:
            with open(f"/proc/{grandchild}/stat") as fh:
                if fh.read().split()[2] == "Z":
                    break
        except FileNotFoundError:
            break
        time.sleep(0.1)
    else:
        os.kill(grandchild, signal.SIGKILL)
        raise AssertionError("the grandchild survived the parent's SIGTERM")


def test_default_transport_is_torch(monkeypatch):
    import bench
    monkeypatch.delenv("SPN_DP_TRANSPORT", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    args = bench.parse()
    assert args.dp_transport == "torch" and args.dp_init_timeout > 0
    assert bench.pick_transport(args, object(), None)[0] == "torch"     # no probe, no collective: `dist` is not touched


def test_deadline_expires_with_status_3_and_is_silent_when_met():
    import time
    import bench
    codes = []
    with bench.Deadline(0.2, "unit test", _exit=codes.append):
        time.sleep(0.6)
    assert codes == [3]
    codes.clear()
    with bench.Deadline(5.0, "unit test", _exit=codes.append):
        pass
    time.sleep(0.1)
    assert codes == []


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
    elif mode == "init_hangs_on_rank1":  # a communicator start-up that never returns on one rank: BOTH ranks leave with status 3 in time
        comm.available = lambda: None
        if rank == 1:
            import time
            comm.NativeComm.from_group = classmethod(lambda cls, group=None: time.sleep(600))
    torch.cuda.synchronize = lambda *a, **k: None            # (the check synchronises the device it does not have here)
    args = types.SimpleNamespace(dp_transport="auto", dp_init_timeout=8.0 if mode == "init_hangs_on_rank1" else 120.0)
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


def test_a_hung_native_start_up_ends_every_rank_with_status_3():
    """Rank 1 never returns from the native communicator's start-up; rank 0 waits for it in the id broadcast.  Each rank's own deadline
    fires: both processes exit with status 3 (no hang, no result), well before the test's own limit."""
    import socket
    import time
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, "init_hangs_on_rank1", q)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        if p.exitcode is None:
            p.kill()
            raise AssertionError("a rank was still running long after the deadline")
    assert [p.exitcode for p in procs] == [3, 3]
    assert time.time() - t0 < 140
    assert q.empty()
