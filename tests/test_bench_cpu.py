"""bench.py's launcher side (no GPU): `python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts its own ranks as a child
process -- the command it builds, what it relays, and that it decides before anything imports torch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_self_launch_command_is_one_rank_per_gpu_on_localhost():
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "20", "--warmup", "3"], 8, 29544)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29544"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]     # the ranks get the very same arguments
    assert all(not c.startswith("--") or c in ("--nnodes=1", "--nproc-per-node=8", "--master-addr", "--master-port") for c in cmd[3:script])


def test_self_launch_relays_rank0_line_and_exit_code(monkeypatch, capfd):
    line = {"metric": "join rows/sec, 1Bx1B int64 radix hash join", "value": 1.0, "n_gpus": 2}
    child = ("import sys, json; print('noise from a rank'); print(json.dumps(%r)); sys.stderr.write('warn\\n'); sys.exit(%%d)" % (line,))
    for rc in (0, 7):
        monkeypatch.setattr(bench, "self_launch_command", lambda argv, n, port, rc=rc: [sys.executable, "-c", child % rc])
        got = bench.self_launch(type("A", (), {"gpus": 2})(), ["--gpus", "2"])
        out, err = capfd.readouterr()
        assert got == rc
        assert [json.loads(ln) for ln in out.splitlines() if ln.strip()] == [line]      # exactly one line on stdout
        assert "noise from a rank" in err


def test_self_launch_without_a_line_is_a_failure(monkeypatch, capfd):
    monkeypatch.setattr(bench, "self_launch_command", lambda argv, n, port: [sys.executable, "-c", "print('nothing useful')"])
    assert bench.self_launch(type("A", (), {"gpus": 2})(), []) != 0
    capfd.readouterr()


def test_main_launches_before_torch_is_imported():
    """`python bench.py --gpus 2` (no WORLD_SIZE): the decision to launch is taken with neither torch nor the library loaded --
    this process must never have touched the GPU when it starts its child."""
    probe = (
        "import sys, os; sys.argv = ['bench.py', '--gpus', '2', '--rehearse', '--rows', '1000']; os.environ.pop('WORLD_SIZE', None)\n"
        "import bench\n"
        "seen = {}\n"
        "def fake(args, argv):\n"
        "    seen['torch'] = 'torch' in sys.modules; seen['lib'] = 'gpu_olap_engine_amd' in sys.modules; seen['argv'] = argv; return 0\n"
        "bench.self_launch = fake\n"
        "try:\n    bench.main()\nexcept SystemExit as e:\n    seen['rc'] = e.code\n"
        "print(seen)\n")
    out = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    seen = eval(out.stdout.strip().splitlines()[-1])
    assert seen == {"torch": False, "lib": False, "argv": ["--gpus", "2", "--rehearse", "--rows", "1000"], "rc": 0}


def test_an_external_launcher_is_left_alone():
    """under torch.distributed.run WORLD_SIZE is set: no second launch; a --gpus that contradicts it is refused with a message."""
    probe = ("import sys, os; sys.argv = ['bench.py', '--gpus', '4']; os.environ['WORLD_SIZE'] = '2'\n"
             "import bench\n"
             "bench.self_launch = lambda *a: (_ for _ in ()).throw(AssertionError('launched again'))\n"
             "try:\n    bench.main()\nexcept SystemExit as e:\n    print('EXIT', e.code)\n")
    out = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "launched again" not in out.stderr
    assert "WORLD_SIZE=2" in out.stdout + out.stderr
