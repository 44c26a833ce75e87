"""bench.py's N > 1 entry: `python bench.py --gpus N` with no RANK in the environment must start
N fresh rank processes itself, before anything touches the GPU, and pass their exit code on."""
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_command_is_the_drivers_own_form():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3"]


def test_self_launch_takes_the_launcher_path_and_forwards_the_exit_code(monkeypatch):
    seen = {}

    def fake_run(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.delenv("RANK", raising=False)
    rc = bench.self_launch(2, ["--gpus", "2", "--rehearse-on-one-gpu"], run=fake_run)
    assert rc == 7
    assert "--nproc-per-node=2" in seen["cmd"] and "--rehearse-on-one-gpu" in seen["cmd"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_with_gpus_2_launches_before_importing_torch(tmp_path):
    # A real child process: bench.py --gpus 2 with a stub `torch` package first on the path.  If
    # bench.py imported torch before deciding to launch, or tried to run the benchmark in this
    # process, the stub would make it fail differently; the launcher it starts is the stub's
    # torch.distributed.run, which records its arguments.
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    # (`python -m torch.distributed.run` imports torch/__init__ too, in the CHILD: only an import
    # from bench.py's own process is the failure)
    (tmp_path / "torch" / "__init__.py").write_text(
        "import os, sys, json\n"
        "if os.environ.get('BENCH_LAUNCH_RECORD') and sys.argv[0] != '-m' and 'bench.py' in sys.argv[0]:\n"
        "    raise SystemExit('torch imported in the launcher process')\n")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text(
        "import os, sys, json\n"
        "json.dump(sys.argv[1:], open(os.environ['BENCH_LAUNCH_RECORD'], 'w'))\n"
        "print(json.dumps({'metric': 'stub', 'n_gpus': 2}))\n"
        "sys.exit(5)\n")
    rec = tmp_path / "argv.json"
    env = dict(os.environ, PYTHONPATH=str(tmp_path), BENCH_LAUNCH_RECORD=str(rec))
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--rehearse-on-one-gpu"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 5, p.stderr            # the children's exit code
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 2   # rank 0's line passes through
    argv = json.load(open(rec))
    assert "--nproc-per-node=2" in argv and "--rehearse-on-one-gpu" in argv and "--gpus" in argv
