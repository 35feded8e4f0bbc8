"""`python bench.py --gpus N` is self-launching (the reference's DP site forks its own workers,
schaaaafrichter/train.py:159-191): the parent forks N ranks through torch.distributed.run BEFORE anything touches the GPU,
relays rank 0's single JSON line and propagates a failing rank's exit code.

No GPU here, so the ranks run bench.py's dry mode (LOANS_BENCH_DRY=1: same launcher, rendezvous, Communicator and
gradient-arena all-reduce over gloo, no kernels, value = null).  The real step through the same launcher runs in
tests/test_gpu_launch.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, LOANS_BENCH_DRY='1', LOANS_DIST_BACKEND='gloo', **env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e, timeout=300,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


def test_bench_gpus2_forks_its_own_ranks_and_prints_one_line():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['config']['world_size'] == 2 and out['config']['dist_backend'] == 'gloo'
    assert out['value'] is None and 'dry-run' in out['data']          # never mistaken for a measurement


def test_bench_gpus1_stays_a_single_process():
    r = _run(['--gpus', '1', '--steps', '2', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 1 and lines[0]['config']['dist_backend'] is None


def test_failing_rank_fails_the_launch():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], LOANS_BENCH_DRY_FAIL_RANK='1')
    assert r.returncode != 0
    assert not _json_lines(r.stdout)


def test_launcher_parent_decision_needs_no_torch():
    """the fork decision is taken from argv / env alone (loans_amd/launch.py imports neither torch nor the HIP library)"""
    code = ("import sys, importlib.util; "
            "spec = importlib.util.spec_from_file_location('l', %r); m = importlib.util.module_from_spec(spec); "
            "spec.loader.exec_module(m); "
            "assert m.requested_gpus(['--steps', '3', '--gpus', '4']) == 4; "
            "assert m.requested_gpus(['--gpus=8']) == 8; assert m.requested_gpus([]) == 1; "
            "cmd = m.command('bench.py', ['--gpus', '2'], 2, 29500); "
            "assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--master-addr' in cmd and '127.0.0.1' in cmd; "
            "assert 'torch' not in sys.modules and 'loans_amd' not in sys.modules"
            % os.path.join(ROOT, 'loans_amd', 'launch.py'))
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr


def test_rank_env_means_no_second_fork():
    import importlib.util
    spec = importlib.util.spec_from_file_location('l', os.path.join(ROOT, 'loans_amd', 'launch.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    old = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE')}
    os.environ.update(RANK='0', WORLD_SIZE='2')
    try:
        assert m.is_rank()
        assert m.launch_if_parent('bench.py', ['--gpus', '2']) is None        # returns: this process is a rank
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_sigterm_to_the_parent_ends_every_rank():
    """ADVICE (round 2): a scheduler's SIGTERM (or a closed terminal) must not orphan torch.distributed.run and its ranks with
    the GPUs in their hands -- the launcher passes the signal to the ranks' whole process group and exits 128 + signal"""
    import signal
    import time
    import psutil
    e = dict(os.environ, LOANS_BENCH_DRY='1', LOANS_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '100000000', '--warmup', '0'],
                         cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 120
        kids = []
        while time.time() < deadline:               # until both ranks exist (python -m torch.distributed.run -> 2 x bench.py)
            kids = psutil.Process(p.pid).children(recursive=True)
            if sum('bench.py' in ' '.join(k.cmdline()) and 'torch.distributed.run' not in ' '.join(k.cmdline()) for k in kids) >= 2:
                break
            time.sleep(0.5)
        assert len(kids) >= 3, [k.cmdline() for k in kids]
        time.sleep(2.0)
        p.send_signal(signal.SIGTERM)
        rc = p.wait(timeout=60)
        assert rc == 128 + signal.SIGTERM, rc
        gone, alive = psutil.wait_procs(kids, timeout=20)
        assert not alive, [k.cmdline() for k in alive]
    finally:
        if p.poll() is None:
            p.kill()
        for k in psutil.Process().children(recursive=True):
            if 'bench.py' in ' '.join(k.cmdline()):
                k.kill()


def test_a_child_that_ignores_sigterm_is_killed_after_the_grace_period():
    """ADVICE (round 3): the SIGKILL escalation never ran -- the wait was entered without a timeout and, after the handler had
    run, silently re-entered without one.  A child that ignores SIGTERM (a rank hung in a collective behaves the same) must be
    gone `grace_seconds` after the signal, and the launcher must report the kill."""
    import signal
    import time
    child = "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); print('up', flush=True); time.sleep(600)"
    parent = ("import sys; sys.path.insert(0, %r); from loans_amd import launch; "
              "rc, got = launch.run_group([sys.executable, '-c', %r], grace_seconds=1.0); print('rc', rc, 'got', got, flush=True)"
              % (ROOT, child))
    p = subprocess.Popen([sys.executable, '-c', parent], stdout=subprocess.PIPE, text=True)
    try:
        assert p.stdout.readline().strip() == 'up'
        t0 = time.time()
        p.send_signal(signal.SIGTERM)
        out, _ = p.communicate(timeout=30)
        assert time.time() - t0 < 10, 'the grace period did not end the child'
        assert out.strip() == 'rc %d got [%d]' % (-signal.SIGKILL, signal.SIGTERM), out
    finally:
        if p.poll() is None:
            p.kill()


def test_a_signal_between_handler_setup_and_child_start_is_not_lost():
    """the handlers exist before the child does: a signal recorded while Popen is still running is forwarded on the first turn
    of the wait loop (the child below exits 0 on SIGTERM; without the forward it would sleep its 600 s)"""
    import signal
    import time
    from loans_amd import launch
    child = "import signal, sys, time; signal.signal(signal.SIGTERM, lambda *a: sys.exit(0)); time.sleep(600)"
    real_popen = subprocess.Popen

    def popen_then_signal(*a, **kw):
        proc = real_popen(*a, **kw)
        time.sleep(1.0)                                   # let the child install its handler
        os.kill(os.getpid(), signal.SIGTERM)              # arrives "during Popen": only recorded
        return proc
    launch.subprocess.Popen, saved = popen_then_signal, launch.subprocess.Popen
    try:
        t0 = time.time()
        rc, got = launch.run_group([sys.executable, '-c', child], grace_seconds=20.0)
    finally:
        launch.subprocess.Popen = saved
    assert got == [signal.SIGTERM] and rc == 0 and time.time() - t0 < 15
