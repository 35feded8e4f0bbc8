"""Self-contained N-GPU launch: ``python bench.py --gpus 8`` / ``python train_sheep_localizer.py --gpus 8`` fork
their own ranks, one process per GPU.

The reference's one data-parallel call site forks its workers itself (schaaaafrichter/train.py:159-191: Chainer's
``MultiprocessParallelUpdater`` forks one process per device from the trainer process).  Here the parent starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <args>``
as a CHILD process -- the same command line the driver uses when it wraps the script itself -- waits for it, relays its
output and exits with its code.  Two rules of the GPU pool shape this:

* the parent never touches the GPU (this module imports neither torch nor the HIP library; the check runs before the
  script's own imports), so the ranks are the only processes holding a device;
* nothing is ever re-exec'ed: a process that has initialised HIP must not be replaced by another program.

A script that already runs under a launcher (``RANK`` / ``WORLD_SIZE`` in the environment) is a rank and returns at once.
"""
import os
import signal
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def is_rank():
    """True when this process was started by a distributed launcher (torch.distributed.run sets both)."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def requested_gpus(argv, flag='--gpus', default=1):
    """The value of ``--gpus N`` / ``--gpus=N`` in argv without building the script's argument parser (which imports
    torch): the decision to fork has to be taken before anything else is loaded."""
    n = default
    for i, a in enumerate(argv):
        if a == flag and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith(flag + '='):
            n = int(a.split('=', 1)[1])
    return n


def command(script, argv, n, port=None):
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
            '--master-addr', '127.0.0.1', '--master-port', str(port or _free_port()), script] + list(argv)


def launch_if_parent(script, argv=None, flag='--gpus'):
    """Call first thing in ``__main__``.  Returns (the caller goes on as a rank, or as the single process of an N = 1
    run) or does not return: for ``--gpus N`` > 1 outside a launcher it runs the N ranks as a child process group,
    waits, and exits with the group's exit code."""
    argv = list(sys.argv[1:] if argv is None else argv)
    n = requested_gpus(argv, flag)
    if n <= 1 or is_rank():
        return
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # this pool's driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    # the ranks live in a session of their own: a signal that ends the parent (Ctrl-C, a scheduler's SIGTERM, a closed
    # terminal) is passed on to the WHOLE group -- torch.distributed.run and every rank -- so that no rank is left holding a GPU
    rc, got = run_group(command(os.path.abspath(script), argv, n), env)
    sys.stdout.flush()
    if got:
        sys.exit(128 + got[0])
    sys.exit(rc if rc >= 0 else 128 - rc)


def run_group(cmd, env=None, grace_seconds=None):
    """Run ``cmd`` as a process group of its own and wait for it.  SIGINT / SIGTERM / SIGHUP that reach THIS process are passed
    on to the whole group; a group that has not wound down ``grace_seconds`` after the first signal (a rank hung in a
    collective, a child that ignores SIGTERM) gets SIGKILL.  Returns (exit code, [signals received]).

    The handlers are in place BEFORE the child exists and only record the signal: with them installed after Popen a signal
    in between killed the parent and orphaned the new session; and a handler that signals from inside a blocking
    ``proc.wait()`` never gets the wait re-entered with a timeout (PEP 475 retries waitpid), so the grace period never ran.
    The wait is a short poll instead, and everything -- forwarding, the clock, the escalation -- happens in this loop."""
    import time
    grace = GRACE_SECONDS if grace_seconds is None else grace_seconds
    got = []
    sigs = (signal.SIGINT, signal.SIGTERM, signal.SIGHUP)
    old = {sig: signal.signal(sig, lambda signum, frame: got.append(signum)) for sig in sigs}
    try:
        proc = subprocess.Popen(cmd, env=env, start_new_session=True)
        forwarded, t_first, killed = 0, None, False
        while True:
            try:
                rc = proc.wait(timeout=0.2)
                break
            except subprocess.TimeoutExpired:
                pass
            while forwarded < len(got):                     # pass on what arrived since the last look
                _signal_group(proc, got[forwarded])
                forwarded += 1
                t_first = t_first if t_first is not None else time.monotonic()
            if t_first is not None and not killed and time.monotonic() - t_first > grace:
                _signal_group(proc, signal.SIGKILL)         # signalled, and the group did not wind down in time
                killed = True
        if got:
            _signal_group(proc, signal.SIGKILL)             # the leader is gone: no member of its group may outlive the launch
    finally:
        for sig, handler in old.items():
            signal.signal(sig, handler)
    return rc, got


GRACE_SECONDS = 15.0


def _signal_group(proc, signum):
    try:
        os.killpg(proc.pid, signum)             # start_new_session: the child's pid is its process-group id
    except (ProcessLookupError, PermissionError):
        pass
