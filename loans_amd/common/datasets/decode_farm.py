"""A farm of decode PROCESSES for the training input path.

The reference feeds its loop through ``chainer.iterators.MultithreadIterator`` (train_sheep_localizer.py:115-116).  Measured
here (tools/feed_bench.py): Pillow's JPEG / PNG decode holds the GIL, so a thread pool tops out near ONE core's rate (~500
frames/s of 480 x 640 JPEG whatever the pool size) while one MI355X consumes 5 000 frames/s -- the step would wait for its
input nine tenths of the time.  Decode therefore runs in N small worker processes (``_decode_worker.py``: Pillow + NumPy only,
started as plain child programs, no fork of a process that holds a GPU); a thread per worker moves requests and pixels over
pipes (blocking I/O releases the GIL), everything else -- random draws, upload, GPU stages -- stays where it was."""
import os
import struct
import subprocess
import sys

import numpy

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_decode_worker.py')


class DecodeFarm:

    def __init__(self, n_processes):
        self.n = max(1, int(n_processes))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        self.procs = [subprocess.Popen([sys.executable, '-u', _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                      for _ in range(self.n)]
        for p in self.procs:        # a pipe that holds a whole frame lets a worker decode ahead of the thread that drains it
            try:
                import fcntl
                fcntl.fcntl(p.stdout.fileno(), 1031, 1 << 20)          # F_SETPIPE_SZ (Linux)
            except Exception:
                pass

    def _serve(self, job):
        """one worker's share of a batch: send every path, then read every reply (the worker decodes ahead into its pipe)"""
        k, paths = job
        proc = self.procs[k]
        proc.stdin.write(b''.join(p.encode('utf-8') + b'\n' for p in paths))
        proc.stdin.flush()
        out = []
        for p in paths:
            head = proc.stdout.read(12)
            if len(head) != 12:
                raise RuntimeError('decode worker %d died (exit code %s)' % (k, proc.poll()))
            status, H, W = struct.unpack('<iii', head)
            if status != 0:
                out.append(status)
                continue
            frame = numpy.empty((H, W, 3), numpy.uint8)
            view, got = memoryview(frame).cast('B'), 0
            while got < len(view):
                n = proc.stdout.readinto(view[got:])
                if not n:
                    raise RuntimeError('decode worker %d died mid-frame' % k)
                got += n
            out.append(frame)
        return out

    def decode(self, paths, map_fn):
        """uint8 HWC RGB frames of ``paths`` in order (an int status instead of a frame where the worker declined: 1 = not an
        8-bit image, 2 = unreadable).  ``map_fn``: the ``map`` of a thread pool with at least ``n`` threads."""
        paths = list(paths)
        shares = [(k, paths[k::self.n]) for k in range(self.n) if paths[k::self.n]]
        results = list(map_fn(self._serve, shares))
        out = [None] * len(paths)
        for (k, _), frames in zip(shares, results):
            out[k::self.n] = frames
        return out

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.procs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
