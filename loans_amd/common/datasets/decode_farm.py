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

    # requests a worker may have outstanding: its stdin pipe holds 64 KiB, a path is at most PATH_MAX = 4096 bytes -- eight
    # always fit, so the parent never blocks writing requests while the worker blocks writing a frame nobody reads (with a
    # whole share written up front, a large batch on few processes with long paths did exactly that: a deadlock without a
    # timeout).  Eight frames ahead is more than the reply pipe holds anyway.
    WINDOW = 8

    def __init__(self, n_processes):
        self.n = max(1, int(n_processes))
        self.procs = []
        self._start()

    def _start(self):
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
        """one worker's share of a batch: at most WINDOW requests outstanding -- a new one goes out for every reply read -- so
        the worker decodes ahead into its pipe and neither side can fill the other's"""
        k, paths = job
        proc = self.procs[k]
        sent = 0

        def send(upto):
            nonlocal sent
            if upto > sent:
                try:
                    proc.stdin.write(b''.join(p.encode('utf-8') + b'\n' for p in paths[sent:upto]))
                    proc.stdin.flush()
                except (BrokenPipeError, ValueError):
                    raise RuntimeError('decode worker %d died (exit code %s)' % (k, proc.poll())) from None
                sent = upto
        out = []
        for i in range(len(paths)):
            send(min(len(paths), i + self.WINDOW))
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
        try:
            results = list(map_fn(self._serve, shares))
        except BaseException:
            # one worker failed: the others' pipes may still hold replies nobody will read, so none of them can be trusted
            # with the next batch -- the whole farm is replaced before the error goes up
            self.close()
            self._start()
            raise
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
