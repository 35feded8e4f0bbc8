"""``chainer.optimizers.Adam(alpha, amsgrad=True)`` (train_sheep_localizer.py:130-134)
as ONE fused kernel launch over the model's flat parameter arena.

Chainer 4.1.0 semantics (oracle/chainer_ops.py:adam_amsgrad_update): eps sits outside
the bias correction, ``lr_t = alpha * sqrt(1 - beta2^t) / (1 - beta1^t)``, parameters
without a gradient are updated with zeros (``reallocate_cleared_grads``): for a parameter
that never had one (m = v = 0) that is a no-op, so the arena's never-used tail -- res6 / res7
below 225 / 301 px -- is skipped; once a parameter HAS been stepped with a gradient its moments
keep decaying and it keeps moving on zero gradients, so the step covers the largest prefix any
update of this optimiser has seen, with the gradients beyond the current graph zeroed.  With a communicator attached
(``parallel.create_multi_node_optimizer``) gradients are all-reduced over RCCL first."""
import math
from types import SimpleNamespace

import torch

from .. import ops


class Adam:

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8, eta=1.0, weight_decay_rate=0, amsgrad=False):
        self.hyperparam = SimpleNamespace(alpha=alpha, beta1=beta1, beta2=beta2, eps=eps, eta=eta,
                                          weight_decay_rate=weight_decay_rate, amsgrad=amsgrad)
        self.t = 0
        self.target = None
        self.comm = None
        self._state = None
        self._hooks = {}
        self._pending, self._exchanged_from, self._plan = [], None, None      # staged gradient exchange (data parallel)

    def setup(self, link):
        self.target = link
        return self

    # chainer.Optimizer.add_hook / remove_hook / call_hooks: a hook is called with the optimiser once per update(), after
    # the gradients are complete (data parallel: after the all-reduce) and before the parameters move -- where Chainer's
    # GradientMethod.update calls them.  The parity tests read every step's gradients through one.
    def add_hook(self, hook, name=None):
        if not callable(hook):
            raise TypeError('hook function is not callable')
        name = name or getattr(hook, 'name', None) or getattr(hook, '__name__', None) or 'hook%d' % len(self._hooks)
        if name in self._hooks:
            raise KeyError('hook %s already exists' % name)
        self._hooks[name] = hook

    def remove_hook(self, name):
        del self._hooks[name]

    def call_hooks(self):
        for hook in list(self._hooks.values()):
            hook(self)

    @property
    def alpha(self):
        return self.hyperparam.alpha

    @alpha.setter
    def alpha(self, v):
        self.hyperparam.alpha = v

    @property
    def lr(self):
        hp = self.hyperparam
        if self.t == 0:
            raise RuntimeError("Can't determine the learning rate of Adam optimizer because the update steps have not been started.")
        fix1 = 1. - math.pow(hp.beta1, self.t)
        fix2 = 1. - math.pow(hp.beta2, self.t)
        return hp.alpha * math.sqrt(fix2) / fix1

    def _ensure_state(self):
        arena = self.target.arena
        if arena is None:
            arena = self.target.finalize()
        if self._state is None or self._state[0].numel() != arena.numel:
            z = lambda: torch.zeros(arena.numel, device=arena.device, dtype=torch.float32)   # noqa: E731
            self._state = (z(), z(), z())
        return arena

    def update(self, lossfun=None, *args, **kwds):
        if lossfun is not None:
            # chainer.GradientMethod.update(lossfun, *args): evaluate, clear, differentiate, then the step below
            # (the LoANs updater calls update() bare, sheep_updater.py:52,66)
            loss = lossfun(*args, **kwds)
            self.target.cleargrads()
            loss.backward()
            del loss
        if not any(p.update_rule.enabled for p in self.target.params()):
            return
        arena = self._ensure_state()
        grad_scale = 1.0
        if self._exchanged_from is not None:    # stage_done() / update_begin() already started the exchange of this step
            if self._exchanged_from > 0:
                self._exchange(arena, 0)        # what the stage boundaries left (stem .. res3, or everything below the last one)
            for work in self._pending:
                work.wait()                     # the current stream waits for RCCL's
            self._pending, self._exchanged_from = [], None
            ops.join_side_stream(arena.device)
            grad_scale = 1.0 / self.comm.size
        else:
            ops.join_side_stream(arena.device)  # all weight gradients of this step have landed
            if self.comm is not None and getattr(self.comm, 'active', self.comm.size > 1):
                self.comm.allreduce_grad(arena)
                grad_scale = 1.0 / self.comm.size
        self.grad_scale = grad_scale        # data parallel: the arena holds the SUM over ranks, the kernel applies 1 / world size
        if self._hooks:
            if arena.grad.is_cuda and torch.cuda.is_current_stream_capturing():
                # hooks run on the host.  A data-parallel step is captured in two segments with the exchange between their
                # replays (sheep_updater._capture_segments): there the replay loop calls the hooks at that point of every step
                if not getattr(self, 'hooks_by_replay', False):
                    raise RuntimeError('optimizer hooks run on the host: not inside a captured step')
            else:
                self.call_hooks()
        hp = self.hyperparam
        m, v, vhat = self._state
        # parameters beyond the prefix the current graph touches have no gradient.  Those that never had one are skipped
        # (m = v = 0: Chainer's zero-gradient step is a no-op); those that were trained before -- the frame height crossed
        # 224 / 300 px between steps -- are stepped with a zero gradient like Chainer does (their m decays, they keep moving)
        n = arena.active_numel
        seen = max(getattr(self, '_stepped_numel', 0), n)
        if seen > n:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('the active parameter prefix shrank inside a captured step')
            arena.grad[n:seen].zero_()
            n = seen
        self._stepped_numel = seen
        if torch.cuda.is_current_stream_capturing():
            # being recorded into a hipGraph (SheepAssessor(use_graph=True)): nothing executes now, and the replays
            # must not bake in this step's rate -- the kernel reads it from device memory, `begin_replay()` advances it
            if getattr(self, '_lr_dev', None) is None:
                raise RuntimeError('call prepare_capture() before recording Adam.update() into a graph')
            lr = self._lr_dev
            self._captured = True
        else:
            self.t += 1
            lr = self.lr
        ops.adam_amsgrad(arena.data[:n], arena.grad[:n], m[:n], v[:n], vhat[:n] if hp.amsgrad else None, lr, hp.beta1, hp.beta2,
                         hp.eps, hp.eta, hp.weight_decay_rate, grad_scale)

    def _exchange_active(self):
        return self.comm is not None and getattr(self.comm, 'active', self.comm.size > 1) and \
            any(p.update_rule.enabled for p in self.target.params())

    def _exchange(self, arena, lo):
        """Start the all-reduce of the gradient floats [lo, what has been started already) and return.  It is issued from the
        weight-gradient stream after that stream has caught up with the current one: RCCL's stream then waits for every
        gradient kernel enqueued so far -- weight gradients (side stream) and the BN gamma / beta sums (current stream) --
        and for nothing that is enqueued afterwards; neither stream waits for the collective."""
        hi = arena.active_numel if self._exchanged_from is None else self._exchanged_from
        if lo < hi:
            if arena.grad.is_cuda:
                main = torch.cuda.current_stream(arena.device)
                side = ops._side_stream(arena.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._pending += self.comm.allreduce_range(arena, lo, hi, async_op=True)
            else:
                self._pending += self.comm.allreduce_range(arena, lo, hi, async_op=True)
        self._exchanged_from = min(lo, hi)

    def stage_done(self, name):
        """A backbone's StageBoundary reports that the backward has left stage ``name``: every gradient from that stage's
        first float to the end of the arena's active prefix is complete (parallel.exchange_plan) and its exchange starts now,
        beside the backward of the stages in front of it (attached by parallel.create_multi_node_optimizer)."""
        if not self._exchange_active():
            return
        arena = self._ensure_state()
        if arena.grad.is_cuda and torch.cuda.is_current_stream_capturing():
            return
        if self._plan is None:
            from .. import parallel
            self._plan = parallel.exchange_plan(self.target)
        lo = self._plan.get(name)
        if lo is not None and lo < arena.active_numel:
            self._exchange(arena, lo)

    def update_begin(self):
        """Data parallel only: start the all-reduce of whatever part of the gradient arena the stage boundaries have not
        started yet and return, so that it runs beside whatever the caller issues next; the following ``update()`` waits for
        all parts and applies the step.  The gradients must not be touched (no ``cleargrads``) in between.  Without an active
        communicator this is a no-op."""
        if not self._exchange_active():
            return
        self._exchange(self._ensure_state(), 0)

    def exchange_wait(self):
        """Make the current stream wait for every exchange started so far and forget them -- for a caller that applies the
        step itself (the captured update of a data-parallel hipGraph, sheep_updater._capture_segments: the graph recorded
        "already exchanged", so nothing inside it waits)."""
        for work in self._pending:
            work.wait()
        self._pending, self._exchanged_from = [], None

    def prepare_capture(self):
        """Allocate what a captured update reads at replay time -- outside the capture, so that neither the buffer nor
        its initialisation belongs to the graph."""
        arena = self._ensure_state()
        if getattr(self, '_lr_dev', None) is None:
            self._lr_dev = torch.zeros(1, device=arena.device, dtype=torch.float32)

    def begin_replay(self):
        """Before replaying a captured step: count it and publish its bias-corrected rate to the kernel."""
        if getattr(self, '_captured', False):
            self.t += 1
            self._lr_dev.fill_(self.lr)
