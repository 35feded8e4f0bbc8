"""The joint LoANs training step (reference sheep/sheep_updater.py:8-68).

``SheepAssessor`` keeps the reference's constructor keywords, ``update_core``
statement order, optimiser names (``opt_gen`` / ``opt_dis``), iterator names
(``main`` / ``real``) and reported keys (``loss_localizer`` / ``loss_dis``).
``SheepUpdater`` is an alias (the name BASELINE.json uses)."""
import os

import torch

from .. import ops
from ..common.utils import DirectionLossCalculator, OutOfImageLossCalculator, Size
from ..functions import mean_squared_error
from ..runtime import training
from ..runtime.core import Variable, report, reporter


# The assessor's own chain (real_images -> loss_dis -> its gradients) shares only read-only inputs with the localizer's
# chain until the two optimiser steps.  With LOANS_CONCURRENT_CHAINS (default on) it is issued on a second stream once the
# forward passes of the localizer's chain are enqueued, so its GEMMs fill the GEMM-free stretches of the localizer's
# backward (the 4-channel crop gradient: 1.2 ms, the stem's pool / BN passes: 1.1 ms at B = 256) and the tails of its
# launches: 54.8 -> 52.9 ms per step.  Results are identical: same kernels, same operands, same order per chain.
CONCURRENT_CHAINS = os.environ.get('LOANS_CONCURRENT_CHAINS', '1') != '0'
# LOANS_EARLY_CHAIN=1 enqueues that chain FIRST (it needs nothing of this step's localizer), in the reference's D(real)-before-
# D(fake) order, so that it runs beside the localizer's FORWARD, whose BN / pooling passes leave the matrix cores idle and have
# no weight gradients to fill them: fp32 B = 256 50.7 -> 49.9 ms, bf16 configs unchanged (round 3).  Not the default: the
# localizer's conv-forward launches then share the GPU with it, and the per-launch durations bench.py's roofline, the rocprofv3
# trace and the PMC passes are built on (8.8 ms per step alone) read 15.8 ms -- a property of the schedule, not of the kernel.
# Superseded by the second weight-gradient stream (ops.wgrad_streams): with it the default order measures 49.9-50.0 ms and this
# one 50.2-50.3 on the same box.
EARLY_CHAIN = os.environ.get('LOANS_EARLY_CHAIN', '0') != '0'
_fork = {}


def _fork_stream(device):
    st = _fork.get(device.index)
    if st is None:
        st = _fork[device.index] = torch.cuda.Stream(device=device)
    return st


class SheepAssessor(training.StandardUpdater):

    def __init__(self, *args, **kwargs):
        self.localizer, self.discriminator = kwargs.pop('models')
        self.anchor_iter = kwargs.pop('anchor_iter', None)
        self.create_pca = kwargs.pop('create_pca', False)
        self.n_components_pca = kwargs.pop('n_components_pca', 2)
        self.pca = None
        self.freeze_discriminator = kwargs.pop('resume_discriminator', None) is not None
        self.localizer_target = kwargs.pop('localizer_target', 1.0)
        # use_graph (not in the reference): after `graph_warmup` eager iterations the whole step -- ~600 kernel launches --
        # is captured once into a hipGraph and replayed; at the reference's default batch of 16 the eager step is bound
        # by host launch overhead, not by the GPU
        self.use_graph = kwargs.pop('use_graph', False)
        self.graph_warmup = kwargs.pop('graph_warmup', 2)
        self._graph = None
        self._segment_split = None      # set while a data-parallel step is being captured in two segments (_capture_segments)

        super().__init__(*args, **kwargs)

        self.regularizers = [
            DirectionLossCalculator(self.localizer.xp),
            OutOfImageLossCalculator(self.localizer.xp)
        ]
        self.regularizers[1].batch_sum_scale = float(self.comm.size) if self.comm is not None else 1.0

    def update_core(self):
        with torch.cuda.device(self.device):
            batch = next(self.get_iterator('real'))
            real_images, labels = self.converter(batch, self.device)[:2]
            batch = next(self.get_iterator('main'))
            fake_images = self.converter(batch, self.device)
            # the step's own launches (the crops, the losses, begin_step's weight preparation) run in the localizer's
            # arithmetic when it has one (Link.set_precision); the assessor's __call__ switches to its own where it differs
            want = self.localizer.__dict__.get('precision') or ops.current_precision()
            with ops.precision(*want):
                if self.use_graph and self._graph_legal():
                    self._graph_step(real_images, labels, fake_images)
                else:
                    self._step(real_images, labels, fake_images)

    def _dist_active(self):
        return self.comm is not None and getattr(self.comm, 'active', False)

    def _graph_legal(self):
        """A single process captures any schedule.  Under data parallel the step is captured in TWO segments with the gradient
        exchange between them (`_capture_segments`), which needs both backward chains complete at one point of the schedule:
        the concurrent chains (the default) or a frozen assessor."""
        if not self._dist_active():
            return True
        return self.freeze_discriminator or (CONCURRENT_CHAINS and ops.CAPTURE_STREAMS and not EARLY_CHAIN)

    def _graph_step(self, real_images, labels, fake_images):
        """Eager for the first iterations (tile autotuning, lazy links, first-launch attributes), then capture the step on
        static input buffers and replay it.  A change of input shapes falls back to eager execution."""
        ins = [t.data if isinstance(t, Variable) else t for t in (real_images, labels, fake_images)]
        g = self._graph
        if g is None:
            if self.iteration < self.graph_warmup:
                return self._step(real_images, labels, fake_images)
            dev = torch.device('cuda', self.device) if isinstance(self.device, int) else self.device
            static = [t.to(dev, torch.float32).contiguous().clone() for t in ins]
            for opt in self.get_all_optimizers().values():
                opt.prepare_capture()
            ops.join_side_stream()
            torch.cuda.synchronize()
            before = dict(reporter.observation)
            if self._dist_active():
                graphs = self._capture_segments(static)
            else:
                graphs = [torch.cuda.CUDAGraph()]
                with torch.cuda.graph(graphs[0]):
                    self._step(*static)
            obs = {k: v for k, v in reporter.observation.items() if before.get(k) is not v}
            g = self._graph = {'graphs': graphs, 'static': static, 'obs': obs, 'shapes': [tuple(t.shape) for t in static]}
        if [tuple(t.shape) for t in ins] != g['shapes']:
            return self._step(real_images, labels, fake_images)
        for s, t in zip(g['static'], ins):
            s.copy_(t, non_blocking=True)
        opts = self._exchanging_optimizers() if len(g['graphs']) > 1 else []
        for opt in self.get_all_optimizers().values():
            opt.begin_replay()                  # this step's bias-corrected rate into device memory, BEFORE the captured Adam reads it
        g['graphs'][0].replay()                 # single process: the whole step; data parallel: everything up to both backwards
        for opt in opts:
            opt.update_begin()                  # the whole active prefix of each gradient arena, RCCL beside RCCL
        for opt in opts:
            opt.exchange_wait()                 # this stream waits for the collectives' streams
        for opt in opts:
            if opt._hooks:                      # chainer's hook point: gradients complete (here: exchanged), step not yet applied
                opt.call_hooks()
        for graph in g['graphs'][1:]:
            graph.replay()                      # the two Adam steps on the summed gradients
        report(g['obs'])

    def _exchanging_optimizers(self):
        opts = [self.get_optimizer('opt_gen')] + ([] if self.freeze_discriminator else [self.get_optimizer('opt_dis')])
        return [o for o in opts if o._exchange_active()]

    def _capture_segments(self, static):
        """hipGraph replay under data parallel (round 4): a collective cannot sit inside the captured step -- the staged exchange
        is driven from host callbacks in the backward, and a graph that bakes in RCCL's kernels ties the capture to one
        communicator state -- so the step is captured as TWO graphs out of one memory pool and the exchange runs between their
        replays: graph 1 = both forward passes and both backward chains (every gradient of both arenas complete), eager
        all-reduce of the two arenas, graph 2 = the two fused Adam steps.  `_step_body` calls the split at that point."""
        graphs = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        pool = torch.cuda.graph_pool_handle()
        opts = self._exchanging_optimizers()
        current = torch.cuda.current_stream()
        stream = torch.cuda.Stream(device=current.device)
        stream.wait_stream(current)

        def split():
            graphs[0].capture_end()
            for opt in opts:                    # what graph 2 records is "the exchange has happened": scale by 1 / world size
                opt._exchanged_from, opt._pending = 0, []
                opt.hooks_by_replay = True      # ... and the replay loop below calls the optimiser hooks between the two graphs
            graphs[1].capture_begin(pool=pool)

        state = {'open': None}

        def split_tracked():
            split()
            state['open'] = graphs[1]

        with torch.cuda.stream(stream):
            graphs[0].capture_begin(pool=pool)
            state['open'] = graphs[0]
            self._segment_split = split_tracked
            try:
                self._step(*static)
            except BaseException:
                # leave no capture open behind an error: the stream would stay in capture mode and every later launch fail
                try:
                    state['open'].capture_end()
                except Exception:
                    pass
                raise
            finally:
                self._segment_split = None
            if state['open'] is not graphs[1]:
                graphs[0].capture_end()
                raise RuntimeError('the step never reached its segment split (both backward chains complete): not captured')
            graphs[1].capture_end()
        current.wait_stream(stream)
        return graphs

    def _step(self, real_images, labels, fake_images):
        ops.begin_step(torch.device('cuda', torch.cuda.current_device()))     # one memset for the step's accumulators
        try:
            self._step_body(real_images, labels, fake_images)
        finally:
            ops.end_step()

    def _step_body(self, real_images, labels, fake_images):
        localizer_optimizer = self.get_optimizer('opt_gen')
        discriminator_optimizer = self.get_optimizer('opt_dis')
        xp = self.localizer.xp
        concurrent = CONCURRENT_CHAINS and not self.freeze_discriminator and \
            (ops.CAPTURE_STREAMS or not torch.cuda.is_current_stream_capturing())

        if not concurrent:
            y_real = self.discriminator(real_images)
        early = concurrent and EARLY_CHAIN

        ops.probe('step begin')
        if early:
            # the assessor's own chain needs nothing of this step's localizer: it is enqueued FIRST, in the reference's
            # order (sheep_updater.py:37: D(real) before D(fake)), and runs beside the localizer's forward, whose BN /
            # pooling passes leave the matrix cores idle
            self.discriminator.cleargrads()
            self.localizer.cleargrads()
            for param in self.discriminator.params():
                param.skip_grad_when_disabled = True
            main = torch.cuda.current_stream()
            fork = _fork_stream(main.device)
            fork.wait_stream(main)
            with torch.cuda.stream(fork):
                y_real = self.discriminator(real_images)
                real_forward_done = torch.cuda.Event()
                real_forward_done.record(fork)
                loss_dis = mean_squared_error(y_real, labels)
                loss_dis.backward()
        x_fake, bboxes = self.localizer(fake_images)
        ops.probe('localizer forward enqueued')
        if early:
            main.wait_event(real_forward_done)     # the BN running statistics of the assessor: D(real)'s update, then D(fake)'s
        y_fake = self.discriminator(x_fake)

        localization_labels = xp.full((len(y_fake), 1), self.localizer_target, dtype=xp.float32,
                                      device=y_fake.data.device)
        loss_localizer = mean_squared_error(y_fake, localization_labels)

        for regularizer in self.regularizers:
            loss_localizer += regularizer.calc_loss(bboxes, Size._make(fake_images.shape[-2:]))

        ops.probe('losses enqueued')
        if concurrent and not early:
            # the reference's second half (sheep_updater.py:55-66), enqueued first and on its own stream; both
            # gradient arenas are cleared before the fork because clearing joins the weight-gradient stream
            self.discriminator.cleargrads()
            self.localizer.cleargrads()
            main = torch.cuda.current_stream()
            fork = _fork_stream(main.device)
            fork.wait_stream(main)
            with torch.cuda.stream(fork):
                y_real = self.discriminator(real_images)
                loss_dis = mean_squared_error(y_real, labels)
                loss_dis.backward()
            # Chainer also computes the assessor's parameter gradients in the localizer's backward and clears them
            # unused (sheep_updater.py:48-51,63); here they would land in the arena the other stream is filling
            for param in self.discriminator.params():
                param.skip_grad_when_disabled = True

        self.discriminator.disable_update()

        if not concurrent:
            self.localizer.cleargrads()
        ops.probe('assessor chain enqueued (fork stream)')
        loss_localizer.backward()
        ops.probe('localizer backward enqueued')
        if self._segment_split is not None:
            # a data-parallel step being captured (_capture_segments): segment 1 ends here with both chains' gradients
            # complete on the capture stream, segment 2 is the two updates on gradients that have been exchanged in between
            assert concurrent or self.freeze_discriminator
            if concurrent:
                main.wait_stream(fork)
            else:
                loss_dis = mean_squared_error(y_real, labels)
            ops.join_side_stream()
            self._segment_split()
            localizer_optimizer.update()
            report({'loss_localizer': loss_localizer})
            self.discriminator.enable_update()
            x_fake.unchain_backward()
            bboxes.unchain_backward()
            if not self.freeze_discriminator:
                self.localizer.cleargrads()
                discriminator_optimizer.update()
            report({'loss_dis': loss_dis})
            ops.probe('step end')
            return
        # Data parallel: the localizer's gradient all-reduce (50 MB) is started here and overlaps the assessor's backward
        # below, which touches neither those gradients nor the localizer's parameters; the Adam step then lands where
        # the reference has it in effect (both updates are independent) -- `overlap` is False on a single GPU
        overlap = self.comm is not None and getattr(self.comm, 'active', False) and not self.freeze_discriminator
        if overlap:
            localizer_optimizer.update_begin()
        else:
            localizer_optimizer.update()
        report({'loss_localizer': loss_localizer})

        self.discriminator.enable_update()

        x_fake.unchain_backward()
        bboxes.unchain_backward()

        if concurrent:
            main.wait_stream(fork)
        else:
            loss_dis = mean_squared_error(y_real, labels)

        if not self.freeze_discriminator:
            if not concurrent:
                self.discriminator.cleargrads()
                if not overlap:
                    self.localizer.cleargrads()
                loss_dis.backward()
            if overlap:
                localizer_optimizer.update()        # waits for the exchange started above
            if overlap or concurrent:
                self.localizer.cleargrads()
            discriminator_optimizer.update()

        report({'loss_dis': loss_dis})
        ops.probe('step end')


SheepUpdater = SheepAssessor
