"""Host runtime mirroring the operator / link interface the reference reaches
through Chainer: ``Variable`` with ``backward`` /
``unchain_backward``, old-style ``Function`` (``forward(inputs)`` /
``backward(inputs, grad_outputs)`` on raw arrays, as in the reference's only
user-defined op functions/rotation_droput.py:9-52), ``Link`` / ``Chain`` /
``ChainList`` with ``init_scope``, ``params``, ``cleargrads``,
``disable_update`` / ``enable_update``, and npz (de)serialisation with Chainer
key paths (train_sheep_localizer.py:45-47,182-186).

MI355X-first differences, invisible at that surface:
* arrays are device tensors in NHWC; 4-D parameters are stored OHWI and exposed
  as OIHW *views*, so snapshots interchange with the reference;
* all parameters of a top-level model live in ONE flat arena (data + grad) so
  the optimiser is a single fused kernel launch and data-parallel training is a
  single bucketed RCCL all-reduce;
* parameter gradients are accumulated by the kernels directly into the arena
  (``cleargrads`` = one memset).
"""
import contextlib
import heapq
import weakref

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# global configuration (chainer.config.train / using_config)
# --------------------------------------------------------------------------- #
class _Config:
    train = True
    enable_backprop = True


config = _Config()


@contextlib.contextmanager
def using_config(name, value):
    old = getattr(config, name)
    setattr(config, name, value)
    try:
        yield
    finally:
        setattr(config, name, old)


def no_backprop_mode():
    return using_config('enable_backprop', False)


# --------------------------------------------------------------------------- #
# Variable / Function
# --------------------------------------------------------------------------- #
class Variable:
    """A device array plus the graph edge that produced it."""

    def __init__(self, data=None, name=None, requires_grad=True):
        self.data = data
        self.name = name
        self.grad = None
        self.creator = None
        self.rank = 0
        self.requires_grad = requires_grad

    # array-ish helpers the reference's callers use
    @property
    def array(self):
        return self.data

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def dtype(self):
        return self.data.dtype

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        # read-only slicing of values (evaluation code); not differentiable
        return self.data[idx]

    def __float__(self):
        return float(self.data)

    def __add__(self, other):
        from ..functions.basic import add
        return add(self, other)

    __radd__ = __add__

    def cleargrad(self):
        self.grad = None

    def unchain_backward(self):
        """Cut the graph upstream of this variable (sheep_updater.py:57-58)."""
        stack = [self]
        seen = set()
        while stack:
            v = stack.pop()
            f = v.creator
            v.creator = None
            if f is None or id(f) in seen:
                continue
            seen.add(id(f))
            stack.extend(f.inputs)
            f.inputs = ()
            f.release()

    def backward(self, retain_grad=False):
        """Reverse-mode sweep in decreasing rank order (Chainer's algorithm)."""
        from .. import ops
        if self.creator is None:
            return
        if self.grad is None:
            self.grad = torch.ones_like(self.data)
        grads = {id(self): self.grad}
        keep = {id(self): self}
        owned = set()       # ids whose gradient buffer was made by this sweep (safe to accumulate into in place)
        heap, seen = [], set()

        def push(f):
            if f is not None and id(f) not in seen:
                seen.add(id(f))
                heapq.heappush(heap, (-f.rank, len(seen), f))

        push(self.creator)
        while heap:
            _, _, f = heapq.heappop(heap)
            outs = [o() for o in f.outputs]
            gys = tuple(None if o is None else grads.get(id(o)) for o in outs)
            if all(g is None for g in gys):
                continue
            if f.precision != (ops.COMPUTE, ops.STORAGE):
                with ops.precision(*f.precision):
                    gxs = f.backward(tuple(x.data for x in f.inputs), gys)
            else:
                gxs = f.backward(tuple(x.data for x in f.inputs), gys)
            if not isinstance(gxs, tuple):
                gxs = (gxs,)
            if ops.PROBE_LOG is not None:       # development probes (tools/host_lead.py): an event behind every function's backward
                ops.probe('  backward of ' + type(f).__name__)
            for o in outs:      # gradients of intermediate outputs are consumed
                if o is not None and not retain_grad and o is not self:
                    grads.pop(id(o), None)
            for x, gx in zip(f.inputs, gxs):
                if gx is None:
                    continue
                if isinstance(x, Parameter):
                    x.accumulate_grad(gx)
                    continue
                if not x.requires_grad:
                    continue
                if id(x) in grads:
                    # the tensor that arrived first may be shared: Add.backward hands the SAME gy to both inputs, Reshape /
                    # layout functions return views of theirs.  Accumulating into it in place would also change the other
                    # branch's gradient, so the first accumulation goes into a buffer of this sweep's own.
                    if id(x) not in owned:
                        first = grads[id(x)]
                        buf = ops._empty(first.shape, device=first.device, dtype=first.dtype)
                        buf.copy_(first)
                        grads[id(x)] = buf
                        owned.add(id(x))
                    ops.axpby(1.0, gx.contiguous(), 1.0, grads[id(x)])
                else:
                    grads[id(x)] = gx
                    keep[id(x)] = x
                if x.creator is not None:
                    push(x.creator)
                else:
                    x.grad = grads[id(x)]
        if retain_grad:     # Chainer's backward(retain_grad=True): intermediate variables keep the gradient that reached them
            for i, v in keep.items():
                if i in grads:
                    v.grad = grads[i]


def as_variable(x):
    return x if isinstance(x, Variable) else Variable(x, requires_grad=False)


class Function:
    """Old-style Chainer ``Function``: subclasses implement ``forward(inputs)``
    and ``backward(inputs, grad_outputs)`` on raw device arrays.  A function
    that accumulates parameter gradients itself (every conv / BN block here)
    returns ``None`` in the parameter slots of ``backward``."""

    def __call__(self, *inputs):
        from .. import ops
        inputs = tuple(as_variable(x) for x in inputs)
        self.inputs = inputs
        self.precision = (ops.COMPUTE, ops.STORAGE)       # backward runs in the arithmetic forward ran in (ops.precision)
        outs = self.forward(tuple(x.data for x in inputs))
        if not isinstance(outs, tuple):
            outs = (outs,)
        needs_graph = config.enable_backprop and any(
            x.requires_grad or x.creator is not None for x in inputs)
        rets = []
        for o in outs:
            v = Variable(o, requires_grad=needs_graph)
            if needs_graph:
                v.creator = self
            rets.append(v)
        if needs_graph:
            self.rank = max([x.rank for x in inputs] + [0])
            for v in rets:
                v.rank = self.rank + 1
            self.outputs = [weakref.ref(v) for v in rets]
        else:
            self.inputs = ()
            self.release()
        return rets[0] if len(rets) == 1 else tuple(rets)

    rank = 0
    outputs = ()
    precision = ('f32', 'f32')

    def forward(self, inputs):
        raise NotImplementedError

    def backward(self, inputs, grad_outputs):
        raise NotImplementedError

    def release(self):
        """Drop saved activations (called when the graph is unchained)."""

    def retain_inputs(self, indexes):     # Chainer API compatibility; inputs are always reachable here
        pass


# --------------------------------------------------------------------------- #
# Parameter / Link tree
# --------------------------------------------------------------------------- #
class UpdateRule:
    def __init__(self):
        self.enabled = True


class Parameter(Variable):
    """A trainable array.  Until the owning model is *finalised* onto a device
    the value lives on the host (initialisers run in NumPy like Chainer's);
    afterwards ``data``/``grad`` are views into the model's flat arena.

    ``logical_shape`` is the Chainer shape (OIHW for conv weights);
    ``to_logical`` / ``from_logical`` convert between it and the physical
    (OHWI, channel-padded) storage."""

    def __init__(self, host_physical, logical_shape, to_logical=None, from_logical=None, name=None):
        super().__init__(None, name=name, requires_grad=True)
        self.host = np.ascontiguousarray(host_physical, dtype=np.float32)
        self.physical_shape = tuple(self.host.shape)
        self.logical_shape = tuple(logical_shape)
        self._to_logical = to_logical or (lambda a: a)
        self._from_logical = from_logical or (lambda a: a)
        self.update_rule = UpdateRule()
        self.grad_view = None

    @property
    def size(self):
        return int(np.prod(self.physical_shape))

    def bind(self, data_view, grad_view):
        self.data = data_view
        self.grad_view = grad_view
        self.grad = grad_view
        self.host = None

    def accumulate_grad(self, g):
        from .. import ops
        ops.axpby(1.0, g.contiguous(), 1.0, self.grad_view)

    # ---- logical (Chainer-layout) access for serialisation and tests ----
    def get_logical(self):
        phys = self.host if self.data is None else self.data.detach().cpu().numpy()
        return np.ascontiguousarray(self._to_logical(phys))

    def set_logical(self, array):
        phys = np.ascontiguousarray(self._from_logical(np.asarray(array, dtype=np.float32)), dtype=np.float32)
        assert phys.shape == self.physical_shape, (phys.shape, self.physical_shape)
        if self.data is None:
            self.host = phys
        else:
            self.data.copy_(torch.from_numpy(phys))

    def grad_logical(self):
        from .. import ops
        ops.join_side_stream()
        return np.ascontiguousarray(self._to_logical(self.grad_view.detach().cpu().numpy()))


class Link:
    """Holds named parameters / persistents / child links (chainer.Link + Chain)."""

    def __init__(self):
        self.__dict__['_params'] = []
        self.__dict__['_persistent'] = []
        self.__dict__['_children'] = []
        self.__dict__['_in_scope'] = False
        self.__dict__['name'] = None
        self.__dict__['_arena'] = None
        self.__dict__['_device'] = None
        self.__dict__['precision'] = None

    def __init_subclass__(cls, **kwargs):
        # a link that has a precision of its own (set_precision) runs its __call__ inside that ops.precision scope
        super().__init_subclass__(**kwargs)
        call = cls.__dict__.get('__call__')
        if call is not None and not getattr(call, '_scoped', False):
            def scoped(self, *args, _call=call, **kw):
                want = self.__dict__.get('precision')
                if want is None:
                    return _call(self, *args, **kw)
                from .. import ops
                if want == (ops.COMPUTE, ops.STORAGE):
                    return _call(self, *args, **kw)
                with ops.precision(*want):
                    return _call(self, *args, **kw)
            scoped._scoped = True
            scoped.__doc__, scoped.__name__ = call.__doc__, '__call__'
            cls.__call__ = scoped

    def set_precision(self, compute, storage=None):
        """The arithmetic of THIS model ('f32' | 'bf16' contractions; 'f32' | 'bf16' activations of the residual stages), for
        every link of the tree: replaces the process-wide ops.set_compute_dtype / set_storage_dtype (which stay as the default
        of models that never call this)."""
        from .. import ops
        want = ops.check_precision(compute, storage if storage is not None else ('bf16' if compute == 'bf16' else 'f32'))
        for l in self.links():
            l.__dict__['precision'] = want
        return self

    @contextlib.contextmanager
    def init_scope(self):
        old = self._in_scope
        self.__dict__['_in_scope'] = True
        try:
            yield
        finally:
            self.__dict__['_in_scope'] = old

    def __setattr__(self, name, value):
        if self.__dict__.get('_in_scope'):
            if isinstance(value, Parameter):
                value.name = name
                if name not in self._params:
                    self._params.append(name)
            elif isinstance(value, Link):
                value.__dict__['name'] = name
                if name not in self._children:
                    self._children.append(name)
        object.__setattr__(self, name, value)

    def add_persistent(self, name, value):
        self._persistent.append(name)
        object.__setattr__(self, name, value)

    # ---- traversal ----
    def children(self):
        for n in self._children:
            yield getattr(self, n)

    def namedlinks(self, prefix=''):
        yield prefix or '/', self
        for n in self._children:
            yield from getattr(self, n).namedlinks(prefix + '/' + n)

    def links(self):
        for _, l in self.namedlinks():
            yield l

    def namedparams(self, include_uninit=True, prefix=''):
        for n in self._params:
            yield prefix + '/' + n, getattr(self, n)
        for n in self._children:
            yield from getattr(self, n).namedparams(include_uninit, prefix + '/' + n)

    def params(self, include_uninit=True):
        for _, p in self.namedparams(include_uninit):
            yield p

    def namedpersistents(self, prefix=''):
        for n in self._persistent:
            yield prefix + '/' + n, self, n
        for n in self._children:
            yield from getattr(self, n).namedpersistents(prefix + '/' + n)

    # ---- Chainer verbs used on the hot path ----
    @property
    def xp(self):
        return torch

    @property
    def _device_id(self):
        return None if self._device is None else self._device.index

    def cleargrads(self):
        if self._arena is not None:
            self._arena.zero_grads()

    def disable_update(self):
        for p in self.params():
            p.update_rule.enabled = False

    def enable_update(self):
        for p in self.params():
            p.update_rule.enabled = True

    @property
    def update_enabled(self):
        return any(p.update_rule.enabled for p in self.params())

    def to_gpu(self, device=None):
        self.finalize(device)
        return self

    # ---- arena ----
    def finalize(self, device=None):
        """Move every parameter / persistent of this tree into one flat device arena."""
        if self._arena is not None:
            return self._arena
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        elif isinstance(device, int):
            device = torch.device('cuda', device)
        arena = ParamArena(self, device)
        for l in self.links():
            l.__dict__['_device'] = device
        self.__dict__['_arena'] = arena
        return arena

    @property
    def arena(self):
        return self._arena

    # ---- serialisation with Chainer key paths ----
    def state_dict_chainer(self):
        out = {}
        for k, p in self.namedparams():
            out[k[1:]] = p.get_logical()
        for k, link, n in self.namedpersistents():
            v = getattr(link, n)
            out[k[1:]] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        return out

    def load_state_dict_chainer(self, state, strict=True):
        for k, p in self.namedparams():
            if k[1:] in state:
                p.set_logical(state[k[1:]])
            elif strict:
                raise KeyError(k[1:])
        for k, link, n in self.namedpersistents():
            if k[1:] not in state:
                if strict:
                    raise KeyError(k[1:])
                continue
            cur = getattr(link, n)
            if torch.is_tensor(cur):
                cur.copy_(torch.from_numpy(np.asarray(state[k[1:]], dtype=np.float32)))
            elif isinstance(cur, np.ndarray):
                cur[...] = state[k[1:]]
            else:
                object.__setattr__(link, n, type(cur)(state[k[1:]]))


Chain = Link


class ChainList(Link):
    def __init__(self, *links):
        super().__init__()
        self.__dict__['_list'] = []
        for l in links:
            self.add_link(l)

    def add_link(self, link):
        name = str(len(self._list))
        link.__dict__['name'] = name
        self._list.append(link)
        self._children.append(name)
        object.__setattr__(self, name, link)

    def __getitem__(self, i):
        return self._list[i]

    def __len__(self):
        return len(self._list)

    def children(self):
        return iter(self._list)


class ParamArena:
    """One flat float32 buffer for all parameter values of a model, one for
    their gradients; persistents (BN running statistics) in a third."""

    def __init__(self, root, device):
        self.device = device
        named = list(root.namedparams())
        # parameters of links listed in root.cold_links (e.g. res6 / res7, which only run on frames
        # taller than 224 / 300 px) go to the END of the arena, so that memset / Adam / all-reduce can
        # be restricted to the prefix that is in use (`active_numel`)
        cold = tuple('/' + n + '/' for n in getattr(root, 'cold_links', ()))
        hot = [(k, p) for k, p in named if not k.startswith(cold)]
        self.cold_offsets = {}
        self.params = [p for _, p in hot]
        total = 0
        for p in self.params:
            total += (p.size + 7) // 8 * 8
        for name in getattr(root, 'cold_links', ()):
            self.cold_offsets[name] = total
            grp = [p for k, p in named if k.startswith('/' + name + '/')]
            self.params += grp
            for p in grp:
                total += (p.size + 7) // 8 * 8
        total = 0
        self.offsets = []
        for p in self.params:
            self.offsets.append(total)
            total += (p.size + 7) // 8 * 8          # every view 32-byte aligned: its bf16 shadow (ops._WeightPrep) then is 16-byte aligned
        self.numel = total
        host = np.zeros(total, np.float32)
        for p, o in zip(self.params, self.offsets):
            host[o:o + p.size] = p.host.ravel()
        self.data = torch.from_numpy(host).to(device)
        self.grad = torch.zeros(total, device=device, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            p.bind(self.data[o:o + p.size].view(p.physical_shape), self.grad[o:o + p.size].view(p.physical_shape))
        for _, link, n in root.namedpersistents():
            v = getattr(link, n)
            if isinstance(v, np.ndarray) and v.dtype.kind == 'f':
                object.__setattr__(link, n, torch.from_numpy(v.astype(np.float32)).to(device))
        self.active_numel = total
        self.root = weakref.ref(root)
        self.data16 = None          # bf16 shadow of `data`, made and refreshed by ops._WeightPrep (bf16 arm)
        if device.type == 'cuda':
            from .. import ops
            ops.register_arena(self)

    @property
    def precision(self):
        root = self.root()
        return root.__dict__.get('precision') if root is not None else None

    def set_active(self, first_unused_cold_link=None):
        """Everything from `first_unused_cold_link` on is not touched by the current graph."""
        self.active_numel = self.cold_offsets.get(first_unused_cold_link, self.numel)

    def zero_grads(self):
        from .. import ops
        ops.join_side_stream(self.device)      # weight-gradient kernels still in flight write here
        self.grad[:self.active_numel].zero_()


# --------------------------------------------------------------------------- #
# reporter (chainer.reporter.report)
# --------------------------------------------------------------------------- #
class _Reporter:
    def __init__(self):
        self.observation = {}

    def report(self, values):
        self.observation.update(values)


reporter = _Reporter()


def report(values, observer=None):
    reporter.report(values)


# --------------------------------------------------------------------------- #
# npz snapshot helpers (chainer.serializers.save_npz / NpzDeserializer(strict=False))
# --------------------------------------------------------------------------- #
def save_npz(path, link):
    np.savez(path, **link.state_dict_chainer())


def load_npz(path, link, strict=True):
    with np.load(path) as handle:
        link.load_state_dict_chainer({k: handle[k] for k in handle.files}, strict=strict)
