"""Shared machinery of the HIP-backed drop-in modules: every nn.Parameter is a view of ONE flat
fp32 arena whose layout the HIP library defines, and the forward is a ``torch.autograd.Function``
whose backward launches the HIP backward and hands each parameter its gradient."""
import torch
import torch.nn as nn


class _Holder(nn.Module):
    """Name-space node so parameters get the reference's dotted key names."""


def _register(root, dotted, tensor, is_param=True):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Holder())
        node = getattr(node, p)
    if is_param:
        node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=True))
    else:
        node.register_buffer(parts[-1], tensor)


class _HipFn(torch.autograd.Function):
    """log-probs = HIP forward; backward = HIP backward into the gradient arena."""

    @staticmethod
    def forward(ctx, module, inputs, *params):
        eng = module._engine_for(inputs[0].shape[0], inputs[0].shape[1])
        logp = eng.forward(*inputs, train=True).clone()
        ctx.module, ctx.eng = module, eng
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        eng = ctx.eng
        eng.seed_dlogp(dlogp)
        eng.backward()
        # the fused clip+SGD kernel advances the dropout step counter; on this path (stock torch.optim / skorch loop)
        # nothing else would, and every step would redraw the SAME masks
        eng.rng[1:2].add_(1)
        gv = eng.views(eng.grads)
        dead = ctx.module._dead_params
        grads = tuple(None if name in dead else gv[name].clone() for name in ctx.module._param_names)
        return (None, None) + grads


class ArenaModule(nn.Module):
    """Subclasses set ``_entries`` [(name, shape, offset)], ``_arena_floats`` and implement
    ``_state_order(views)`` (registration order incl. buffers) and ``_make_engine(B, S, old)``."""

    _dead_params = frozenset()

    def _setup_arena(self, entries, total, init_values, recipe=None, recipe_seed=0):
        """init_values: {name: tensor} drawn by the caller (the reference-identical CPU stream), or None with a `recipe`
        {name: (kind, scale[, zero_row])}: the same distributions drawn by ``_materialize`` on whatever device the arena lands
        on, from a generator seeded with `recipe_seed` -- no CPU modules are built and no global RNG is consumed."""
        self._entries = entries
        self._param_names = [n for n, _, _ in entries]
        self._engines = {}
        self._opt_state = None      # (grads, momentum, rng, lr) on the device: ONE set per module, shared by every plan
        self._pending_recipe = None
        if init_values is None:
            assert recipe is not None, "either initial values or an init recipe"
            self._pending_recipe = (dict(recipe), int(recipe_seed))
            self._build(torch.empty(total, dtype=torch.float32))      # untouched pages until it is materialised
        else:
            self._build(torch.zeros(total, dtype=torch.float32), init_values)

    def _materialize(self):
        """Draw the pending init recipe into the arena, on the arena's device (one generator, tensors in layout order)."""
        pend = getattr(self, "_pending_recipe", None)
        if pend is None:
            return
        recipe, seed = pend
        self._pending_recipe = None
        arena = self._arena
        g = torch.Generator(device=arena.device)
        g.manual_seed(seed)
        with torch.no_grad():
            for name, shape, off in self._entries:
                n = 1
                for d in shape:
                    n *= d
                v = arena[off:off + n].view(*shape)
                kind, scale, *rest = recipe[name]
                if kind == "normal":
                    v.normal_(0.0, scale, generator=g)
                elif kind == "uniform":
                    v.uniform_(-scale, scale, generator=g)
                elif kind == "const":
                    v.fill_(scale)
                else:
                    raise ValueError(f"init recipe: unknown kind {kind!r} for {name}")
                if rest and rest[0] is not None:
                    v[rest[0]].zero_()                   # nn.Embedding(padding_idx=...) keeps that row at zero

    def state_dict(self, *args, **kwargs):
        self._materialize()
        return super().state_dict(*args, **kwargs)

    def _shared_state(self):
        """Gradient / momentum arenas, dropout rng {seed, step} and lr live on the module, not on a plan: training with
        more than one sequence length (per-batch padding) must not split the momentum or restart the mask stream."""
        dev = self._arena.device
        st = self._opt_state
        if st is None or st["grads"].device != dev:
            n = self._arena.numel()
            st = self._opt_state = {
                "grads": torch.zeros(n, dtype=torch.float32, device=dev), "momentum": torch.zeros(n, dtype=torch.float32, device=dev),
                "rng": torch.tensor([self.seed, 0], dtype=torch.int64, device=dev), "lr": torch.zeros(1, dtype=torch.float32, device=dev),
                "scalars": torch.zeros(4, dtype=torch.float32, device=dev)}
        return st

    def adam_second_moment(self):
        """Arena-shaped exp_avg_sq buffer of the fused Adam update (allocated on first use)."""
        st = self._shared_state()
        if "exp_avg_sq" not in st:
            st["exp_avg_sq"] = torch.zeros_like(st["momentum"])
        return st["exp_avg_sq"]

    def _state_order(self, views):
        raise NotImplementedError

    def _make_engine(self, B, S, old):
        raise NotImplementedError

    def _build(self, arena, values=None):
        for name in list(self._modules):
            del self._modules[name]
        self._arena = arena
        views = {}
        for name, shape, off in self._entries:
            n = 1
            for s in shape:
                n *= s
            views[name] = arena[off:off + n].view(*shape)
            if values is not None:
                with torch.no_grad():
                    views[name].copy_(values[name])
        for name, tensor, is_param in self._state_order(views):
            _register(self, name, tensor, is_param)

    def to(self, device):
        """Reference contract (transformer.py:50-58, bkp.py:383-386): move, remember the device, return self."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self._arena.device:
            grads = {n: p.grad for n, p in self.named_parameters() if p.grad is not None}
            self._move_buffers(device)
            if getattr(self, "_pending_recipe", None) is not None:      # nothing drawn yet: allocate there, draw there
                self._build(torch.empty(self._arena.numel(), dtype=torch.float32, device=device))
            else:
                self._build(self._arena.detach().to(device))
            for n, p in self.named_parameters():
                if n in grads:
                    p.grad = grads[n].to(device)
            self._engines = {}
        self.device = device
        self._materialize()
        return self

    def _move_buffers(self, device):
        pass

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def cpu(self):
        return self.to("cpu")

    def _engine_for(self, B, S):
        """One HIP plan per sequence length, grown when a larger batch arrives; all share the arenas."""
        if not self._arena.is_cuda:
            raise RuntimeError("%s: the module is on %s -- the HIP path is the only compute path "
                               "(no CPU fallback); call .to('cuda') first" % (type(self).__name__, self._arena.device))
        self._materialize()
        eng = self._engines.get(S)
        if eng is None or eng.cfg.B < B:
            old = eng
            eng = self._make_engine(max(B, old.cfg.B if old is not None else 0), S, self._shared_state())
            self._engines[S] = eng
        return eng

    def engine(self, B, S):
        """Public handle for the fused-step estimator (slnlp.net)."""
        return self._engine_for(B, S)

    def _run(self, inputs):
        if self.training and torch.is_grad_enabled():
            named = dict(self.named_parameters())
            return _HipFn.apply(self, inputs, *[named[n] for n in self._param_names])
        eng = self._engine_for(inputs[0].shape[0], inputs[0].shape[1])
        return eng.forward(*inputs, train=self.training).clone()

    # engines hold ctypes handles: rebuild lazily after copy / pickle (sklearn.clone, checkpoints).  copy.deepcopy
    # and pickle restore every tensor separately, so the parameters would stop aliasing ``_arena`` (the memory the HIP
    # plans compute from): ``__setstate__`` re-creates them as views of the restored arena.
    def __getstate__(self):
        self._materialize()
        d = self.__dict__.copy()
        d["_engines"] = {}
        d["_opt_state"] = None
        d["_param_grads"] = {n: p.grad for n, p in self.named_parameters() if p.grad is not None}
        return d

    def __setstate__(self, state):
        grads = state.pop("_param_grads", {})
        super().__setstate__(state)
        self._engines, self._opt_state = {}, None
        self._build(self._arena.detach())
        for n, p in self.named_parameters():
            if n in grads:
                p.grad = grads[n]

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__getstate__(), memo))
        return new
