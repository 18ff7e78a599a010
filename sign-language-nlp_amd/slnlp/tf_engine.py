"""Host-side owner of one Transformer plan (libslnlp ``slnlp_tf_*``).

PyTorch is plumbing here: it allocates the flat parameter / gradient / momentum
arenas and the activation workspace in HBM and provides the stream; the layout,
the launch sequence and all arithmetic live in the HIP library.
"""
import ctypes as C
import math

import torch

from . import _lib
from .launch import LaunchPolicy
from ._lib import TfBuffers, TfConfig, check, load, ptr, stream_ptr


def positional_table(max_len, d_model):
    """Sinusoidal table, same arithmetic (torch fp32 ops) as
    /root/reference/model/component/positional_encoding.py:27-35 -> bit-identical
    to the reference's ``pe`` buffer.  Shape [max_len, d_model]."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def make_config(E, H, N, F, Vs, Vt, B, S, pad_src=1, pad_tgt=1, dropout=0.0, precision=3):
    return TfConfig(E, H, N, F, Vs, Vt, B, S, pad_src, pad_tgt, float(dropout), precision)


def layout(cfg):
    """[(name, shape tuple, offset in floats)] in reference state_dict order + arena size.
    Pure host query: works without a GPU."""
    lib = load()
    n = lib.slnlp_tf_num_params(C.byref(cfg))
    if n < 0:
        check(1, "tf_num_params")
    out = []
    for i in range(n):
        name = C.create_string_buffer(128)
        shape = (C.c_int64 * 2)()
        ndim, off = C.c_int32(0), C.c_int64(0)
        check(lib.slnlp_tf_param_info(C.byref(cfg), i, name, C.byref(shape), C.byref(ndim), C.byref(off)),
              "tf_param_info")
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value)), int(off.value)))
    return out, int(lib.slnlp_tf_arena_floats(C.byref(cfg)))


class TransformerEngine:
    """One plan = one (config, max batch) on one GPU / one stream."""

    def __init__(self, cfg, device="cuda", seed=0, max_len=5000, params=None, grads=None, momentum=None, pe=None, rng=None, lr=None, scalars=None):
        """``params`` / ``grads`` / ``momentum`` / ``pe``: adopt arenas owned by the caller (the
        drop-in ``model.Transformer`` keeps its nn.Parameters as views of ``params``)."""
        _lib.require_gpu()
        self._alloc_stream = self._last_stream = torch.cuda.current_stream(torch.device(device))   # whose pool the buffers come from
        self.cfg = cfg
        self.device = torch.device(device)
        self.entries, self.arena_floats = layout(cfg)
        dev = self.device
        mk = lambda t: torch.zeros(self.arena_floats, dtype=torch.float32, device=dev) if t is None else t
        self.params, self.grads, self.momentum = mk(params), mk(grads), mk(momentum)
        for t in (self.params, self.grads, self.momentum):
            assert t.is_cuda and t.dtype == torch.float32 and t.numel() == self.arena_floats and t.is_contiguous()
        self.pe = positional_table(max_len, cfg.E).to(dev) if pe is None else pe   # [max_len, E]
        ws = int(load().slnlp_tf_workspace_bytes(C.byref(cfg)))
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=dev)
        # rng = {seed, dropout step counter}; lr: read from device memory by the update kernel.  A module with several
        # plans (one per sequence length) hands every plan the same two tensors
        self.rng = torch.tensor([seed, 0], dtype=torch.int64, device=dev) if rng is None else rng
        self.lr = torch.zeros(1, dtype=torch.float32, device=dev) if lr is None else lr
        self.scalars = torch.zeros(4, dtype=torch.float32, device=dev) if scalars is None else scalars   # {loss, grad norm, Adam step count, -}
        self.logp = torch.empty(cfg.B, cfg.Vt, dtype=torch.float32, device=dev)
        bufs = TfBuffers(ptr(self.params), ptr(self.grads), ptr(self.momentum), ptr(self.pe), ptr(self.workspace),
                         ptr(self.rng), ptr(self.lr), ptr(self.scalars))
        handle = C.c_void_p()
        check(load().slnlp_tf_create(C.byref(cfg), C.byref(bufs), C.byref(handle)), "tf_create")
        self.handle = handle
        # every buffer of this plan is a torch tensor from the stream-ordered caching allocator, and __del__ waits for the
        # plan's last stream when that is not the allocating one: the plan itself needs no device-wide wait when it goes
        # away (which would stall the other host threads' queued work each time a fit ends).  Per plan, not process-wide.
        check(load().slnlp_tf_set_destroy_sync(handle, 0), "tf_set_destroy_sync")
        self._graph_keys = {}
        self._launch = LaunchPolicy()
        self._xbuf = self._ybuf = None
        self._pv = None

    def sync_params_version(self):
        """The fused update keeps the bf16 weight planes current; a write to the arena from the torch side (load_state_dict,
        a torch optimizer, an in-place edit -- all of which move the tensor's version counter, which kernel launches through
        raw pointers do not) makes them stale: tell the plan before the next launch."""
        v = self.params._version
        if v != self._pv:
            check(load().slnlp_tf_params_changed(self.handle), "tf_params_changed")
            self._pv = v

    def _sp(self):
        """Pointer of the stream this call runs on; remembered for the destructor."""
        st = self._last_stream = torch.cuda.current_stream(self.device)
        return st.cuda_stream

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                # the buffers return to the pool of the stream they were allocated on: if the plan last ran on another
                # stream, that work must be over first (same stream: the allocator's stream order covers it)
                ls, al = getattr(self, "_last_stream", None), getattr(self, "_alloc_stream", None)
                if ls is not None and al is not None and ls != al:
                    ls.synchronize()
                load().slnlp_tf_destroy(h)
            except Exception:
                pass
            self.handle = None

    # ---- parameter access ------------------------------------------------
    def views(self, arena=None):
        """name -> tensor view into ``arena`` (default: the parameter arena)."""
        arena = self.params if arena is None else arena
        out = {}
        for name, shape, off in self.entries:
            n = 1
            for s in shape:
                n *= s
            out[name] = arena[off:off + n].view(*shape)
        return out

    def load_state(self, sd):
        v = self.views()
        for k, t in v.items():
            t.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32))

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    # ---- compute -----------------------------------------------------------
    def forward(self, X, y, train=False):
        """X int64 [B,S], y int64 [B] on the device -> log-probs [B,Vt] (a view
        of the engine's output buffer, valid until the next call)."""
        self.sync_params_version()
        B = X.shape[0]
        X = X.contiguous()
        y = y.contiguous()
        self._keep = (X, y)  # backward reads the ids again
        check(load().slnlp_tf_forward(self.handle, ptr(X), ptr(y), B, int(train), ptr(self.logp), self._sp()),
              "tf_forward")
        return self.logp[:B]

    def seed_dlogp(self, dlogp):
        check(load().slnlp_tf_seed_dlogp(self.handle, ptr(dlogp.contiguous()), self._sp()), "tf_seed_dlogp")

    def backward(self):
        check(load().slnlp_tf_backward(self.handle, self._sp()), "tf_backward")

    def optim(self, momentum=0.9, max_norm=0.5):
        check(load().slnlp_tf_optim(self.handle, momentum, max_norm, self._sp()), "tf_optim")

    def optim_adam(self, exp_avg_sq, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.5):
        """clip_grad_norm_ + torch.optim.Adam fused (exp_avg = the momentum arena, exp_avg_sq = ``exp_avg_sq``, step count in
        ``scalars[2]``)."""
        check(load().slnlp_tf_optim_adam(self.handle, ptr(exp_avg_sq), betas[0], betas[1], eps, weight_decay, max_norm, self._sp()),
              "tf_optim_adam")

    def train_step_adam(self, X, y, exp_avg_sq, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.5, lengths=None):
        logp = self.forward(X, y, train=True)            # (lengths: unused by the Transformer, transformer.py:60)
        self.backward()
        self.optim_adam(exp_avg_sq, betas, eps, weight_decay, max_norm)
        return logp

    def train_step(self, X, y, momentum=0.9, max_norm=0.5):
        """Eager fwd + criterion + bwd + clip + SGD; returns log-probs view.
        loss / grad-norm stay on the device in ``scalars[0:2]``."""
        self.sync_params_version()
        B = X.shape[0]
        X = X.contiguous()
        y = y.contiguous()
        self._keep = (X, y)
        check(load().slnlp_tf_train_step(self.handle, ptr(X), ptr(y), B, momentum, max_norm, ptr(self.logp),
                                         self._sp()), "tf_train_step")
        return self.logp[:B]

    def train_step_graph(self, X, y, momentum=0.9, max_norm=0.5):
        """Same step replayed from a captured hipGraph (one per batch size):
        the batch is copied into fixed staging buffers, then one graph launch."""
        self.sync_params_version()
        B = X.shape[0]
        key = (B, float(momentum), float(max_norm))
        if self._xbuf is None:
            self._xbuf = torch.empty(self.cfg.B, self.cfg.S, dtype=torch.int64, device=self.device)
            self._ybuf = torch.empty(self.cfg.B, dtype=torch.int64, device=self.device)
        xb, yb = self._xbuf[:B], self._ybuf[:B]
        xb.copy_(X)
        yb.copy_(y)
        st = self._sp()
        if st == 0:
            raise RuntimeError("train_step_graph needs a non-default stream (use torch.cuda.stream(...))")
        if self._graph_keys.get(B) != key:       # one captured graph per batch size, kept by the plan
            check(load().slnlp_tf_graph_capture_train(self.handle, ptr(xb), ptr(yb), B, momentum, max_norm,
                                                      ptr(self.logp), st), "tf_graph_capture_train")
            self._graph_keys[B] = key
        check(load().slnlp_tf_graph_launch(self.handle, B, st), "tf_graph_launch")
        return self.logp[:B]

    def tap(self, name, rows, cols):
        out = torch.empty(rows, cols, dtype=torch.float32, device=self.device)
        n = C.c_int64(0)
        check(load().slnlp_tf_tap(self.handle, name.encode(), ptr(out), out.numel(), C.byref(n), self._sp()),
              "tf_tap")
        assert n.value == rows * cols, (name, n.value, rows, cols)
        return out

    @property
    def loss(self):
        return float(self.scalars[0])

    @property
    def grad_norm(self):
        return float(self.scalars[1])

    def step(self, X, y, lengths=None, momentum=0.9, max_norm=0.5, graph="auto"):
        """Uniform fused-step entry (estimator): the Transformer ignores ``lengths`` (transformer.py:60).
        graph: True (hipGraph replay) / False (eager launches) / "auto" (time both, keep the faster; launch.py)."""
        if graph == "auto" and self._sp() == 0:
            graph = False                    # graph capture needs a non-default stream
        if graph == "auto":
            return self._launch.run((X.shape[0], float(momentum), float(max_norm)),
                                    lambda: self.train_step_graph(X, y, momentum, max_norm),
                                    lambda: self.train_step(X, y, momentum, max_norm))
        return (self.train_step_graph if graph else self.train_step)(X, y, momentum, max_norm)
