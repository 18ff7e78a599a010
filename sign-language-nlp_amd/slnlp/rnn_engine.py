"""Host-side owner of one EncoderDecoder{LSTM,GRU}Attn plan (libslnlp ``slnlp_rnn_*``).
Mirrors ``tf_engine.TransformerEngine``: torch allocates arenas / workspace and provides the
stream; layout, launch sequence and arithmetic live in the HIP library."""
import ctypes as C

import torch

from . import _lib
from .launch import LaunchPolicy
from ._lib import RnnConfig, TfBuffers, check, load, ptr, stream_ptr


def make_config(rnn_type, E, Hd, N, Vs, Vt, B, S, pad_src=1, pad_tgt=1, bos_idx=0, dropout=0.0, precision=3):
    assert rnn_type in ("lstm", "gru"), "Invalid `rnn_type`."       # bkp.py:347
    return RnnConfig(int(rnn_type == "lstm"), E, Hd, N, Vs, Vt, B, S, pad_src, pad_tgt, bos_idx, float(dropout), precision)


def layout(cfg):
    """[(name, shape, offset)] in reference state_dict order + arena size; host-only query."""
    lib = load()
    n = lib.slnlp_rnn_num_params(C.byref(cfg))
    if n < 0:
        check(1, "rnn_num_params")
    out = []
    for i in range(n):
        name = C.create_string_buffer(128)
        shape = (C.c_int64 * 2)()
        ndim, off = C.c_int32(0), C.c_int64(0)
        check(lib.slnlp_rnn_param_info(C.byref(cfg), i, name, C.byref(shape), C.byref(ndim), C.byref(off)), "rnn_param_info")
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value)), int(off.value)))
    return out, int(lib.slnlp_rnn_arena_floats(C.byref(cfg)))


class RnnEngine:
    def __init__(self, cfg, device="cuda", seed=0, params=None, grads=None, momentum=None, rng=None, lr=None, scalars=None):
        _lib.require_gpu()
        self._alloc_stream = self._last_stream = torch.cuda.current_stream(torch.device(device))   # whose pool the buffers come from
        self.cfg, self.device = cfg, torch.device(device)
        self.entries, self.arena_floats = layout(cfg)
        dev = self.device
        mk = lambda t: torch.zeros(self.arena_floats, dtype=torch.float32, device=dev) if t is None else t
        self.params, self.grads, self.momentum = mk(params), mk(grads), mk(momentum)
        for t in (self.params, self.grads, self.momentum):
            assert t.is_cuda and t.dtype == torch.float32 and t.numel() == self.arena_floats and t.is_contiguous()
        self.workspace = torch.empty(int(load().slnlp_rnn_workspace_bytes(C.byref(cfg))), dtype=torch.uint8, device=dev)
        # rng = {seed, dropout step counter}; lr: read from device memory by the update kernel.  A module with several
        # plans (one per sequence length) hands every plan the same two tensors
        self.rng = torch.tensor([seed, 0], dtype=torch.int64, device=dev) if rng is None else rng
        self.lr = torch.zeros(1, dtype=torch.float32, device=dev) if lr is None else lr
        self.scalars = torch.zeros(4, dtype=torch.float32, device=dev) if scalars is None else scalars   # {loss, grad norm, Adam step count, -}
        self.logp = torch.empty(cfg.B, cfg.Vt, dtype=torch.float32, device=dev)
        bufs = TfBuffers(ptr(self.params), ptr(self.grads), ptr(self.momentum), None, ptr(self.workspace),
                         ptr(self.rng), ptr(self.lr), ptr(self.scalars))
        handle = C.c_void_p()
        check(load().slnlp_rnn_create(C.byref(cfg), C.byref(bufs), C.byref(handle)), "rnn_create")
        self.handle = handle
        check(load().slnlp_rnn_set_destroy_sync(handle, 0), "rnn_set_destroy_sync")   # torch-allocated buffers: see tf_engine.py
        self._graph_keys = {}
        self._launch = LaunchPolicy()
        self._xbuf = self._ybuf = self._lbuf = None

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
                load().slnlp_rnn_destroy(h)
            except Exception:
                pass
            self.handle = None

    def views(self, arena=None):
        arena = self.params if arena is None else arena
        out = {}
        for name, shape, off in self.entries:
            n = 1
            for s in shape:
                n *= s
            out[name] = arena[off:off + n].view(*shape)
        return out

    def load_state(self, sd):
        for k, t in self.views().items():
            t.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32))

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def forward(self, X, y, lengths, train=False):
        B = X.shape[0]
        self._keep = (X.contiguous(), y.contiguous(), lengths.contiguous())
        X, y, L = self._keep
        check(load().slnlp_rnn_forward(self.handle, ptr(X), ptr(y), ptr(L), B, int(train), ptr(self.logp), self._sp()),
              "rnn_forward")
        return self.logp[:B]

    def seed_dlogp(self, dlogp):
        check(load().slnlp_rnn_seed_dlogp(self.handle, ptr(dlogp.contiguous()), self._sp()), "rnn_seed_dlogp")

    def backward(self):
        check(load().slnlp_rnn_backward(self.handle, self._sp()), "rnn_backward")

    def optim(self, momentum=0.9, max_norm=0.5):
        check(load().slnlp_rnn_optim(self.handle, momentum, max_norm, self._sp()), "rnn_optim")

    def optim_adam(self, exp_avg_sq, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.5):
        """clip_grad_norm_ + torch.optim.Adam fused (exp_avg = the momentum arena, step count in ``scalars[2]``), as
        TransformerEngine.optim_adam."""
        check(load().slnlp_rnn_optim_adam(self.handle, ptr(exp_avg_sq), betas[0], betas[1], eps, weight_decay, max_norm, self._sp()),
              "rnn_optim_adam")

    def train_step_adam(self, X, y, exp_avg_sq, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.5, lengths=None):
        logp = self.forward(X, y, lengths, train=True)
        self.backward()
        self.optim_adam(exp_avg_sq, betas, eps, weight_decay, max_norm)
        return logp

    def train_step(self, X, y, lengths, momentum=0.9, max_norm=0.5):
        B = X.shape[0]
        self._keep = (X.contiguous(), y.contiguous(), lengths.contiguous())
        X, y, L = self._keep
        check(load().slnlp_rnn_train_step(self.handle, ptr(X), ptr(y), ptr(L), B, momentum, max_norm, ptr(self.logp),
                                          self._sp()), "rnn_train_step")
        return self.logp[:B]

    def train_step_graph(self, X, y, lengths, momentum=0.9, max_norm=0.5):
        B = X.shape[0]
        key = (B, float(momentum), float(max_norm))
        if self._xbuf is None:
            dev = self.device
            self._xbuf = torch.empty(self.cfg.B, self.cfg.S, dtype=torch.int64, device=dev)
            self._ybuf = torch.empty(self.cfg.B, dtype=torch.int64, device=dev)
            self._lbuf = torch.empty(self.cfg.B, dtype=torch.int64, device=dev)
        xb, yb, lb = self._xbuf[:B], self._ybuf[:B], self._lbuf[:B]
        xb.copy_(X); yb.copy_(y); lb.copy_(lengths)
        st = self._sp()
        if st == 0:
            raise RuntimeError("train_step_graph needs a non-default stream (use torch.cuda.stream(...))")
        if self._graph_keys.get(B) != key:
            check(load().slnlp_rnn_graph_capture_train(self.handle, ptr(xb), ptr(yb), ptr(lb), B, momentum, max_norm,
                                                       ptr(self.logp), st), "rnn_graph_capture_train")
            self._graph_keys[B] = key
        check(load().slnlp_rnn_graph_launch(self.handle, B, st), "rnn_graph_launch")
        return self.logp[:B]

    def tap(self, name, rows, cols):
        out = torch.empty(rows, cols, dtype=torch.float32, device=self.device)
        n = C.c_int64(0)
        check(load().slnlp_rnn_tap(self.handle, name.encode(), ptr(out), out.numel(), C.byref(n), self._sp()), "rnn_tap")
        assert n.value == rows * cols, (name, n.value, rows, cols)
        return out

    def set_fused_backward(self, on):
        """False: backward through time as the cell kernel + K-sliced grouped GEMM pair (the comparison path of the tests)."""
        check(load().slnlp_rnn_set_fused_backward(self.handle, int(bool(on))), "rnn_set_fused_backward")

    def set_persistent(self, on):
        """True: each encoder layer's timesteps in one persistent launch (opt-in; one fit per GPU only)."""
        check(load().slnlp_rnn_set_persistent(self.handle, int(bool(on))), "rnn_set_persistent")

    def health(self):
        """0 = every device-wide barrier of the persistent layer kernels completed; synchronises."""
        st = C.c_int32(-1)
        check(load().slnlp_rnn_health(self.handle, C.byref(st)), "rnn_health")
        return st.value

    @property
    def loss(self):
        return float(self.scalars[0])

    @property
    def grad_norm(self):
        return float(self.scalars[1])

    def step(self, X, y, lengths, momentum=0.9, max_norm=0.5, graph="auto"):
        """Uniform fused-step entry (estimator).  graph: True / False / "auto" (see launch.py)."""
        if graph == "auto" and self._sp() == 0:
            graph = False                    # graph capture needs a non-default stream
        if graph == "auto":
            return self._launch.run((X.shape[0], float(momentum), float(max_norm)),
                                    lambda: self.train_step_graph(X, y, lengths, momentum, max_norm),
                                    lambda: self.train_step(X, y, lengths, momentum, max_norm))
        return (self.train_step_graph if graph else self.train_step)(X, y, lengths, momentum, max_norm)
