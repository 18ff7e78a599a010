"""Oracle (test infrastructure): the per-batch training step skorch runs for
the reference, restated explicitly.

Step = zero_grad -> module(X, y, lengths) -> CrossEntropyLoss(ignore_index=pad)
-> backward -> clip_grad_norm_(0.5) -> SGD(momentum .9, nesterov False).step()
(/root/reference/helper.py:61-70 criterion ignore_index, :227-229 gradient
clipping callback, config/config-transformer.yaml:19-20,36-43).  Gradients
come from torch autograd over the explicit-arithmetic forwards in
``transformer_ref`` / ``rnn_ref``; loss, clipping and the momentum update are
written out by hand below.
"""
import torch


def cross_entropy_on_logprobs(logp, y, ignore_index):
    """torch.nn.CrossEntropyLoss applied to the model's *log-probs*
    (log_softmax is applied a second time -- idempotent up to rounding);
    mean over targets != ignore_index."""
    lsm = torch.log_softmax(logp, dim=-1)
    keep = (y != ignore_index)
    picked = lsm[torch.arange(y.shape[0]), y.clamp(min=0)]
    return -(picked * keep.to(lsm.dtype)).sum() / keep.sum().to(lsm.dtype)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(norm_type=2): returns (total_norm, coef);
    scales ``grads`` in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total, coef


def sgd_momentum_step(params, grads, bufs, lr, momentum=0.9):
    """torch.optim.SGD(dampening=0, nesterov=False, weight_decay=0): the first
    step initialises buf = grad, which equals momentum * 0 + grad."""
    for k in params:
        if grads.get(k) is None:
            continue
        bufs[k] = momentum * bufs[k] + grads[k] if k in bufs else grads[k].clone()
        params[k] = params[k] - lr * bufs[k]


class Trainer:
    """Holds parameters (dict name -> tensor), momentum buffers, and steps."""

    def __init__(self, sd, forward_fn, pad_tgt=1, lr=0.01, momentum=0.9,
                 max_norm=0.5, frozen=()):
        self.sd = {k: v.clone().float() for k, v in sd.items()}
        self.forward_fn = forward_fn  # (sd, X, y, lengths) -> logp
        self.pad_tgt, self.lr, self.momentum, self.max_norm = pad_tgt, lr, momentum, max_norm
        self.bufs = {}
        self.frozen = set(frozen)     # params that never receive grad (dead weights)

    def loss_and_grads(self, X, y, lengths):
        leaves = {k: v.clone().requires_grad_(True) for k, v in self.sd.items()}
        logp = self.forward_fn(leaves, X, y, lengths)
        loss = cross_entropy_on_logprobs(logp, y, self.pad_tgt)
        loss.backward()
        grads = {k: (None if (v.grad is None or k in self.frozen) else v.grad.detach().clone())
                 for k, v in leaves.items()}
        return loss.detach(), logp.detach(), grads

    def step(self, X, y, lengths):
        loss, logp, grads = self.loss_and_grads(X, y, lengths)
        live = [g for g in grads.values() if g is not None]
        total, _ = clip_grad_norm(live, self.max_norm) if self.max_norm else (None, None)
        sgd_momentum_step(self.sd, grads, self.bufs, self.lr, self.momentum)
        return loss, total, logp
