"""Training step of the MuZero models (mlp_model, lstm_model, vision_model) on the MI355X: host side of `Muzero.train`
(/root/reference muzero_model.py:791-811).

Host work (this file): reshape_batch (muzero_model.py:440-472), the learning-rate schedulers
(muzero_model.py:423-433, stepped with `epoch=count`, :771) and the data-parallel exchange.
Device work (muzero_amd/csrc/mzs_train.h through the C-ABI): forward, losses, backward,
weight gradients, optimizer.

Data parallelism: one process per GPU; every rank runs `mzs_train_grad` on its shard of the
batch with `inv_global_batch = 1 / global batch size`, the flat fp32 bucket (the gradient, 12 734
floats for the CartPole MLP, with the 3K loss scalars behind it) is summed with ONE `all_reduce`
(RCCL over xGMI under the "nccl" backend, gloo in the CPU tests) and every rank applies the same
optimizer step, so replicas
that start identical stay bit-identical.  They start identical because Trainer broadcasts rank
0's weights once at construction (every process draws its initial weights from its own torch
generator; shard.broadcast_blob).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

OPTIMIZERS = {"adam": _lib.OPT_ADAM, "sgd": _lib.OPT_SGD}
SCHEDULERS = (None, "none", "steplr", "cosineannealinglr", "cosineannealinglrwarmrestarts", "onecyclelr")


def _one_cycle(max_lr, total_steps, step):
    """torch.optim.lr_scheduler.OneCycleLR(max_lr, total_steps) at step_num = step: (lr, momentum).
    Defaults: pct_start 0.3, cosine annealing, div_factor 25, final_div_factor 1e4, momentum cycled
    0.95 -> 0.85 -> 0.95 (Adam: beta1), two phases."""
    total = int(total_steps)
    if step > total:                 # torch raises when step_num > total_steps (the step AT total still exists)
        raise ValueError(f"Tried to step {step} times. The specified number of total steps is {total}")
    initial, base_m, max_m = max_lr / 25.0, 0.85, 0.95
    phases = ((0.3 * total - 1.0, initial, max_lr, max_m, base_m),
              (total - 1.0, max_lr, initial / 1e4, base_m, max_m))
    anneal = lambda a, b, pct: b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)  # noqa: E731
    start = 0.0
    for i, (end, lr0, lr1, m0, m1) in enumerate(phases):
        if step <= end or i == len(phases) - 1:
            pct = (step - start) / (end - start)
            return anneal(lr0, lr1, pct), anneal(m0, m1, pct)
        start = end


def scheduled_momentum(scheduler, num_of_epoch, count):
    """Adam beta1 / SGD momentum the reference's optimizer holds when train() number `count` runs:
    0.9 (muzero_model.py:418-421) unless OneCycleLR cycles it (same lag as scheduled_lr)."""
    if scheduler != "onecyclelr":
        return 0.9
    return _one_cycle(1.0, num_of_epoch, max(count - 1, 0))[1]


def scheduled_lr(scheduler, base_lr, num_of_epoch, count):
    """Learning rate the reference's optimizer holds when train() number `count` (0-based) runs.

    The reference calls `scheduler.step(epoch=self.count)` AFTER `optimizer.step()` and before
    `count += 1` (muzero_model.py:764-786), which sets the closed-form value for epoch = count;
    so step 0 runs at the base rate and step i > 0 at closed_form(i - 1)."""
    if scheduler == "onecyclelr":                  # OneCycleLR(max_lr=lr, total_steps=num_of_epoch): the constructor
        return _one_cycle(float(base_lr), num_of_epoch, max(count - 1, 0))[0]   # already sets the rate of step 0
    if scheduler in (None, "none") or count == 0:
        return float(base_lr)
    e = count - 1
    if scheduler == "steplr":                      # StepLR(step_size=1, gamma=0.99)
        return float(base_lr) * 0.99 ** e
    if scheduler == "cosineannealinglr":           # CosineAnnealingLR(T_max=num_of_epoch), eta_min 0
        return float(base_lr) * (1.0 + math.cos(math.pi * e / int(num_of_epoch))) / 2.0
    if scheduler == "cosineannealinglrwarmrestarts":  # CosineAnnealingWarmRestarts(T_0=num_of_epoch)
        t0 = int(num_of_epoch)
        return float(base_lr) * (1.0 + math.cos(math.pi * (e % t0) / t0)) / 2.0
    raise NotImplementedError(f"lr_scheduler={scheduler!r}: the reference knows {SCHEDULERS[2:]} (muzero_model.py:424)")


def reshape_batch(batches, obs_dim, n_actions, unroll_steps):
    """muzero_model.py:440-472 on numpy: (batch, importance_sampling_ratio, game_pos) ->
    obs [B, obs], actions [K, B, A], target value / reward [K, B], target policy [K, B, A],
    is_ratio [B] or None (a single-element ratio means "no importance weighting", :730)."""
    batch, ratio, game_pos = batches
    B, K = len(batch), int(unroll_steps)
    obs = np.empty((B, obs_dim), np.float32)
    actions = np.zeros((K, B, n_actions), np.float32)
    tv = np.zeros((K, B), np.float32)
    tr = np.zeros((K, B), np.float32)
    tp = np.zeros((K, B, n_actions), np.float32)
    for b, (o, acts, targets) in enumerate(batch):
        o = o.detach().cpu().numpy() if torch.is_tensor(o) else np.asarray(o)
        obs[b] = np.asarray(o, np.float32).reshape(-1)
        if len(acts) != K or len(targets) != K:
            raise ValueError(f"sample {b}: {len(acts)} actions / {len(targets)} targets, expected {K}")
        for k in range(K):
            actions[k, b] = np.asarray(acts[k], np.float32).reshape(-1)
            tv[k, b] = targets[k][0]
            tr[k, b] = targets[k][1]
            tp[k, b] = np.asarray(targets[k][2], np.float32).reshape(-1)
    ratio = np.asarray(ratio, np.float32).reshape(-1)
    return obs, actions, tv, tr, tp, (None if ratio.size == 1 else ratio), game_pos


def allreduce_gradient(bucket, group=None):
    """The only collective of the training path: ONE all_reduce(SUM) over the data-parallel ranks, in
    place, of the flat bucket = the gradient followed by the 3K loss scalars (Trainer.bucket: both are
    views into it).  No-op for one rank / no process group.  Returns the world size."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)


def make_bucket(n_weights, unroll_steps, device=None):
    """(bucket, grad view, loss view): the gradient (n_weights floats, padded to a 256-byte boundary)
    and behind it the 3K loss scalars, in one allocation so that a step is one collective."""
    n, k3 = int(n_weights), 3 * int(unroll_steps)
    pad = -(-n // 64) * 64
    bucket = torch.zeros(pad + k3, dtype=torch.float32, device=device)
    return bucket, bucket[:n], bucket[pad:pad + k3]


# loss_type (init_criterion_and_optimizer, muzero_model.py:350-403) -> the criteria of the value and reward
# terms (MZS_LOSS_CE / KL / MSE / ZERO of include/mzsearch.h); the policy term is always the cross-entropy
LOSS_TYPES = {"general": (0, 0), "general_kkc": (1, 1), "game": (2, 3), "game_mmc": (2, 2)}


class Trainer:
    """Owns the training state of one engine handle (one GPU)."""

    def __init__(self, engine, unroll_steps, max_batch, optimizer="adam", learning_rate=1e-3,
                 lr_scheduler=None, num_of_epoch=300, loss_type="general"):
        if loss_type not in LOSS_TYPES:
            raise ValueError(f"loss_type={loss_type!r}: one of {sorted(LOSS_TYPES)} (muzero_model.py:350-403)")
        if optimizer not in OPTIMIZERS:
            raise ValueError(f"optimizer={optimizer!r}: 'adam' or 'sgd' (muzero_model.py:418-421)")
        if lr_scheduler not in SCHEDULERS:
            scheduled_lr(lr_scheduler, learning_rate, num_of_epoch, 1)   # raises with the reason
        self.eng = engine
        self.lib = engine.lib
        self.K, self.max_batch = int(unroll_steps), int(max_batch)
        self.opt, self.lr, self.sch, self.epoch = optimizer, float(learning_rate), lr_scheduler, num_of_epoch
        self.count = 0
        _lib.check(self.lib.mzs_train_create(engine.h, self.max_batch, self.K), "mzs_train_create")
        self.loss_type = loss_type
        _lib.check(self.lib.mzs_train_set_loss(engine.h, *LOSS_TYPES[loss_type]), "mzs_train_set_loss")
        dev, n = engine.device, engine.weight_count()
        self.bucket, self.grad, self.loss = make_bucket(n, self.K, dev)
        self.priority = torch.zeros((self.K, self.max_batch), dtype=torch.float32, device=dev)
        self._sync_pending = False
        self.sync_replicas()

    def sync_replicas(self):
        """One broadcast of rank 0's weights (no-op for one rank): the all-reduced gradient step
        keeps replicas identical only from an identical start.  The decision is the same on every
        rank: if no rank's engine has weights yet the broadcast is deferred to the first step() (which
        raises if there are still none); ranks that disagree raise here, on all of them."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        from . import shard
        dev = self.eng.device if dist.get_backend() == "nccl" else None
        hw = 1.0 if self.eng.has_weights else 0.0
        flag = torch.tensor([hw, -hw], dtype=torch.float32, device=dev or "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        any_has, all_have = float(flag[0]) > 0.5, float(flag[1]) < -0.5
        if any_has != all_have:
            raise RuntimeError("data-parallel Trainer: some ranks' engines have weights and some do not "
                               "(call engine.set_weights on every rank, or on none, before building the Trainer)")
        self._sync_pending = not all_have
        if all_have:
            self.eng.set_weights(shard.broadcast_blob(self.weights(), dev))

    def dropout_count(self, batch):
        """vision_model: mask bytes one step on `batch` samples consumes (mzs_train_dropout_count)"""
        n = C.c_int64(0)
        _lib.check(self.lib.mzs_train_dropout_count(self.eng.h, int(batch), C.byref(n)), "mzs_train_dropout_count")
        return int(n.value)

    def set_dropout(self, masks=None, seed=0):
        """vision_model: inject the Dropout2d masks of the NEXT step (uint8, one per (call, sample, channel) in the
        reference's forward order — what tests/golden/make_vision_train_fixtures.py records), or with
        masks=None restart the device generator at `seed`."""
        if masks is None:
            self._masks = None
            _lib.check(self.lib.mzs_train_set_dropout(self.eng.h, None, 0, C.c_uint64(int(seed))), "mzs_train_set_dropout")
            return
        self._masks = self.eng._dev(np.ascontiguousarray(masks, np.uint8).ravel(), torch.uint8)
        _lib.check(self.lib.mzs_train_set_dropout(self.eng.h, C.c_void_p(self._masks.data_ptr()),
                                                  self._masks.numel(), C.c_uint64(0)), "mzs_train_set_dropout")

    def current_lr(self):
        return scheduled_lr(self.sch, self.lr, self.epoch, self.count)

    def grad_step(self, obs, actions, tv, tr, tp, is_ratio=None, global_batch=None):
        """Enqueue forward + losses + backward for this rank's shard; fills self.grad / self.loss /
        self.priority[:, :B] (device tensors, no host sync)."""
        e = self.eng
        obs = e._dev(obs, torch.float32)
        B = obs.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        actions, tv, tr, tp = (e._dev(a, torch.float32) for a in (actions, tv, tr, tp))
        assert actions.shape == (self.K, B, e.A) and tv.shape == (self.K, B) and tp.shape == (self.K, B, e.A)
        isr = e._dev(is_ratio, torch.float32)
        self._keep = (obs, actions, tv, tr, tp, isr)
        prio = self.priority if B == self.max_batch else torch.zeros((self.K, B), dtype=torch.float32,
                                                                     device=e.device)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        _lib.check(self.lib.mzs_train_grad(e.h, B, p(obs), p(actions), p(tv), p(tr), p(tp), p(isr),
                                           C.c_float(1.0 / float(global_batch or B)), p(self.grad),
                                           p(self.loss), p(prio), e._stream()), "mzs_train_grad")
        return prio

    def apply(self):
        e = self.eng
        if self.sch == "onecyclelr":
            _lib.check(self.lib.mzs_train_set_momentum(e.h, scheduled_momentum(self.sch, self.epoch, self.count)),
                       "mzs_train_set_momentum")
        _lib.check(self.lib.mzs_train_apply(e.h, C.c_void_p(self.grad.data_ptr()), OPTIMIZERS[self.opt],
                                            C.c_double(self.current_lr()), e._stream()), "mzs_train_apply")
        self.count += 1
        if self.sch == "onecyclelr" and self.count - 1 > int(self.epoch):
            # torch's OneCycleLR: optimizer.step() has run, then scheduler.step(epoch=count) raises
            # (muzero_model.py:764-786) — same state, same error, same moment
            raise ValueError(f"Tried to step {self.count - 1} times. The specified number of total steps is "
                             f"{int(self.epoch)}")

    def step(self, obs, actions, tv, tr, tp, is_ratio=None, group=None):
        """One data-parallel training step.  With torch.distributed initialised every rank passes its
        own shard (equal sizes); gradients and loss terms are summed over ranks."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        B = np.shape(obs)[0] if not torch.is_tensor(obs) else obs.shape[0]
        if self._sync_pending:                       # weights arrived after construction: broadcast them now
            if not self.eng.has_weights:
                raise RuntimeError("Trainer.step: the engine has no weights (engine.set_weights)")
            self.sync_replicas()
        prio = self.grad_step(obs, actions, tv, tr, tp, is_ratio, global_batch=B * world)
        allreduce_gradient(self.bucket, group)                               # one flat bucket, one collective
        self.apply()
        return prio

    def weights(self):
        n = self.eng.weight_count()
        out = np.empty(n, np.float32)
        _lib.check(self.lib.mzs_get_weights(self.eng.h, out.ctypes.data_as(C.c_void_p), n), "mzs_get_weights")
        return out
