"""CPU restatement of the reference's training step on its 'vision_model' — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
path (muzero_amd/, the HIP kernels of muzero_amd/csrc/mzs_vision_train.hip) never does.

What it restates (file:line in /root/reference), in numpy float64 with a reverse-mode tape of its own:
  muzero_model.py:609-647   compute_forward: representation, prediction, K x (dynamics, prediction), the
                            x 0.5 gradient hook on every next hidden state (:640)
  muzero_model.py:650-734   evaluate_loss: value / policy / reward criteria on softmax + 1e-9, support
                            targets (:536-563), 1/K gradient hooks (:601, :657), new_priority (:710), the l2
                            term over zip()-truncated parameter lists (neural_network_mlp_model.py:280-285:
                            the FIRST 15 parameters of each of the three networks), importance weights, mean
  muzero_model.py:505-516   the action plane of the RGB model: (argmax(one-hot) + 1) / A, constant over 7 x 7
  neural_network_vision_model.py:41-82    Residual_block (ResNet v2): ONE BatchNorm2d, ONE Dropout2d and two
                            3x3 convolutions (`convolution_1` used twice) behind three applications
  neural_network_vision_model.py:84-122   Down_sample;  :125-163 representation;  :167-230 dynamics;
                            :234-300 prediction
  torch.nn.BatchNorm2d in training mode: batch statistics (biased variance, eps 1e-5), running statistics
                            updated with momentum 0.1 and the UNBIASED variance, once per application in
                            call order — the K-th dynamics / prediction calls included, whose outputs the
                            zip() of evaluate_loss drops (:655) but whose forward passes have run
  torch.nn.Dropout2d(p=0.5): whole (sample, channel) maps zeroed, survivors x 2.  The masks are an INPUT
                            (one byte per (sample, channel) per call, in call order): torch's generator stream
                            is not reproducible outside torch, so the fixtures record the masks the reference
                            drew (tests/golden/make_vision_train_fixtures.py).
  torch.optim.Adam / SGD(momentum 0.9) (muzero_model.py:418-421), on float32 copies as torch steps them.

Pinned by tests/golden/train_vis*.npz (the reference's own Muzero.train, two steps each):
tests/test_vision_train_oracle.py.
"""
import numpy as np

C, SIDE, HID = 3, 98, 147
EPS = 1e-5


# ------------------------------------------------------------------------------------------- layout
def blob_layout(A, Sd, H, L):
    """(name, shape) of every tensor of the blob, in order (include/mzs_layout.h mzs_vision_layout)."""
    def bn(p, c):
        return [(p + "_w", (c,)), (p + "_b", (c,)), (p + "_mean", (c,)), (p + "_var", (c,))]

    def res(p, c):
        return [(p + "_conv_a", (c, c, 3, 3)), (p + "_conv_b", (c, c, 3, 3))] + bn(p + "_bn", c)

    def mlp(p, out):
        mid = [(p + "_mid_w", (H, H)), (p + "_mid_b", (H,))] if L > 0 else []
        return [(p + "_in_w", (H, HID)), (p + "_in_b", (H,))] + mid + [(p + "_out_w", (out, H)), (p + "_out_b", (out,))]

    return ([("r_conv_in", (C // 2, C, 3, 3))] + res("r_res_in", C // 2) + [("r_conv_out", (C, C // 2, 3, 3))] +
            res("r_res_out", C) + res("r_res", C) + [("r_conv", (C, C, 3, 3))] + bn("r_bn", C) +
            [("d_conv", (C, C + 1, 3, 3))] + bn("d_bn", C) + res("d_res", C) +
            [("d_conv_rew", (C, C + 1, 1, 1)), ("d_conv_rew_b", (C,))] + mlp("d_fc", Sd) +
            res("p_res", C) + [("p_conv_val", (C, C, 1, 1)), ("p_conv_val_b", (C,)),
                               ("p_conv_pol", (C, C, 1, 1)), ("p_conv_pol_b", (C,))] +
            mlp("p_v", Sd) + mlp("p_p", A))


def l2_names(L):
    """The tensors inside the reference's l2 term: zip() over the three networks' parameters() stops at
    the shortest list — dynamics, 15 parameters with L >= 1 — so the first 15 of each count
    (registration order: a Residual_block lists its BatchNorm2d weight, bias, then convolution_1, convolution_3)."""
    assert L >= 1
    res = lambda p: [p + "_bn_w", p + "_bn_b", p + "_conv_a", p + "_conv_b"]  # noqa: E731
    mlp = lambda p: [p + "_in_w", p + "_in_b", p + "_mid_w", p + "_mid_b", p + "_out_w", p + "_out_b"]  # noqa: E731
    r = ["r_conv_in"] + res("r_res_in") + ["r_conv_out"] + res("r_res_out") + res("r_res") + ["r_conv", "r_bn_w", "r_bn_b"]
    d = ["d_conv", "d_bn_w", "d_bn_b"] + res("d_res") + ["d_conv_rew", "d_conv_rew_b"] + mlp("d_fc")
    p = res("p_res") + ["p_conv_val", "p_conv_val_b"] + mlp("p_v") + ["p_conv_pol", "p_conv_pol_b"] + mlp("p_p")
    n = min(len(r), len(d), len(p))
    return r[:n] + d[:n] + p[:n]


def unpack(blob, A, Sd, H, L):
    out, p = {}, 0
    for name, shape in blob_layout(A, Sd, H, L):
        n = int(np.prod(shape))
        out[name] = np.asarray(blob[p:p + n], np.float64).reshape(shape).copy()
        p += n
    assert p == len(blob), (p, len(blob))
    return out


def pack(tensors, A, Sd, H, L, dtype=np.float32):
    return np.concatenate([np.asarray(tensors[n], dtype).ravel() for n, _ in blob_layout(A, Sd, H, L)])


# ------------------------------------------------------------------------------------------- tape
class V:
    __slots__ = ("v", "g")

    def __init__(self, v):
        self.v, self.g = v, None

    def add_grad(self, g):
        self.g = g if self.g is None else self.g + g


class Tape:
    def __init__(self):
        self.ops = []

    def backward(self):
        for fn in reversed(self.ops):
            fn()


def _pad(x):
    return np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))


def conv3(tp, x, w, stride):
    """3x3, padding 1, no bias (torch.nn.Conv2d cross-correlation)"""
    xp = _pad(x.v)
    S = x.v.shape[2]
    So = (S - 1) // stride + 1
    span = stride * (So - 1) + 1
    y = np.zeros((x.v.shape[0], w.v.shape[0], So, So))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("oc,bchw->bohw", w.v[:, :, ky, kx], xp[:, :, ky:ky + span:stride, kx:kx + span:stride])
    out = V(y)

    def bwd():
        if out.g is None:
            return
        dxp = np.zeros_like(xp)
        dw = np.zeros_like(w.v)
        for ky in range(3):
            for kx in range(3):
                sl = (slice(None), slice(None), slice(ky, ky + span, stride), slice(kx, kx + span, stride))
                dxp[sl] += np.einsum("oc,bohw->bchw", w.v[:, :, ky, kx], out.g)
                dw[:, :, ky, kx] = np.einsum("bohw,bchw->oc", out.g, xp[sl])
        x.add_grad(dxp[:, :, 1:-1, 1:-1])
        w.add_grad(dw)
    tp.ops.append(bwd)
    return out


def conv1(tp, x, w, b):
    y = np.einsum("oc,bchw->bohw", w.v[:, :, 0, 0], x.v) + b.v[None, :, None, None]
    out = V(y)

    def bwd():
        if out.g is None:
            return
        x.add_grad(np.einsum("oc,bohw->bchw", w.v[:, :, 0, 0], out.g))
        w.add_grad(np.einsum("bohw,bchw->oc", out.g, x.v)[:, :, None, None])
        b.add_grad(out.g.sum((0, 2, 3)))
    tp.ops.append(bwd)
    return out


def bn_train(tp, x, bn, stats):
    """BatchNorm2d, training mode; bn = (gamma V, beta V, running_mean array, running_var array) — the
    running statistics are updated in place; `stats` collects (mean, biased var) of every application"""
    gamma, beta, rm, rv = bn
    n = x.v.shape[0] * x.v.shape[2] * x.v.shape[3]
    mean = x.v.mean((0, 2, 3))
    var = x.v.var((0, 2, 3))
    rstd = 1.0 / np.sqrt(var + EPS)
    xh = (x.v - mean[None, :, None, None]) * rstd[None, :, None, None]
    out = V(gamma.v[None, :, None, None] * xh + beta.v[None, :, None, None])
    rm[:] = 0.9 * rm + 0.1 * mean
    rv[:] = 0.9 * rv + 0.1 * var * (n / (n - 1.0))
    stats.append((mean, var))

    def bwd():
        if out.g is None:
            return
        dy = out.g
        dg = (dy * xh).sum((0, 2, 3))
        db = dy.sum((0, 2, 3))
        gamma.add_grad(dg)
        beta.add_grad(db)
        x.add_grad((gamma.v * rstd / n)[None, :, None, None] *
                   (n * dy - db[None, :, None, None] - xh * dg[None, :, None, None]))
    tp.ops.append(bwd)
    return out


def relu(tp, x, scale=None):
    """relu, optionally times a per-(sample, channel) factor (the dropout mask x 2)"""
    pos = x.v > 0
    f = pos if scale is None else pos * scale[:, :, None, None]
    out = V(x.v * f)

    def bwd():
        if out.g is not None:
            x.add_grad(out.g * f)
    tp.ops.append(bwd)
    return out


def add(tp, a, b):
    out = V(a.v + b.v)

    def bwd():
        if out.g is not None:
            a.add_grad(out.g)
            b.add_grad(out.g)
    tp.ops.append(bwd)
    return out


def avgpool(tp, x):
    """AvgPool2d(3, stride 2, padding 1), count_include_pad: always / 9"""
    xp = _pad(x.v)
    S = x.v.shape[2]
    So = (S - 1) // 2 + 1
    span = 2 * (So - 1) + 1
    y = np.zeros(x.v.shape[:2] + (So, So))
    for ky in range(3):
        for kx in range(3):
            y += xp[:, :, ky:ky + span:2, kx:kx + span:2]
    out = V(y / 9.0)

    def bwd():
        if out.g is None:
            return
        dxp = np.zeros_like(xp)
        for ky in range(3):
            for kx in range(3):
                dxp[:, :, ky:ky + span:2, kx:kx + span:2] += out.g / 9.0
        x.add_grad(dxp[:, :, 1:-1, 1:-1])
    tp.ops.append(bwd)
    return out


def linear(tp, x, w, b, act):
    z = x.v @ w.v.T + b.v
    pos = z > 0
    out = V(z * pos if act else z)

    def bwd():
        if out.g is None:
            return
        dz = out.g * pos if act else out.g
        x.add_grad(dz @ w.v)
        w.add_grad(dz.T @ x.v)
        b.add_grad(dz.sum(0))
    tp.ops.append(bwd)
    return out


def reshape(tp, x, shape):
    out = V(x.v.reshape(shape))

    def bwd():
        if out.g is not None:
            x.add_grad(out.g.reshape(x.v.shape))
    tp.ops.append(bwd)
    return out


def cat_plane(tp, h, plane):
    """torch.cat([hidden, action plane], dim=1); the plane is data"""
    out = V(np.concatenate([h.v, plane], axis=1))

    def bwd():
        if out.g is not None:
            h.add_grad(out.g[:, :C])
    tp.ops.append(bwd)
    return out


def grad_scale(tp, x, s):
    out = V(x.v)

    def bwd():
        if out.g is not None:
            x.add_grad(out.g * s)
    tp.ops.append(bwd)
    return out


# ------------------------------------------------------------------------------------------- the networks
class Net:
    def __init__(self, blob, A, Sd, H, L):
        self.A, self.Sd, self.H, self.L = A, Sd, H, L
        self.t = {k: V(v) for k, v in unpack(blob, A, Sd, H, L).items()}
        self.masks = None
        self.mask_pos = 0
        self.stats = []

    def bn(self, p):
        t = self.t
        return (t[p + "_w"], t[p + "_b"], t[p + "_mean"].v, t[p + "_var"].v)

    def mask(self, B, c):
        n = B * c
        m = self.masks[self.mask_pos:self.mask_pos + n]
        assert m.size == n, "dropout masks exhausted"
        self.mask_pos += n
        return m.reshape(B, c).astype(np.float64) * 2.0

    def R(self, tp, x, p):
        """dropout(relu(batchnorm(x))) of a residual block"""
        B, c = x.v.shape[:2]
        return relu(tp, bn_train(tp, x, self.bn(p + "_bn"), self.stats), self.mask(B, c))

    def res(self, tp, x, p):
        t = self.t
        y = conv3(tp, self.R(tp, x, p), t[p + "_conv_a"], 1)
        y = conv3(tp, self.R(tp, y, p), t[p + "_conv_b"], 1)
        y = conv3(tp, self.R(tp, y, p), t[p + "_conv_a"], 1)
        return add(tp, y, x)

    def mlp(self, tp, x, p):
        t = self.t
        y = linear(tp, x, t[p + "_in_w"], t[p + "_in_b"], True)
        for _ in range(self.L):
            y = linear(tp, y, t[p + "_mid_w"], t[p + "_mid_b"], True)
        return linear(tp, y, t[p + "_out_w"], t[p + "_out_b"], False)

    def representation(self, tp, obs):
        t = self.t
        x = conv3(tp, obs, t["r_conv_in"], 2)
        x = self.res(tp, self.res(tp, x, "r_res_in"), "r_res_in")
        x = conv3(tp, x, t["r_conv_out"], 2)
        x = self.res(tp, self.res(tp, x, "r_res_out"), "r_res_out")
        x = avgpool(tp, x)
        for _ in range(3):
            x = self.res(tp, x, "r_res_out")
        x = avgpool(tp, x)
        return self.res(tp, x, "r_res")

    def prediction(self, tp, h):
        t = self.t
        x = h
        for _ in range(self.L):
            x = self.res(tp, x, "p_res")
        B = x.v.shape[0]
        v = self.mlp(tp, reshape(tp, conv1(tp, x, t["p_conv_val"], t["p_conv_val_b"]), (B, HID)), "p_v")
        p = self.mlp(tp, reshape(tp, conv1(tp, x, t["p_conv_pol"], t["p_conv_pol_b"]), (B, HID)), "p_p")
        return p, v

    def dynamics(self, tp, h, action_rows):
        t = self.t
        B = h.v.shape[0]
        idx = np.argmax(action_rows, axis=1)                      # zero rows (absorbing) -> action 0
        plane = np.broadcast_to(((idx + 1.0) / self.A)[:, None, None, None], (B, 1, 7, 7))
        x = cat_plane(tp, h, plane)
        y = relu(tp, bn_train(tp, conv3(tp, x, t["d_conv"], 1), self.bn("d_bn"), self.stats))
        for _ in range(self.L):
            y = self.res(tp, y, "d_res")
        y = relu(tp, y)
        r = self.mlp(tp, reshape(tp, conv1(tp, x, t["d_conv_rew"], t["d_conv_rew_b"]), (B, HID)), "d_fc")
        return r, y


# ------------------------------------------------------------------------------------------- losses
LOSS_CE, LOSS_KL, LOSS_MSE, LOSS_ZERO = 0, 1, 2, 3
LOSS_CE_CLAMPED = 100


def support_target(x, S):
    """transform_with_support (muzero_model.py:536-563) in float32 as the reference evaluates it:
    rows with ONE non-zero bin (index, weight)"""
    x = np.asarray(x, np.float32)
    sd, rem = S // 2, S & 1
    t = np.sign(x) * (np.sqrt(np.abs(x) + np.float32(1)) - np.float32(1)) + np.float32(0.001) * x
    t = np.clip(t, np.float32(-sd), np.float32(sd + rem)).astype(np.float32)
    sg = np.sign(t)
    idx = np.clip(np.floor(t) + sd + 1, -(S - 1), S - 1).astype(np.int64)
    w = ((np.ceil(t) - t) * sg).astype(np.float32)
    out = np.zeros((x.size, S))
    out[np.arange(x.size), idx] = w          # idx >= 1: floor(t) >= -sd
    return out


def loss_term(tp, z, target, mode, upstream):
    """one criterion of neural_network_mlp_model.py:235-262 on q = softmax(z) + 1e-9; per-sample losses;
    `upstream` [B] = d loss_nn / d loss[b] (importance weight / B x the 1/K hook)"""
    zz = z.v - z.v.max(1, keepdims=True)
    e = np.exp(zz)
    s = e / e.sum(1, keepdims=True)
    q = s + 1e-9
    n = z.v.shape[1]
    if mode in (LOSS_CE, LOSS_CE_CLAMPED):
        p = target + 1e-9 if mode == LOSS_CE_CLAMPED else target
        loss, c = (-p * np.log(q)).sum(1), -p / q
    elif mode == LOSS_KL:
        p = target + 1e-9
        with np.errstate(invalid="ignore"):
            loss, c = (p * (np.log(p) - np.log(q))).sum(1), -p / q
    elif mode == LOSS_MSE:
        loss, c = ((target - q) ** 2).mean(1), -2.0 * (target - q) / n
    else:
        loss, c = np.zeros(z.v.shape[0]), np.zeros_like(q)

    def bwd():
        z.add_grad(upstream[:, None] * s * (c - (c * s).sum(1, keepdims=True)))
    tp.ops.append(bwd)
    return loss, s


def inverse_value(s, S):
    """inverse_transform_with_support (muzero_model.py:571-587); the closed form is evaluated in float32
    operation by operation as torch does (its sqrt(1 + 0.004 (|y| + 1.001)) - 1 cancels seven bits: in
    float64 the result would differ from the reference's by ~5e-4)"""
    sd, rem = S // 2, S & 1
    f = np.float32
    y = (s * np.arange(-sd, sd + rem)[None, :]).sum(1).astype(f)
    inner = (np.sqrt(f(1) + f(4 * 0.001) * (np.abs(y) + f(1) + f(0.001))) - f(1)) / f(2 * 0.001)
    return (np.sign(y) * (inner ** 2 - f(1))).astype(np.float64)


# ------------------------------------------------------------------------------------------- the step
def train_grad(blob, dims, K, obs, actions, tv, tr, tpol, is_ratio, masks, loss_value=LOSS_CE, loss_reward=LOSS_CE,
               inv_global_batch=None):
    """-> dict(grad, losses [3K] (3K-1 raw terms then loss_nn), priority [K][B], running (blob with the running
    statistics after the forward passes), bn_stats).  Arguments as mzs_train_grad (include/mzsearch.h)."""
    _, A, Sd, H, L = [int(v) for v in dims]
    B = obs.shape[0]
    net = Net(blob, A, Sd, H, L)
    net.masks = np.asarray(masks).ravel()
    tp = Tape()
    wb = (np.ones(B) if is_ratio is None else np.asarray(is_ratio, np.float64)) * (inv_global_batch or 1.0 / B)
    pol_mode = LOSS_CE_CLAMPED if loss_value == LOSS_KL else LOSS_CE

    h = net.representation(tp, V(np.asarray(obs, np.float64)))
    preds = [net.prediction(tp, h) + (None,)]
    for k in range(K):
        r, nh = net.dynamics(tp, h, actions[k])
        # register_hook(grad * 0.5) on next_state_normalized (:640): a tensor hook scales the TOTAL gradient
        # arriving at the tensor, from the prediction that has already consumed it and from the next dynamics
        h = grad_scale(tp, nh, 0.5)
        p, v = net.prediction(tp, h)
        preds.append((p, v, r))
    assert net.mask_pos == net.masks.size, (net.mask_pos, net.masks.size)

    n_terms = 3 * K - 1
    losses = np.zeros((B, n_terms))
    priority = np.zeros((K, B))
    for k in range(K):
        p, v, r = preds[k][:3]
        gs = wb * (1.0 / K if k > 0 else 1.0)
        t0 = 0 if k == 0 else 3 * k - 1
        lv, sv = loss_term(tp, v, support_target(tv[k], Sd), loss_value, gs)
        priority[k] = np.abs(inverse_value(sv, Sd) - np.asarray(tv[k], np.float64))
        lp, _ = loss_term(tp, p, np.asarray(tpol[k], np.float64), pol_mode, gs)
        losses[:, t0], losses[:, t0 + 1] = lv, lp
        if k > 0:
            lr_, _ = loss_term(tp, r, support_target(tr[k], Sd), loss_reward, gs)
            losses[:, t0 + 2] = lr_
    tp.backward()

    names = l2_names(L)
    sq = sum(float((net.t[n].v ** 2).sum()) for n in names)
    grads = {}
    for name, shape in blob_layout(A, Sd, H, L):
        g = net.t[name].g
        g = np.zeros(shape) if g is None or name.endswith(("_mean", "_var")) else g.reshape(shape)
        if name in names:
            g = g + 2e-4 * wb.sum() * net.t[name].v
        grads[name] = g
    out_losses = np.concatenate([losses.sum(0) * (inv_global_batch or 1.0 / B),
                                 [(wb * losses.sum(1)).sum() + wb.sum() * 1e-4 * sq]])
    return dict(grad=pack(grads, A, Sd, H, L, np.float64), losses=out_losses, priority=priority,
                running=pack({k: v.v for k, v in net.t.items()}, A, Sd, H, L, np.float64), bn_stats=net.stats)


class Optimizer:
    """torch.optim.Adam(lr, betas (0.9, 0.999), eps 1e-8) / SGD(lr, momentum 0.9) on float32 state, as
    muzero_model.py:418-421 builds them; running statistics are not parameters (taken from `running`)."""

    def __init__(self, kind, n, dims):
        self.kind, self.step_n = kind, 0
        self.m = np.zeros(n, np.float32)
        self.v = np.zeros(n, np.float32)
        _, A, Sd, H, L = [int(v) for v in dims]
        self.is_stat = np.concatenate([np.full(int(np.prod(s)), nm.endswith(("_mean", "_var")))
                                       for nm, s in blob_layout(A, Sd, H, L)])

    def apply(self, blob, grad, running, lr, momentum=0.9):
        self.step_n += 1
        w = np.asarray(blob, np.float32).copy()
        g = np.asarray(grad, np.float32)
        lr = float(lr)
        if self.kind == "adam":
            self.m = (self.m + (g - self.m) * np.float32(1.0 - momentum)).astype(np.float32)
            self.v = (self.v * np.float32(0.999) + np.float32(0.001) * g * g).astype(np.float32)
            bc1, bc2 = 1.0 - momentum ** self.step_n, 1.0 - 0.999 ** self.step_n
            denom = np.sqrt(self.v) * np.float32(1.0 / np.sqrt(bc2)) + np.float32(1e-8)
            w = (w - np.float32(lr / bc1) * (self.m / denom)).astype(np.float32)
        else:
            self.m = g.copy() if self.step_n == 1 else (np.float32(momentum) * self.m + g).astype(np.float32)
            w = (w - np.float32(lr) * self.m).astype(np.float32)
        w[self.is_stat] = np.asarray(running, np.float32)[self.is_stat]
        return w
