#!/usr/bin/env python3
"""Golden-vector generator (container-only; never runs on the GPU box).

Imports the reference implementation *by path* from /root/reference (no
reference source is copied into this repo), runs its own
``Monte_carlo_tree_search.run`` / ``Muzero.*_inference`` on seeded inputs and
stores inputs + outputs as small ``.npz`` fixtures next to this script.

Reference entry points exercised (file:line in /root/reference):
  * monte_carlo_tree_search.py:224  Monte_carlo_tree_search.run
  * muzero_model.py:828/:848/:883   representation/dynamics/prediction inference
  * neural_network_mlp_model.py:7-143  the three MLP nets + scale_to_bound_action
  * muzero_model.py:571             inverse_transform_with_support

Checkpoints (model_checkpoint/<tag>_*.pt, GPL-3.0 data of DHDev0/Muzero) are
read with ``torch.load(weights_only=True)`` under an explicit class allow-list;
nothing is unpickled outside that list.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py
"""
import collections
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


# --------------------------------------------------------------------------
# gymnasium is not installed: the reference only reads .n / .shape / .dtype of
# the space objects (muzero_model.py:478-486, :1002, :1006-1023).
# --------------------------------------------------------------------------
def _install_gym_stub():
    gym = types.ModuleType("gymnasium")
    sp = types.ModuleType("gymnasium.spaces")
    bx = types.ModuleType("gymnasium.spaces.box")
    tp = types.ModuleType("gymnasium.spaces.tuple")

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, shape):
            self.shape, self.dtype = shape, np.dtype("float32")

    class Tuple(tuple):
        pass

    sp.Discrete, bx.Box, tp.Tuple = Discrete, Box, Tuple
    sp.box, sp.tuple, gym.spaces = bx, tp, sp
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": sp,
                        "gymnasium.spaces.box": bx, "gymnasium.spaces.tuple": tp})
    return Discrete, Box


Discrete, Box = _install_gym_stub()
sys.path.insert(0, REF)
import muzero_model as mm                      # noqa: E402
import neural_network_mlp_model as mlp         # noqa: E402
from monte_carlo_tree_search import Monte_carlo_tree_search  # noqa: E402

ALLOW = [mlp.Representation_function, mlp.Dynamics_function, mlp.Prediction_function,
         nn.Sequential, nn.Linear, nn.ELU, nn.Tanh, collections.OrderedDict]


def load_checkpoint(tag):
    mz = mm.Muzero(load=True, type_format=torch.float32)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with torch.serialization.safe_globals(ALLOW):
            mz.load_model(tag=tag, observation_space_dimensions=None, device="cpu")
    finally:
        os.chdir(cwd)
    return mz


def fresh_model(obs, A, Sd, H, L, seed, std):
    mz = mm.Muzero(model_structure="mlp_model", observation_space_dimensions=Box((obs,)),
                   action_space_dimensions=Discrete(A), state_space_dimensions=Sd,
                   hidden_layer_dimensions=H, number_of_hidden_layer=L,
                   k_hypothetical_steps=5, device="cpu")
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
            for p in net.parameters():          # shared tensors are visited once
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return mz


def pack_weights(mz):
    """Flatten the three nets into the blob order of include/mzsearch.h.

    Every tensor is row-major [out][in] exactly as torch stores nn.Linear.weight.
    The tied mid layer (neural_network_mlp_model.py:33-39, :75-84, :116-122: the
    same linear_mid object repeated L times) is stored once per net.
    """
    L = mz.number_of_hidden_layer
    last = 2 * L + 2
    rs = mz.representation_function.state_dict()
    ds = mz.dynamics_function.state_dict()
    ps = mz.prediction_function.state_dict()
    parts = [rs["state_norm.0.weight"], rs["state_norm.0.bias"]]
    if L > 0:
        parts += [rs["state_norm.2.weight"], rs["state_norm.2.bias"]]
    parts += [rs[f"state_norm.{last}.weight"], rs[f"state_norm.{last}.bias"]]
    parts += [ds["reward.0.weight"], ds["reward.0.bias"]]
    if L > 0:
        parts += [ds["reward.2.weight"], ds["reward.2.bias"]]
    parts += [ds[f"reward.{last}.weight"], ds[f"reward.{last}.bias"],
              ds[f"next_state_normalized.{last}.weight"], ds[f"next_state_normalized.{last}.bias"]]
    parts += [ps["policy.0.weight"], ps["policy.0.bias"]]
    if L > 0:
        parts += [ps["policy.2.weight"], ps["policy.2.bias"]]
    parts += [ps[f"policy.{last}.weight"], ps[f"policy.{last}.bias"],
              ps[f"value.{last}.weight"], ps[f"value.{last}.bias"]]
    # sanity: trunks are shared between the two heads of each net
    assert torch.equal(ds["reward.0.weight"], ds["next_state_normalized.0.weight"])
    assert torch.equal(ps["policy.0.weight"], ps["value.0.weight"])
    return np.concatenate([p.detach().numpy().astype(np.float32).ravel() for p in parts])


def model_dims(mz):
    return dict(obs_dim=int(mz.observation_dimension), A=int(mz.action_dimension),
                S_dim=int(mz.state_dimension), H=int(mz.hidden_layer_dimension),
                L=int(mz.number_of_hidden_layer))


# --------------------------------------------------------------------------
# Recording proxies: they call the reference and log what it returned.
# --------------------------------------------------------------------------
class RecordingModel:
    def __init__(self, mz):
        self.mz = mz
        self.dyn, self.pred, self.repr = [], [], []

    def representation_function_inference(self, obs):
        h = self.mz.representation_function_inference(obs)
        self.repr.append(h.numpy().copy())
        return h

    def dynamics_function_inference(self, h, a):
        r, h2 = self.mz.dynamics_function_inference(h, a)
        self.dyn.append((int(a), np.float32(r), h2.numpy().copy()))
        return r, h2

    def prediction_function_inference(self, h):
        p, v = self.mz.prediction_function_inference(h)
        self.pred.append((p.copy(), np.float32(v)))
        return p, v


class RecordingMCTS(Monte_carlo_tree_search):
    """Reference search + bookkeeping of scores / paths (no behaviour change:
    every override calls the reference method exactly once)."""

    def begin(self):
        self.scores, self.margins, self.paths, self.n_ucb = [], [], [], 0

    def ucb_score(self, parent, child):
        s = super().ucb_score(parent, child)
        self.scores.append(float(s))
        self.n_ucb += 1
        return s

    def select_child(self):
        n0 = len(self.scores)
        pure_noise = self.node.visit_count == 0   # root at sim 0: score == tie-break draw
        out = super().select_child()
        sc = sorted(self.scores[n0:])
        if len(sc) >= 2 and not pure_noise:
            self.margins.append(sc[-1] - sc[-2])
        return out

    def choice_node_to_expand_using_max_ucb_score(self, history, search_path):
        out = super().choice_node_to_expand_using_max_ucb_score(history, search_path)
        self.paths.append(list(history))
        return out


def run_searches(mz, obs_batch, seeds, S, train, gamma, maxdepth_cap=None, dims=None, store_obs=True):
    """dims: override of model_dims(mz) for models whose hidden state is not a vector of S_dim floats
    (make_vision_fixtures.py: "hid_n" = floats of one hidden state); store_obs=False leaves the observations
    out of the record (the caller stores the seed they are regenerated from)."""
    dims = dict(model_dims(mz) if dims is None else dims)
    hid_n = int(dims.pop("hid_n", dims["S_dim"]))
    A = dims["A"]
    N = len(seeds)
    rec = dict(
        obs=np.asarray(obs_batch, np.float32), seed=np.asarray(seeds, np.int64),
        visits=np.zeros((N, A), np.float64), root_value=np.zeros(N, np.float32),
        root_visit=np.zeros(N, np.int32),
        root_prior=np.zeros((N, A), np.float64),        # after noise (f64)
        root_policy=np.zeros((N, A), np.float32),       # network prior, before noise
        root_pred_value=np.zeros(N, np.float32),
        root_hidden=np.zeros((N, hid_n), np.float32),
        dirichlet=np.zeros((N, A), np.float64), n_draws=np.zeros(N, np.int64),
        total_path_len=np.zeros(N, np.int64), min_margin=np.full(N, np.inf),
        max_depth=np.zeros(N, np.int32),
        sim_parent=np.zeros((N, S), np.int32), sim_action=np.zeros((N, S), np.int32),
        sim_reward=np.zeros((N, S), np.float32), sim_value=np.zeros((N, S), np.float32),
        sim_prior=np.zeros((N, S, A), np.float32),
        sim_hidden=np.zeros((N, S, hid_n), np.float32),
        child_visit=np.zeros((N, 1 + A * (S + 1)), np.int32),
        child_vsum=np.zeros((N, 1 + A * (S + 1)), np.float32),
        t0_action=np.zeros(N, np.int32),
    )
    draws = []
    mcts = RecordingMCTS(discount=gamma)
    for i in range(N):
        model = RecordingModel(mz)
        mcts.reset(discount=gamma)
        mcts.begin()
        np.random.seed(int(seeds[i]))
        obs = torch.tensor(rec["obs"][i:i + 1])
        policy, root, action = mcts.run(observation=obs, model=model,
                                        num_simulations=S, train=train)
        assert list(action) == list(range(A))
        # replay the global stream to capture exactly what was consumed
        rs = np.random.RandomState(int(seeds[i]))
        if train and S > 0:
            rec["dirichlet"][i] = rs.dirichlet([mcts.root_dirichlet_alpha] * A)
        u = rs.uniform(1e-7, 2e-7, size=mcts.n_ucb) if mcts.n_ucb else np.zeros(0)
        # the caller's stream position afterwards must match ours
        assert np.random.random_sample() == rs.random_sample()
        draws.append(u)
        rec["n_draws"][i] = mcts.n_ucb
        rec["visits"][i] = policy
        rec["root_value"][i] = np.float32(root.value())
        rec["root_visit"][i] = root.visit_count
        rec["root_prior"][i] = [np.float64(root.children[a].prior) for a in range(A)]
        rec["root_policy"][i] = model.pred[0][0][0]
        rec["root_pred_value"][i] = model.pred[0][1]
        rec["root_hidden"][i] = model.repr[0][0].ravel()
        rec["total_path_len"][i] = sum(len(p) for p in mcts.paths)
        rec["max_depth"][i] = max([len(p) for p in mcts.paths] or [0])
        if mcts.margins:
            rec["min_margin"][i] = min(mcts.margins)
        rec["t0_action"][i] = int(np.argmax(policy))
        # node numbering of the build: root = 0, children of expansion e at 1+e*A+a
        index = {(): 0}
        for a in range(A):
            index[(a,)] = 1 + a
        for s, path in enumerate(mcts.paths):
            leaf = tuple(path)
            rec["sim_parent"][i, s] = index[leaf[:-1]]
            rec["sim_action"][i, s] = leaf[-1]
            for a in range(A):
                index[leaf + (a,)] = 1 + (s + 1) * A + a
            a_, r_, h_ = model.dyn[s]
            assert a_ == leaf[-1]
            rec["sim_reward"][i, s] = r_
            rec["sim_hidden"][i, s] = h_[0].ravel()
            rec["sim_prior"][i, s] = model.pred[s + 1][0][0]
            rec["sim_value"][i, s] = model.pred[s + 1][1]
        # final per-node statistics
        stack = [((), root)]
        while stack:
            key, node = stack.pop()
            rec["child_visit"][i, index[key]] = node.visit_count
            rec["child_vsum"][i, index[key]] = np.float32(node.value_sum)
            for a, ch in node.children.items():
                stack.append((key + (a,), ch))
    off = np.zeros(N + 1, np.int64)
    off[1:] = np.cumsum([len(u) for u in draws])
    rec["draw_offsets"] = off
    rec["draws"] = np.concatenate(draws) if draws else np.zeros(0)
    rec.update({k: np.int64(v) for k, v in dims.items()})
    rec.update(S=np.int64(S), train=np.int64(train), gamma=np.float64(gamma),
               pb_c_base=np.float64(mcts.pb_c_base), pb_c_init=np.float64(mcts.pb_c_init),
               alpha=np.float64(mcts.root_dirichlet_alpha),
               frac=np.float64(mcts.root_exploration_fraction))
    # sqrt(N)*pb_c(N) exactly as numpy evaluates it inside ucb_score (mcts.py:156-157)
    tab = np.zeros(S + 2, np.float64)
    for n in range(S + 2):
        pb_c = np.log((n + mcts.pb_c_base + 1) / mcts.pb_c_base) + mcts.pb_c_init
        tab[n] = np.sqrt(n) * pb_c
    rec["pbc_table"] = tab
    if not store_obs:
        del rec["obs"]
    return rec


def cartpole_obs(n, seed, dim=4, scale=0.05):
    return np.random.RandomState(seed).uniform(-scale, scale, size=(n, dim)).astype(np.float32)


def model_vectors(mz, n, seed):
    """L2 fixture: reference wrapper outputs on random hidden states/actions."""
    d = model_dims(mz)
    rs = np.random.RandomState(seed)
    obs = rs.uniform(-1, 1, size=(n, d["obs_dim"])).astype(np.float32)
    obs[: n // 4] *= 0.05
    hid = rs.uniform(0, 1, size=(n, d["S_dim"])).astype(np.float32)
    act = rs.randint(0, d["A"], size=n).astype(np.int32)
    out = dict(obs=obs, hidden=hid, action=act,
               repr_hidden=np.zeros((n, d["S_dim"]), np.float32),
               dyn_reward=np.zeros(n, np.float32),
               dyn_hidden=np.zeros((n, d["S_dim"]), np.float32),
               pred_policy=np.zeros((n, d["A"]), np.float32),
               pred_value=np.zeros(n, np.float32),
               # chained: what recurrent_inference(hidden, action) must return
               rec_policy=np.zeros((n, d["A"]), np.float32),
               rec_value=np.zeros(n, np.float32),
               ini_policy=np.zeros((n, d["A"]), np.float32),
               ini_value=np.zeros(n, np.float32))
    for i in range(n):
        h0 = mz.representation_function_inference(obs[i:i + 1])
        out["repr_hidden"][i] = h0.numpy()[0]
        p, v = mz.prediction_function_inference(h0)
        out["ini_policy"][i], out["ini_value"][i] = p[0], v
        r, h2 = mz.dynamics_function_inference(torch.tensor(hid[i:i + 1]), int(act[i]))
        out["dyn_reward"][i], out["dyn_hidden"][i] = r, h2.numpy()[0]
        p, v = mz.prediction_function_inference(torch.tensor(hid[i:i + 1]))
        out["pred_policy"][i], out["pred_value"][i] = p[0], v
        p, v = mz.prediction_function_inference(h2)
        out["rec_policy"][i], out["rec_value"][i] = p[0], v
    # scale_to_bound_action edge (neural_network_mlp_model.py:139): constant row
    x = torch.tensor([[0.5] * d["S_dim"], list(np.linspace(0, 3e-6, d["S_dim"]))],
                     dtype=torch.float32)
    out["scale_in"] = x.numpy().copy()
    out["scale_out"] = mlp.scale_to_bound_action(x.clone()).numpy()
    # inverse_transform_with_support sweep (muzero_model.py:571-587)
    logits = torch.tensor(rs.normal(0, 3, size=(64, d["S_dim"])).astype(np.float32))
    logits[0] = 0.0
    logits[1, :] = -50.0
    logits[1, -1] = 50.0
    logits[2, :] = -50.0
    logits[2, 0] = 50.0
    out["inv_in"] = logits.numpy().copy()
    out["inv_out"] = mz.inverse_transform_with_support(logits).numpy()[:, 0]
    out.update({k: np.int64(v) for k, v in d.items()})
    return out


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.set_num_threads(1)
    models = {}
    for tag in (123, 127, 133, 137):
        models[f"tag{tag}"] = load_checkpoint(tag)
    # synthetic shapes the shipped checkpoints do not cover (LunarLander-like A=4)
    models["synA4"] = fresh_model(obs=8, A=4, Sd=31, H=64, L=1, seed=4, std=0.3)
    models["synA3"] = fresh_model(obs=6, A=3, Sd=21, H=32, L=2, seed=3, std=0.4)
    torch.set_default_dtype(torch.float32)

    for name, mz in models.items():
        blob = pack_weights(mz)
        save(f"weights_{name}.npz", blob=blob, **{k: np.int64(v) for k, v in model_dims(mz).items()})
        save(f"model_L2_{name}.npz", **model_vectors(mz, 128 if name == "tag123" else 48, seed=11))

    g = 0.997
    mz = models["tag123"]
    save("search_tag123_S50.npz", **run_searches(mz, cartpole_obs(256, 1000), np.arange(256), 50, True, g))
    save("search_tag123_S11.npz", **run_searches(mz, cartpole_obs(64, 1001), 5000 + np.arange(64), 11, True, g))
    save("search_tag123_S1.npz", **run_searches(mz, cartpole_obs(8, 1002), 6000 + np.arange(8), 1, True, g))
    save("search_tag123_S0.npz", **run_searches(mz, cartpole_obs(8, 1003), 7000 + np.arange(8), 0, True, g))
    save("search_tag123_S50_notrain.npz", **run_searches(mz, cartpole_obs(32, 1004), 8000 + np.arange(32), 50, False, g))
    # wider observation range: deeper / more uneven trees
    save("search_tag123_S50_wide.npz", **run_searches(mz, cartpole_obs(64, 1005, scale=1.0), 9000 + np.arange(64), 50, True, g))
    save("search_tag127_S50.npz", **run_searches(models["tag127"], cartpole_obs(48, 1006), 10000 + np.arange(48), 50, True, g))
    save("search_tag133_S50.npz", **run_searches(models["tag133"], cartpole_obs(48, 1007), 11000 + np.arange(48), 50, True, g))
    save("search_tag137_S25.npz", **run_searches(models["tag137"], cartpole_obs(32, 1008), 12000 + np.arange(32), 25, True, 0.95))
    save("search_synA4_S50.npz", **run_searches(models["synA4"], cartpole_obs(64, 1009, dim=8, scale=1.0), 13000 + np.arange(64), 50, True, g))
    save("search_synA3_S30.npz", **run_searches(models["synA3"], cartpole_obs(32, 1010, dim=6, scale=1.0), 14000 + np.arange(32), 30, True, g))


if __name__ == "__main__":
    main()
