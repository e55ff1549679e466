#!/usr/bin/env python3
"""Golden vectors of the 'vision_model' inference path (container-only; never runs on the GPU box).

Builds the reference's own ResNet-v2 networks (neural_network_vision_model.py through
``Muzero(model_structure='vision_model')``, imported by path as make_fixtures.py does; torch only —
game.py and its torchvision transform are not needed) and records the three inference wrappers
(muzero_model.py:828-905, eval mode: BatchNorm on its running statistics, Dropout2d off) on seeded
inputs.  Every tensor, the BatchNorm running statistics included, is replaced by seeded random
values: with torch's fresh running_mean = 0 / running_var = 1 the normalisation would be untested.

Observations are (1, 3, 98, 98) floats in [0, 1] (game.py:66-69: ToTensor + Resize); they are
regenerated from `obs_seed` by tests (numpy RandomState.uniform, float32) rather than stored.

Blob order: muzero_amd.model.vision_blob_layout (= include/mzs_layout.h mzs_vision_layout).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_vision_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
import make_fixtures as mf  # noqa: E402

mm = mf.mm


def vision_model(A, Sd, H, L, seed):
    mz = mm.Muzero(model_structure="vision_model", observation_space_dimensions=mf.Box((98, 98, 3)),
                   action_space_dimensions=mf.Discrete(A), state_space_dimensions=Sd,
                   hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=5, device="cpu")
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                    m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                    m.running_mean.copy_(0.3 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                elif isinstance(m, torch.nn.Conv2d):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.2 / np.sqrt(m.weight[0].numel())))
                    if m.bias is not None:
                        m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                elif isinstance(m, torch.nn.Linear):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.5 / np.sqrt(m.weight.shape[1])))
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    return mz


def pack_vision(mz, L):
    """The distinct tensors in muzero_amd.model.vision_blob_layout order."""
    f = lambda t: t.detach().numpy().astype(np.float32).ravel()  # noqa: E731

    def bn(m):
        return [f(m.weight), f(m.bias), f(m.running_mean), f(m.running_var)]

    def res(block):          # Residual_block.sequential_container: bn, relu, drop, conv_1, bn, relu, drop, conv_3, ...
        sc = block.sequential_container
        assert sc[3] is sc[11] and sc[0] is sc[4] is sc[8] and sc[7] is not sc[3]
        return [f(sc[3].weight), f(sc[7].weight)] + bn(sc[0])

    def mlp(seq):
        n = len(seq)
        parts = [f(seq[0].weight), f(seq[0].bias)]
        if L > 0:
            assert all(seq[2 * i + 2] is seq[2] for i in range(L))       # one Linear object, L uses
            parts += [f(seq[2].weight), f(seq[2].bias)]
        return parts + [f(seq[n - 1].weight), f(seq[n - 1].bias)]

    r, d, p = mz.representation_function, mz.dynamics_function, mz.prediction_function
    ds = r.sequential_downsampler[0].sequential_container
    assert ds[1] is ds[2] and ds[4] is ds[5] is ds[7] is ds[8] is ds[9]
    rc = r.sequential_convolution_activation
    assert rc[3] is r.sequential_downsampler[1]
    parts = [f(ds[0].weight)] + res(ds[1]) + [f(ds[3].weight)] + res(ds[4]) + res(r.sequential_downsampler[1])
    parts += [f(rc[0].weight)] + bn(rc[1])
    dc = d.sequential_container
    parts += [f(dc[0].weight)] + bn(dc[1])
    if L == 0:
        raise SystemExit("vision fixtures are made with number_of_hidden_layer >= 1 (with 0 the Sequentials "
                         "hold no Residual_block to read d_res / p_res from)")
    parts += res(dc[3])
    dr = d.sequential_reward
    parts += [f(dr[0].weight), f(dr[0].bias)] + mlp(dr[2])
    parts += res(p.resnet[0]) + [f(p.nn_value[0].weight), f(p.nn_value[0].bias), f(p.nn_policy[0].weight),
                                 f(p.nn_policy[0].bias)]
    parts += mlp(p.nn_value[2]) + mlp(p.nn_policy[2])
    return np.concatenate(parts)


def vision_vectors(mz, A, n, obs_seed, seed):
    rs = np.random.RandomState(seed)
    obs = np.random.RandomState(obs_seed).uniform(0.0, 1.0, size=(n, 3, 98, 98)).astype(np.float32)
    hid = rs.uniform(-1.0, 2.0, size=(n, 3, 7, 7)).astype(np.float32)
    act = rs.randint(0, A, size=n).astype(np.int32)
    Sd = mz.state_dimension
    out = dict(obs_seed=np.int64(obs_seed), hidden=hid, action=act,
               repr_hidden=np.zeros((n, 3, 7, 7), np.float32), ini_policy=np.zeros((n, A), np.float32),
               ini_value=np.zeros(n, np.float32), dyn_reward=np.zeros(n, np.float32),
               dyn_hidden=np.zeros((n, 3, 7, 7), np.float32), dyn_reward_logits=np.zeros((n, Sd), np.float32),
               pred_policy=np.zeros((n, A), np.float32), pred_value=np.zeros(n, np.float32),
               rec_policy=np.zeros((n, A), np.float32), rec_value=np.zeros(n, np.float32))
    for i in range(n):
        h0 = mz.representation_function_inference(obs[i:i + 1])
        out["repr_hidden"][i] = h0.numpy()[0]
        p, v = mz.prediction_function_inference(h0)
        out["ini_policy"][i], out["ini_value"][i] = p[0], v
        r, h2 = mz.dynamics_function_inference(torch.tensor(hid[i:i + 1]), int(act[i]))
        out["dyn_reward"][i], out["dyn_hidden"][i] = r, h2.numpy()[0]
        with torch.no_grad():
            plane = mz.one_hot_encode(int(act[i]), torch.tensor(hid[i:i + 1]))
            out["dyn_reward_logits"][i] = mz.dynamics_function(torch.tensor(hid[i:i + 1]), plane)[0].numpy()[0]
        p, v = mz.prediction_function_inference(torch.tensor(hid[i:i + 1]))
        out["pred_policy"][i], out["pred_value"][i] = p[0], v
        p, v = mz.prediction_function_inference(h2)
        out["rec_policy"][i], out["rec_value"][i] = p[0], v
    # intermediate maps of the first observation through Down_sample: where a mismatch starts
    with torch.no_grad():
        x = torch.tensor(obs[:1])
        ds = mz.representation_function.sequential_downsampler[0].sequential_container
        for j, layer in enumerate(ds):
            x = layer(x)
            out[f"ds_stage{j}"] = x.numpy()[0].copy()
    return out


def vision_searches(mz, A, Sd, H, L, n, obs_seed, seed0, S, train=True, gamma=0.997):
    """The reference's Monte_carlo_tree_search.run (monte_carlo_tree_search.py:224-265) on the vision networks
    through make_fixtures.RecordingMCTS: n searches of S simulations on seeded (1, 3, 98, 98) images, with the
    per-simulation model outputs, consumed streams, final trees and decision margins of the MLP search
    fixtures.  The images are regenerated from obs_seed by the tests (conftest.fixture_obs), not stored."""
    obs = np.random.RandomState(obs_seed).uniform(0.0, 1.0, size=(n, 3, 98, 98)).astype(np.float32)
    dims = dict(obs_dim=98, A=A, S_dim=Sd, H=H, L=L, hid_n=147)
    rec = mf.run_searches(mz, obs, seed0 + np.arange(n), S, train, gamma, dims=dims, store_obs=False)
    rec["obs_seed"] = np.int64(obs_seed)
    rec["kind"] = np.int64(2)
    return rec


def main():
    torch.set_num_threads(1)
    for name, (A, Sd, H, L, seed) in {"visA": (2, 31, 64, 1, 41), "visB": (4, 21, 32, 2, 42)}.items():
        mz = vision_model(A, Sd, H, L, seed)
        for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
            net.eval()
        blob = pack_vision(mz, L)
        from muzero_amd.model import vision_blob_from_state_dicts, vision_blob_layout
        assert blob.size == sum(int(np.prod(s)) for _, s in vision_blob_layout(A, Sd, H, L)), blob.size
        # the drop-in's checkpoint mapping (Muzero.load_reference_state_dicts) against the module walk above
        mapped = vision_blob_from_state_dicts(mz.representation_function.state_dict(),
                                              mz.dynamics_function.state_dict(),
                                              mz.prediction_function.state_dict(), L)
        assert np.array_equal(mapped, blob), "vision_blob_from_state_dicts disagrees with pack_vision"
        mf.save(f"weights_{name}.npz", blob=blob, kind=np.int64(2), obs_dim=np.int64(98), A=np.int64(A),
                S_dim=np.int64(Sd), H=np.int64(H), L=np.int64(L))
        mf.save(f"model_L2_{name}.npz", **vision_vectors(mz, A, 12, 900 + seed, 31 + seed))
        S = 30 if name == "visA" else 20
        mf.save(f"search_{name}_S{S}.npz", **vision_searches(mz, A, Sd, H, L, 16, 1900 + seed, 15000 + 100 * seed, S))
    mz = vision_model(2, 31, 64, 1, 41)
    for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
        net.eval()
    mf.save("search_visA_S12_notrain.npz", **vision_searches(mz, 2, 31, 64, 1, 8, 1990, 16000, 12, train=False))


if __name__ == "__main__":
    main()
