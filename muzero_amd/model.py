"""`Muzero` — the reference's model wrapper (muzero_model.py:14) with the inference half
running on the HIP kernels of libmzsearch.so.

Kept from the reference (drop-in surface): the 19 constructor keyword
arguments (muzero_model.py:16-35), the attributes self_play.py / muzero_cli.py read
(`observation_dimension`, `action_dimension`, `is_RGB`, `action_dictionnary`,
`k_hypothetical_steps`, `random_tag`, `store_loss`, `device`, `state_dimension`,
`hidden_layer_dimension`, `number_of_hidden_layer`) and the three inference methods
with their return types (muzero_model.py:828-905).

`model_structure` "mlp_model" (neural_network_mlp_model.py), "lstm_model"
(neural_network_lstm_model.py) and "vision_model" (neural_network_vision_model.py, inference and
search only) have kernels.  Weights live in one packed fp32 blob
(include/mzs_layout.h).
"""
import json
import os

import numpy as np
import torch

from . import _lib
from .engine import SearchEngine


def _space_size(space):
    """muzero_model.py:476-486 obs_space: Discrete -> 1, Box -> prod(shape), tuple -> sum."""
    if isinstance(space, (int, np.integer)):
        return int(space)
    if isinstance(space, (tuple, list)):
        return int(sum(_space_size(s) for s in space))
    if hasattr(space, "shape") and space.shape not in (None, ()):
        return int(np.prod(space.shape))
    if hasattr(space, "n"):
        return 1
    raise TypeError(f"unsupported observation space {space!r}")


def _action_dictionary(space, bin_method, bins):
    """Gym_space_transform.design_observation_space (muzero_model.py:994-1051) for the
    spaces the configs use: Discrete(n) -> [0..n-1]; Box -> `bins` samples per dimension,
    cartesian product."""
    import itertools
    if isinstance(space, (int, np.integer)):
        return list(range(int(space)))

    def one(sp):
        if hasattr(sp, "n"):
            return [list(range(int(sp.n)))]
        low = np.asarray(sp.low, np.float32).ravel()
        high = np.asarray(sp.high, np.float32).ravel()
        parts = []
        for lo, hi in zip(low, high):
            if "int" in str(sp.dtype):
                parts.append(list(range(int(lo), int(hi))))
            elif bin_method == "linear_bin":
                parts.append(torch.linspace(float(lo), float(hi), steps=bins).tolist())
            else:
                parts.append(torch.distributions.uniform.Uniform(float(lo), float(hi)).sample([bins]).tolist())
        return parts

    parts = []
    for sp in (space if isinstance(space, (tuple, list)) else [space]):
        parts.extend(one(sp))
    if len(parts) > 1:
        return list(itertools.product(*parts))
    return parts[0]


def lstm_blob_layout(obs_dim, A, S_dim, H, L):
    """'lstm_model' blob (include/mzs_layout.h mzs_lstm_layout): the state_dict order of
    neural_network_lstm_model.py's three modules."""
    def head(p, in_dim, out):
        t = [(f"{p}_lin_w", (H, in_dim)), (f"{p}_lin_b", (H,))]
        for l in range(L):
            in_l = H if l == 0 else out
            t += [(f"{p}_w_ih_l{l}", (4 * out, in_l)), (f"{p}_w_hh_l{l}", (4 * out, out)),
                  (f"{p}_b_ih_l{l}", (4 * out,)), (f"{p}_b_hh_l{l}", (4 * out,))]
        return t
    return ([("r_w", (S_dim, obs_dim)), ("r_b", (S_dim,))] + head("d_rew", S_dim + A, S_dim) +
            head("d_st", S_dim + A, S_dim) + head("p_pol", S_dim, A) + head("p_val", S_dim, S_dim))


VISION_SIDE, VISION_CHANNELS, VISION_HIDDEN_SIDE = 98, 3, 7      # muzero_model.py:312-314; 98 -> 49 -> 25 -> 13 -> 7
VISION_HIDDEN = VISION_CHANNELS * VISION_HIDDEN_SIDE ** 2           # floats of one hidden state


def vision_blob_layout(A, S_dim, H, L):
    """'vision_model' blob (include/mzs_layout.h mzs_vision_layout): every DISTINCT tensor of
    neural_network_vision_model.py's three modules once (its Sequentials repeat one Residual_block,
    one BatchNorm2d and one Linear object in several places), BatchNorm running statistics included,
    num_batches_tracked left out.  conv_a = the block's first and last 3x3 convolution
    (`convolution_1`, :49-51, used at positions 3 and 11), conv_b the middle one (`convolution_3`)."""
    C, hid = VISION_CHANNELS, VISION_HIDDEN

    def bn(p, c):
        return [(f"{p}_w", (c,)), (f"{p}_b", (c,)), (f"{p}_mean", (c,)), (f"{p}_var", (c,))]

    def res(p, c):
        return [(f"{p}_conv_a", (c, c, 3, 3)), (f"{p}_conv_b", (c, c, 3, 3))] + bn(f"{p}_bn", c)

    def mlp(p, out):
        mid = [(f"{p}_mid_w", (H, H)), (f"{p}_mid_b", (H,))] if L > 0 else []
        return [(f"{p}_in_w", (H, hid)), (f"{p}_in_b", (H,))] + mid + [(f"{p}_out_w", (out, H)), (f"{p}_out_b", (out,))]

    return ([("r_ds_conv_in", (C // 2, C, 3, 3))] + res("r_ds_res_in", C // 2) + [("r_ds_conv_out", (C, C // 2, 3, 3))] +
            res("r_ds_res_out", C) + res("r_res", C) + [("r_conv", (C, C, 3, 3))] + bn("r_bn", C) +     # :141-142, unused path
            [("d_conv", (C, C + 1, 3, 3))] + bn("d_bn", C) + res("d_res", C) +
            [("d_conv_rew", (C, C + 1, 1, 1)), ("d_conv_rew_b", (C,))] + mlp("d_fc", S_dim) +
            res("p_res", C) + [("p_conv_val", (C, C, 1, 1)), ("p_conv_val_b", (C,)),
                               ("p_conv_pol", (C, C, 1, 1)), ("p_conv_pol_b", (C,))] +
            mlp("p_v", S_dim) + mlp("p_p", A))


def blob_layout(obs_dim, A, S_dim, H, L, kind=0, precision=0):
    """(name, shape) of every tensor in the packed blob, in order (include/mzs_layout.h).  The blob
    is fp32 whatever the compute precision (bf16 rounds a copy of it on the device)."""
    if kind == _lib.MODEL_LSTM:
        return lstm_blob_layout(obs_dim, A, S_dim, H, L)
    if kind == _lib.MODEL_VISION:
        return vision_blob_layout(A, S_dim, H, L)
    mid = lambda p: [(f"{p}_mid_w", (H, H)), (f"{p}_mid_b", (H,))] if L > 0 else []  # noqa: E731
    return ([("r_in_w", (H, obs_dim)), ("r_in_b", (H,))] + mid("r") +
            [("r_out_w", (S_dim, H)), ("r_out_b", (S_dim,))] +
            [("d_in_w", (H, S_dim + A)), ("d_in_b", (H,))] + mid("d") +
            [("d_rew_w", (S_dim, H)), ("d_rew_b", (S_dim,)), ("d_st_w", (S_dim, H)), ("d_st_b", (S_dim,))] +
            [("p_in_w", (H, S_dim)), ("p_in_b", (H,))] + mid("p") +
            [("p_pol_w", (A, H)), ("p_pol_b", (A,)), ("p_val_w", (S_dim, H)), ("p_val_b", (S_dim,))])


def blob_from_state_dicts(repr_sd, dyn_sd, pred_sd, L):
    """Pack `state_dict()`s of the reference's three MLP modules
    (neural_network_mlp_model.py:7-128; shared trunk tensors appear once)."""
    last = 2 * L + 2
    g = lambda sd, k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], np.float32)  # noqa: E731
    parts = [g(repr_sd, "state_norm.0.weight"), g(repr_sd, "state_norm.0.bias")]
    if L > 0:
        parts += [g(repr_sd, "state_norm.2.weight"), g(repr_sd, "state_norm.2.bias")]
    parts += [g(repr_sd, f"state_norm.{last}.weight"), g(repr_sd, f"state_norm.{last}.bias")]
    parts += [g(dyn_sd, "reward.0.weight"), g(dyn_sd, "reward.0.bias")]
    if L > 0:
        parts += [g(dyn_sd, "reward.2.weight"), g(dyn_sd, "reward.2.bias")]
    parts += [g(dyn_sd, f"reward.{last}.weight"), g(dyn_sd, f"reward.{last}.bias"),
              g(dyn_sd, f"next_state_normalized.{last}.weight"), g(dyn_sd, f"next_state_normalized.{last}.bias")]
    parts += [g(pred_sd, "policy.0.weight"), g(pred_sd, "policy.0.bias")]
    if L > 0:
        parts += [g(pred_sd, "policy.2.weight"), g(pred_sd, "policy.2.bias")]
    parts += [g(pred_sd, f"policy.{last}.weight"), g(pred_sd, f"policy.{last}.bias"),
              g(pred_sd, f"value.{last}.weight"), g(pred_sd, f"value.{last}.bias")]
    return np.concatenate([p.ravel() for p in parts]).astype(np.float32)


def vision_blob_from_state_dicts(repr_sd, dyn_sd, pred_sd, L):
    """Pack the `state_dict()`s of the reference's three vision modules (neural_network_vision_model.py)
    into vision_blob_layout order.  Their Sequentials repeat module objects, so a state_dict lists
    one tensor under several keys: the first occurrence is taken."""
    g = lambda sd, k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], np.float32)  # noqa: E731

    def bn(sd, p):
        return [g(sd, p + "weight"), g(sd, p + "bias"), g(sd, p + "running_mean"), g(sd, p + "running_var")]

    def res(sd, p):          # Residual_block.sequential_container: 0 bn, 3 conv_1 (= conv_a), 7 conv_3 (= conv_b)
        p += "sequential_container."
        return [g(sd, p + "3.weight"), g(sd, p + "7.weight")] + bn(sd, p + "0.")

    def mlp(sd, p):
        last = 2 * L + 2
        mid = [g(sd, p + "2.weight"), g(sd, p + "2.bias")] if L > 0 else []
        return [g(sd, p + "0.weight"), g(sd, p + "0.bias")] + mid + [g(sd, p + f"{last}.weight"), g(sd, p + f"{last}.bias")]

    if L < 1:
        raise ValueError("vision_model: number_of_hidden_layer >= 1 (with 0 the dynamics and prediction "
                         "Sequentials hold no residual block)")
    ds = "sequential_downsampler.0.sequential_container."
    parts = ([g(repr_sd, ds + "0.weight")] + res(repr_sd, ds + "1.") + [g(repr_sd, ds + "3.weight")] +
             res(repr_sd, ds + "4.") + res(repr_sd, "sequential_downsampler.1.") +
             [g(repr_sd, "sequential_convolution_activation.0.weight")] +
             bn(repr_sd, "sequential_convolution_activation.1."))
    parts += ([g(dyn_sd, "sequential_container.0.weight")] + bn(dyn_sd, "sequential_container.1.") +
              res(dyn_sd, "sequential_container.3.") +
              [g(dyn_sd, "sequential_reward.0.weight"), g(dyn_sd, "sequential_reward.0.bias")] +
              mlp(dyn_sd, "sequential_reward.2."))
    parts += (res(pred_sd, "resnet.0.") +
              [g(pred_sd, "nn_value.0.weight"), g(pred_sd, "nn_value.0.bias"),
               g(pred_sd, "nn_policy.0.weight"), g(pred_sd, "nn_policy.0.bias")] +
              mlp(pred_sd, "nn_value.2.") + mlp(pred_sd, "nn_policy.2."))
    return np.concatenate([p.ravel() for p in parts]).astype(np.float32)


class Muzero:
    def __init__(self, model_structure="mlp_model", observation_space_dimensions=None,
                 action_space_dimensions=None, state_space_dimensions=9, k_hypothetical_steps=10,
                 learning_rate=1e-3, optimizer="adam", lr_scheduler=None, loss_type="general",
                 device="cuda", num_of_epoch=300, hidden_layer_dimensions=16,
                 number_of_hidden_layer=1, load=False, type_format=torch.float32, use_amp=False,
                 scaler_on=False, bin_method="uniform_bin", bin_decomposition_number=10, bf16_mode="x3"):
        """The reference's 19 arguments (muzero_model.py:16-35) plus one of ours, last: bf16_mode, what
        type_format=torch.bfloat16 means on the device — "x3" (default): weights and layer inputs as hi + lo
        bfloat16 pairs, three matrix products per layer, within north_star's 1e-2 of the reference on EVERY
        fixture model (measured <= 5e-4); "fast": both rounded to one bfloat16, a quarter of the matrix
        instructions and half the weight registers, inside 1e-2 on the CartPole checkpoints 123 / 133 only
        (include/mzs_numeric.h "dot_bf16x3" / "dot_bf16", tests/test_bf16_oracle.py)."""
        self.reset(model_structure, observation_space_dimensions, action_space_dimensions,
                   state_space_dimensions, k_hypothetical_steps, learning_rate, optimizer,
                   lr_scheduler, loss_type, device, num_of_epoch, hidden_layer_dimensions,
                   number_of_hidden_layer, load, type_format, use_amp, scaler_on, bin_method,
                   bin_decomposition_number, bf16_mode)

    def reset(self, model_structure=None, observation_space_dimensions=None,
              action_space_dimensions=None, state_space_dimensions=None, k_hypothetical_steps=None,
              learning_rate=1e-3, optimizer="adam", lr_scheduler=None, loss_type="general",
              device="cuda", num_of_epoch=300, hidden_layer_dimensions=64,
              number_of_hidden_layer=1, load=False, type_format=torch.float32, use_amp=False,
              scaler_on=False, bin_method="uniform_bin", bin_decomposition_number=10, bf16_mode="x3"):
        if bf16_mode not in ("x3", "fast"):
            raise ValueError(f"bf16_mode={bf16_mode!r}: 'x3' (compensated) or 'fast'")
        if model_structure not in (None, "mlp_model", "lstm_model", "vision_model"):
            raise NotImplementedError(f"model_structure={model_structure!r}: 'mlp_model', 'lstm_model' and "
                                      "'vision_model' (inference and search) have MI355X kernels")
        if model_structure == "vision_model" and (int(number_of_hidden_layer) < 1 or type_format is torch.bfloat16):
            raise NotImplementedError("vision_model: number_of_hidden_layer >= 1, torch.float32")
        if model_structure == "lstm_model" and not 1 <= int(number_of_hidden_layer) <= 4:
            raise ValueError("lstm_model: number_of_hidden_layer is nn.LSTM's num_layers, 1..4 here")
        if type_format not in (torch.float32, torch.bfloat16, None):
            raise NotImplementedError(f"type_format={type_format}: torch.float32 (bit-exact contract) or "
                                      "torch.bfloat16 (1e-2 contract, include/mzs_numeric.h)")
        if use_amp and type_format is not torch.bfloat16:
            raise NotImplementedError("use_amp: mixed precision exists as type_format=torch.bfloat16 only")
        self.model_structure = model_structure or "mlp_model"
        self.state_dimension = state_space_dimensions
        self.hidden_layer_dimension = hidden_layer_dimensions
        self.number_of_hidden_layer = number_of_hidden_layer
        self.k_hypothetical_steps = k_hypothetical_steps
        self.loss_type, self.lr, self.opt, self.sch = loss_type, learning_rate, optimizer, lr_scheduler
        self.epoch, self.count = num_of_epoch, 0
        self.device = "cuda" if device in ("cpu", None) else device  # there is no CPU path
        # bf16: weights and layer inputs rounded to bfloat16 on the device, fp32 accumulation and
        # epilogues, fp32 master weights and fp32 training.  The reference sets use_amp for every
        # "float16" type (muzero_model.py:202: bfloat16 matches the substring) and keeps fp32 weights
        # under torch.autocast; its own bf16 VALUE output collapses to -1 (inverse_transform_with_support
        # runs in bfloat16 outside autocast, muzero_model.py:903) — not reproduced, see DESIGN.md.
        self.type_format = torch.bfloat16 if type_format is torch.bfloat16 else torch.float32
        self.bf16_mode = bf16_mode
        self.precision = ((_lib.PRECISION_BF16X3 if bf16_mode == "x3" else _lib.PRECISION_BF16)
                          if self.type_format is torch.bfloat16 else _lib.PRECISION_F32)
        self.use_amp, self.scaler_on = self.precision != _lib.PRECISION_F32, False
        self.random_tag = np.random.randint(0, 100000000)  # muzero_model.py:213 (global stream)
        self.store_loss = []
        self.bin_method, self.bin_decomposition_number = bin_method, bin_decomposition_number
        self.is_RGB = "vision" in self.model_structure       # muzero_model.py:282
        self._engine = None
        self._trainer = None
        self._blob = None
        self._blob_stale = False
        self.weights_version = 0
        if not load:
            # vision models see a resized 98 x 98 RGB render whatever the environment (muzero_model.py:312-314)
            self.observation_dimension = ((VISION_SIDE, VISION_SIDE, 3) if self.is_RGB
                                          else _space_size(observation_space_dimensions))
            self.action_dictionnary = _action_dictionary(action_space_dimensions, bin_method,
                                                         bin_decomposition_number)
            self.action_dimension = len(self.action_dictionnary)
            self.initiate_model_weight()

    # ------------------------------------------------------------------ weights
    @property
    def dims(self):
        kind = {"lstm_model": _lib.MODEL_LSTM, "vision_model": _lib.MODEL_VISION}.get(self.model_structure, _lib.MODEL_MLP)
        obs = VISION_SIDE if self.is_RGB else int(self.observation_dimension)
        d = (obs, int(self.action_dimension), int(self.state_dimension), int(self.hidden_layer_dimension),
             int(self.number_of_hidden_layer), kind)
        return d + (self.precision,) if self.precision else d

    def initiate_model_weight(self):
        """weights_init (neural_network_mlp_model.py:287-292): N(0, 1/137.035999) for every
        Linear weight and bias, drawn from torch's global generator."""
        if self.model_structure == "vision_model":
            # muzero_model.py:326-331 skips weights_init for vision models: torch's own defaults stay —
            # Conv2d / Linear kaiming_uniform(a = sqrt 5) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and
            # bias, BatchNorm2d weight 1, bias 0, running_mean 0, running_var 1
            parts, fan_in = [], 1
            for name, shape in blob_layout(*self.dims):
                if name.endswith(("_bn_w", "_var")):
                    parts.append(torch.ones(shape).reshape(-1))
                elif name.endswith(("_bn_b", "_mean")):
                    parts.append(torch.zeros(shape).reshape(-1))
                else:
                    if len(shape) > 1:            # a bias follows its weight and shares its fan-in
                        fan_in = int(np.prod(shape[1:]))
                    parts.append((torch.rand(shape) * 2 - 1).reshape(-1) / np.sqrt(fan_in))
            self.set_weights(torch.cat(parts).numpy())
            return
        if self.model_structure == "lstm_model":
            # weights_init touches nn.Linear only; nn.LSTM keeps torch's own U(-1/sqrt(out), 1/sqrt(out))
            parts = []
            for name, shape in blob_layout(*self.dims):
                if "_lin_" in name or name in ("r_w", "r_b"):
                    parts.append(torch.normal(0.0, 1 / 137.035999, size=shape).reshape(-1))
                else:
                    k = 1.0 / np.sqrt(shape[0] // 4)
                    parts.append((torch.rand(shape) * 2 - 1).reshape(-1) * k)
            self.set_weights(torch.cat(parts).numpy())
            return
        n = sum(int(np.prod(s)) for _, s in blob_layout(*self.dims))
        self.set_weights(torch.normal(0.0, 1 / 137.035999, size=(n,)).numpy())

    def set_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        n = sum(int(np.prod(s)) for _, s in blob_layout(*self.dims))
        if blob.size != n:
            raise ValueError(f"weight blob has {blob.size} floats, expected {n}")
        self._blob = blob
        self._blob_stale = False
        self.weights_version += 1
        if self._engine is not None:
            self._engine.set_weights(blob)

    def weights_blob(self):
        """The packed blob, fetched from the device first if training steps ran since the last copy
        (note_trained marks it stale instead of copying 51 KB back after every step)."""
        if self._blob_stale and self._trainer is not None:
            self._blob = self._trainer.weights()
            self._blob_stale = False
        return self._blob

    def note_trained(self, trainer, loss=None):
        """Bookkeeping after a training step that went through `trainer` directly (SelfPlay.train_step):
        what Muzero.train does after its own steps (muzero_model.py:786-789)."""
        self.count = trainer.count
        self._blob_stale = True
        self.weights_version += 1
        if loss is not None:
            loss = np.asarray(loss, np.float32)
            self.store_loss.append([torch.tensor(loss[-1])] + [torch.tensor(x) for x in loss[:-1]])

    def named_weights(self):
        out, p, blob = {}, 0, self.weights_blob()
        for name, shape in blob_layout(*self.dims):
            n = int(np.prod(shape))
            out[name] = blob[p:p + n].reshape(shape)
            p += n
        return out

    def load_reference_state_dicts(self, repr_sd, dyn_sd, pred_sd):
        if self.model_structure == "vision_model":
            self.set_weights(vision_blob_from_state_dicts(repr_sd, dyn_sd, pred_sd, self.number_of_hidden_layer))
            return
        if self.model_structure == "lstm_model":     # blob order == state_dict order
            g = lambda t: np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, np.float32).ravel()  # noqa: E731
            self.set_weights(np.concatenate([g(t) for sd in (repr_sd, dyn_sd, pred_sd) for t in sd.values()]))
            return
        self.set_weights(blob_from_state_dicts(repr_sd, dyn_sd, pred_sd, self.number_of_hidden_layer))

    # ---------------------------------------------------------------- inference
    def _inference_engine(self):
        if self._engine is None:
            self._engine = SearchEngine(self.dims, max_games=1, max_simulations=0)
            self._engine.set_weights(self.weights_blob())
        return self._engine

    @staticmethod
    def _as_batch(x):
        if torch.is_tensor(x):
            x = x.detach().to(torch.float32)
        else:
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        return x.reshape(1, -1) if x.dim() == 1 else x.reshape(x.shape[0], -1)

    def _as_images(self, x):
        """vision observations: (B, 3, 98, 98) float as game.py's transform makes them (:66-69)"""
        x = x.detach().to(torch.float32) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x[None] if x.dim() == 3 else x
        if tuple(x.shape[1:]) != (3, VISION_SIDE, VISION_SIDE):
            raise ValueError(f"vision_model observation {tuple(x.shape)}: expected (B, 3, 98, 98)")
        return x.contiguous()

    def _hidden_out(self, hid):
        """hidden states as the reference returns them: (B, 3, 7, 7) maps for the vision model"""
        hid = hid.cpu()
        return hid.reshape(-1, VISION_CHANNELS, VISION_HIDDEN_SIDE, VISION_HIDDEN_SIDE) if self.is_RGB else hid

    def representation_function_inference(self, state):
        """muzero_model.py:828-845: observation -> cpu Tensor (B, state_dimension)
        (vision model: (B, 3, 98, 98) -> (B, 3, 7, 7))."""
        obs = self._as_images(state) if self.is_RGB else self._as_batch(state)
        hid, _, _ = self._inference_engine().initial_inference(obs, want_prediction=False)
        return self._hidden_out(hid)

    def dynamics_function_inference(self, state_normalized, action):
        """muzero_model.py:848-880: -> (np.float32 reward, cpu Tensor (1, state_dimension))."""
        h = self._as_batch(state_normalized)
        a = torch.as_tensor(action)
        if a.dim() == 2:            # one-hot rows (muzero_model.py:497-498)
            a = a.argmax(dim=1)
        a = a.reshape(-1).to(torch.int32)
        rew, nh, _, _ = self._inference_engine().recurrent_inference(h, a, want_prediction=False)
        return rew.cpu().numpy()[0], self._hidden_out(nh)

    def prediction_function_inference(self, state_normalized, reanalyze=False):
        """muzero_model.py:883-905: -> (np.ndarray (B, A) softmax policy, np.float32 value),
        or the value array when reanalyze=True."""
        pol, val = self._inference_engine().prediction_inference(self._as_batch(state_normalized))
        val = val.cpu().numpy()
        return pol.cpu().numpy(), (val if reanalyze else val[0])

    def train(self, batch):
        """muzero_model.py:791-811: one optimizer step on `batch` = (samples, importance-sampling
        ratio, game positions) as ReplayBuffer.sample_batch returns it (replay_buffer.py:131-166).
        Returns (new_priority: list of K arrays (B, 1), batch_game_position); appends
        [loss_nn, value_0, policy_0, value_1, policy_1, reward_1, ...] to store_loss.
        With torch.distributed initialised each rank passes its shard of the batch.
        'lstm_model': every head's nn.LSTM reads the (B, H) batch as ONE sequence of length B
        (neural_network_lstm_model.py:35), and so do the kernels: the order of the samples matters,
        and a rank's shard is a sequence of its own.
        'vision_model': training mode as in the reference — BatchNorm2d on the batch statistics of this
        rank's shard (running statistics move), Dropout2d masks from a device generator
        (Trainer.set_dropout seeds it, or injects recorded masks)."""
        from . import train as tr
        if self.loss_type not in tr.LOSS_TYPES:
            raise NotImplementedError(f"loss_type={self.loss_type!r}: the reference knows {sorted(tr.LOSS_TYPES)} "
                                      "(muzero_model.py:350-403)")
        obs_floats = 3 * VISION_SIDE * VISION_SIDE if self.is_RGB else self.observation_dimension
        obs, actions, tv, trw, tp, ratio, game_pos = tr.reshape_batch(
            batch, obs_floats, self.action_dimension, self.k_hypothetical_steps)
        if self.is_RGB:       # (1, 3, 98, 98) tensors as game.py's transform makes them (:66-69)
            obs = obs.reshape(-1, 3, VISION_SIDE, VISION_SIDE)
        B = obs.shape[0]
        if self._trainer is not None and self._trainer.max_batch < B:
            # refuse BEFORE touching any state: a retry must not succeed on a fresh optimizer
            raise NotImplementedError(f"batch of {B} after a first train() call with {self._trainer.max_batch}: "
                                      "growing the batch would drop the optimizer state; create the "
                                      "model with the largest batch first")
        if self._trainer is None:
            self._trainer = tr.Trainer(self._inference_engine(), self.k_hypothetical_steps, B, self.opt, self.lr,
                                       self.sch, self.epoch, self.loss_type)
        t = self._trainer
        t.count = self.count
        prio = t.step(obs, actions, tv, trw, tp, ratio)
        self.count = t.count
        torch.cuda.synchronize(t.eng.device)
        loss = t.loss.cpu().numpy()
        self.store_loss.append([torch.tensor(loss[-1])] + [torch.tensor(x) for x in loss[:-1]])
        self._blob_stale = True
        self.weights_version += 1
        prio = prio.cpu().numpy()
        return [prio[k].reshape(-1, 1) for k in range(prio.shape[0])], game_pos

    # -------------------------------------------------------------- checkpoints
    def _init_variables(self):
        return {"observation_space_dimensions": self.observation_dimension,
                "action_space_dimensions": self.action_dimension,
                "state_space_dimensions": self.state_dimension,
                "hidden_layer_dimensions": self.hidden_layer_dimension,
                "number_of_hidden_layer": self.number_of_hidden_layer,
                "k_hypothetical_steps": self.k_hypothetical_steps, "learning_rate": self.lr,
                "optimizer": self.opt, "loss_type": self.loss_type, "lr_scheduler": self.sch,
                "num_of_epoch": self.epoch, "device": self.device, "random_tag": self.random_tag,
                "action_map": self.action_dictionnary, "model_structure": self.model_structure,
                "use_amp": self.use_amp, "type_format": str(self.type_format)}

    def save_model(self, directory="model_checkpoint", tag=None, model_update_condition=None):
        """Same JSON as muzero_model.py:923-941; the weights go to one `.npz` blob instead of
        three pickled modules."""
        if model_update_condition is not True:
            return
        os.makedirs(directory, exist_ok=True)
        if tag:
            self.random_tag = tag
        np.savez(f"{directory}/{self.random_tag}_muzero_weights.npz", blob=self.weights_blob())
        with open(f"{directory}/{self.random_tag}_muzero_init_variables.json", "w") as f:
            json.dump(self._init_variables(), f)

    def load_model(self, model_directory="model_checkpoint", tag=0, observation_space_dimensions=None,
                   type_format=torch.float32, device=None):
        try:
            with open(f"{model_directory}/{tag}_muzero_init_variables.json") as f:
                iv = json.load(f)
        except OSError:
            raise Exception(f"file not found at : {model_directory}/{tag}_muzero_init_variables.json")
        self.reset(model_structure=iv["model_structure"],
                   state_space_dimensions=iv["state_space_dimensions"],
                   k_hypothetical_steps=iv["k_hypothetical_steps"], optimizer=iv["optimizer"],
                   lr_scheduler=iv["lr_scheduler"], learning_rate=iv["learning_rate"],
                   loss_type=iv["loss_type"], device=device or iv["device"],
                   num_of_epoch=iv["num_of_epoch"],
                   hidden_layer_dimensions=iv["hidden_layer_dimensions"],
                   number_of_hidden_layer=iv["number_of_hidden_layer"], load=True,
                   type_format=torch.bfloat16 if "bfloat16" in str(iv.get("type_format", "")) else torch.float32)
        self.observation_dimension = iv["observation_space_dimensions"]
        self.action_dictionnary = iv["action_map"]
        self.action_dimension = len(self.action_dictionnary)
        self.random_tag = tag if tag else iv["random_tag"]
        path = f"{model_directory}/{iv['random_tag']}_muzero_weights.npz"
        self.set_weights(np.load(path)["blob"])
