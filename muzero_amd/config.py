"""Experiment config loader.

Reads the reference's JSON schema unchanged (config/experiment_*.json: sections `game`,
`random_seed`, `muzero`, `replaybuffer`, `monte_carlo_tree_search`, `gameplay`,
`learning_cycle`, `play_game_from_checkpoint`; muzero_cli.py:48-114 consumes them) and one
optional extra section, `mi355x`, for the batched engine:

    "mi355x": {"concurrent_games": 4096, "num_gpus": 8, "engine": "fused"}

Reference configs (no `mi355x` section) load with the defaults below.
"""
import json

from . import _lib

REQUIRED_SECTIONS = ("game", "random_seed", "muzero", "replaybuffer", "monte_carlo_tree_search",
                     "gameplay", "learning_cycle", "play_game_from_checkpoint")
MI355X_DEFAULTS = {"concurrent_games": 4096, "num_gpus": 1, "engine": "fused"}
_ENGINES = {"fused": _lib.ENGINE_FUSED, "fused_generic": _lib.ENGINE_FUSED_GENERIC}


def load_config(path):
    with open(path) as f:
        cfg = json.load(f)
    missing = [s for s in REQUIRED_SECTIONS if s not in cfg]
    if missing:
        raise KeyError(f"{path}: missing config sections {missing}")
    extra = dict(MI355X_DEFAULTS)
    extra.update(cfg.get("mi355x", {}))
    if extra["engine"] not in _ENGINES:
        raise ValueError(f"mi355x.engine must be one of {sorted(_ENGINES)}")
    cfg["mi355x"] = extra
    return cfg


def mcts_kwargs(cfg):
    """Constructor arguments of Monte_carlo_tree_search (monte_carlo_tree_search.py:38)."""
    m = cfg["monte_carlo_tree_search"]
    return dict(pb_c_base=m["pb_c_base"], pb_c_init=m["pb_c_init"], discount=m["discount"],
                root_dirichlet_alpha=m["root_dirichlet_alpha"],
                root_exploration_fraction=m["root_exploration_fraction"])


def muzero_kwargs(cfg, observation_space_dimensions, action_space_dimensions):
    """Constructor arguments of Muzero as muzero_cli.py:72-91 builds them."""
    m = cfg["muzero"]
    return dict(model_structure=m["model_structure"],
                observation_space_dimensions=observation_space_dimensions,
                action_space_dimensions=action_space_dimensions,
                state_space_dimensions=m["state_space_dimensions"],
                hidden_layer_dimensions=m["hidden_layer_dimensions"],
                number_of_hidden_layer=m["number_of_hidden_layer"],
                k_hypothetical_steps=m["k_hypothetical_steps"], optimizer=m["optimizer"],
                lr_scheduler=m["lr_scheduler"], learning_rate=m["learning_rate"],
                loss_type=m["loss_type"], num_of_epoch=m["num_of_epoch"], device="cuda",
                load=m["load"], use_amp=False, scaler_on=False, bin_method=m["bin_method"],
                bin_decomposition_number=m["bin_decomposition_number"])


def engine_id(cfg):
    return _ENGINES[cfg["mi355x"]["engine"]]
