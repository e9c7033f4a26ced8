"""Minimal hydra-style composition for the reference's YAML layout (hydra/omegaconf are not in the
image): `compose("dps", overrides=["data=moises", "model=musicldm"])` reads configs/<name>.yaml, then
resolves its `defaults:` list into sub-trees configs/<group>/<choice>.yaml (reference: run.py:147-151)."""
import os
import re
import yaml

# OmegaConf (what the reference's hydra loads these files with) reads `5e-5` as a float; PyYAML's YAML-1.1 resolver wants a dot in the
# mantissa and returns the string.  Same values as the reference sees: exponent-form strings become floats.
_EXP_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


class Node(dict):
    __getattr__ = dict.get

    @staticmethod
    def wrap(x):
        if isinstance(x, dict):
            return Node({k: Node.wrap(v) for k, v in x.items()})
        if isinstance(x, list):
            return [Node.wrap(v) for v in x]
        if isinstance(x, str) and _EXP_FLOAT.match(x):
            return float(x)
        return x


def compose(config_name, overrides=(), config_dir=CONFIG_DIR):
    with open(os.path.join(config_dir, f"{config_name}.yaml")) as f:
        top = yaml.safe_load(f) or {}
    choices = {}
    for d in top.pop("defaults", []):
        if isinstance(d, dict):
            choices.update(d)
    for o in overrides:
        k, v = o.split("=", 1)
        if k in choices or os.path.isdir(os.path.join(config_dir, k)):
            choices[k] = v
        else:                                   # dotted scalar override, e.g. scheduler.eta=0.5
            cur = top
            parts = k.split(".")
            for p in parts[:-1]:
                cur = cur.setdefault(p, {})
            cur[parts[-1]] = yaml.safe_load(v)
    cfg = {}
    for group, choice in choices.items():
        with open(os.path.join(config_dir, group, f"{choice}.yaml")) as f:
            cfg[group] = yaml.safe_load(f) or {}
    cfg.update(top)
    return Node.wrap(cfg)
