"""The reference's model-loading convention (riser/riser.py:21-42): for every target of a run

    model/{target}_config_{kit}_{pore}.yaml      ->  config.cnn.{n_layers, depth, channels, kernels, n_classes, classifier}
    model/{target}_model_{kit}_{pore}.pth        ->  Model(model_file, config, logger, target)

with pore = "R9.4.1" for kit RNA002 and "RP4" for RNA004 (`get_pore_version`, riser/riser.py:26-32).  The reference reads
the YAML through `attridict`; here the mapping becomes nested `types.SimpleNamespace` objects (attribute access is all
`Model` / `ConvNet` use, riser/nets/cnn.py:13-21).  PyYAML parses the file when it is installed; the configs the reference
ships are flat `key: value` lines under one or two section headers, which the fallback parser below reads without it.
"""
from __future__ import annotations

import os
import types

_PORE = {"RNA002": "R9.4.1", "RNA004": "RP4"}


def get_pore_version(kit: str) -> str:
    """riser/riser.py:26-32"""
    try:
        return _PORE[kit]
    except KeyError:
        raise Exception(f"Invalid kit {kit}") from None


def _scalar(text: str):
    t = text.strip()
    if t.startswith("[") and t.endswith("]"):
        return [_scalar(x) for x in t[1:-1].split(",") if x.strip()]
    if len(t) >= 2 and t[0] == t[-1] and t[0] in "'\"":
        return t[1:-1]
    for cast in (int, float):
        try:
            return cast(t)
        except ValueError:
            pass
    return {"true": True, "false": False, "null": None, "~": None}.get(t.lower(), t)


def _parse_flat_yaml(text: str) -> dict:
    """`key: value` lines, one level of indented sections, `#` comments, inline `[a, b]` lists: the subset of YAML the
    reference's config files use (riser/model/*.yaml)."""
    root, section = {}, None
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].rstrip()
        if not line.strip():
            continue
        key, sep, val = line.strip().partition(":")
        if not sep:
            raise ValueError(f"config line without `key:`: {raw!r}")
        indented = line[0] in " \t"
        if not indented:
            section = None
        if val.strip() == "":
            if indented:
                raise ValueError(f"nested sections deeper than one level are not supported: {raw!r}")
            section = root.setdefault(key.strip(), {})
            continue
        (section if indented and section is not None else root)[key.strip()] = _scalar(val)
    return root


def _namespace(obj):
    if isinstance(obj, dict):
        return types.SimpleNamespace(**{k: _namespace(v) for k, v in obj.items()})
    return obj


def get_config(filepath: str):
    """riser/riser.py:21-23: the YAML file as an object with attribute access (`config.cnn.channels` ...)."""
    with open(filepath) as f:
        text = f.read()
    try:
        import yaml
        data = yaml.safe_load(text)
    except ImportError:
        data = _parse_flat_yaml(text)
    if not isinstance(data, dict):
        raise ValueError(f"{filepath}: not a mapping")
    return _namespace(data)


def model_files(model_dir: str, target: str, kit: str):
    """(config.yaml, model.pth) of one target, named as riser/riser.py:38-40 names them."""
    pore = get_pore_version(kit)
    return (os.path.join(model_dir, f"{target}_config_{kit}_{pore}.yaml"),
            os.path.join(model_dir, f"{target}_model_{kit}_{pore}.pth"))


def get_models(targets, logger, kit: str, model_dir: str = "model", dtype: str = "f32w", device=None) -> list:
    """riser/riser.py:35-42 with the directory, the arithmetic mode and the device as arguments."""
    from .model import Model
    models = []
    for target in targets:
        cfg_file, model_file = model_files(model_dir, target, kit)
        for p in (cfg_file, model_file):
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p}: the reference's layout is model/{{target}}_config_{{kit}}_{{pore}}.yaml next to "
                                        f"model/{{target}}_model_{{kit}}_{{pore}}.pth (riser/riser.py:35-42)")
        models.append(Model(model_file, get_config(cfg_file), logger, target, dtype=dtype, device=device))
    return models
