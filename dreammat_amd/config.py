"""Config layer: the YAML + dotlist surface of threestudio/utils/config.py:11-123 without omegaconf,
and the step-schedule evaluator C() of threestudio/utils/misc.py:65-86."""
import copy
import dataclasses
import os
import re
import typing
from dataclasses import dataclass, field
from typing import Any, Optional

import yaml


# ---------------------------------------------------------------- resolvers (config.py:11-28)
def _resolve(expr, root):
    m = re.fullmatch(r"\$\{(.*)\}", expr.strip())
    if not m:
        def sub(mm):
            v = _resolve(mm.group(0), root)
            return str(v)
        return re.sub(r"\$\{[^{}]*(?:\{[^{}]*\}[^{}]*)*\}", sub, expr)
    body = m.group(1)
    if ":" in body and body.split(":", 1)[0] in _RESOLVERS:
        name, args = body.split(":", 1)
        parts = _split_args(args)
        vals = [_resolve(a, root) if "${" in a else _literal(a) for a in parts]
        return _RESOLVERS[name](*vals)
    node = root
    for k in body.split("."):
        node = node[k]
    if isinstance(node, str) and "${" in node:
        node = _resolve(node, root)
    return node


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def _literal(s):
    try:
        return yaml.safe_load(s)
    except Exception:
        return s


_RESOLVERS = {
    "calc_exp_lr_decay_rate": lambda factor, n: factor ** (1.0 / n),
    "add": lambda a, b: a + b,
    "sub": lambda a, b: a - b,
    "mul": lambda a, b: a * b,
    "div": lambda a, b: a / b,
    "idiv": lambda a, b: a // b,
    "basename": lambda p: os.path.basename(p),
    "rmspace": lambda s, sub: str(s).replace(" ", str(sub)),
    "tuple2": lambda s: [float(s), float(s)],
    "gt0": lambda s: s > 0,
    "not": lambda s: not s,
}


def _resolve_tree(node, root):
    if isinstance(node, dict):
        return {k: _resolve_tree(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve_tree(v, root) for v in node]
    if isinstance(node, str) and "${" in node:
        return _resolve(node, root)
    return node


def _set_dotted(d, key, value):
    ks = key.split(".")
    for k in ks[:-1]:
        d = d.setdefault(k, {})
    d[ks[-1]] = value


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def load_config(*yamls, cli_args=(), **kwargs):
    """config.py:99-109: merge YAML files + `key=value` dotlist + kwargs, then resolve ${...}."""
    cfg = {}
    for y in yamls:
        if isinstance(y, str):
            with open(y) as fh:
                y = yaml.safe_load(fh)
        _merge(cfg, copy.deepcopy(y))
    for arg in cli_args:
        k, v = arg.split("=", 1)
        _set_dotted(cfg, k, _literal(v))
    _merge(cfg, kwargs)
    cfg = _resolve_tree(cfg, cfg)
    missing = [k for k in _find_missing(cfg)]
    cfg["_missing"] = missing
    return cfg


def _find_missing(node, prefix=""):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _find_missing(v, f"{prefix}{k}.")
    elif node == "???":
        yield prefix[:-1]


def parse_structured(fields_cls, cfg=None):
    """config.py:121-123: validate a dict against a dataclass `Config` (unknown keys are errors,
    `???` mandatory keys must be filled)."""
    cfg = dict(cfg or {})
    names = {f.name: f for f in dataclasses.fields(fields_cls)}
    for k in cfg:
        if k not in names:
            raise KeyError(f"Key '{k}' not in '{fields_cls.__qualname__}'")
    for k, v in cfg.items():
        if v == "???":
            raise ValueError(f"Missing mandatory value: {fields_cls.__qualname__}.{k}")
    return fields_cls(**copy.deepcopy(cfg))


def config_to_primitive(v):
    return v


def C(value: Any, epoch: int, global_step: int) -> float:
    """misc.py:65-86: scalar or [start_step, start_value, end_value, end_step] linear schedule;
    an int end_step counts optimizer steps, a float end_step counts epochs."""
    if isinstance(value, (int, float)):
        return value
    if not isinstance(value, (list, tuple)):
        raise TypeError("Scalar specification only supports list, got", type(value))
    value = list(value)
    if len(value) == 3:
        value = [0] + value
    assert len(value) == 4
    start_step, start_value, end_value, end_step = value
    current = global_step if isinstance(end_step, int) else epoch
    return start_value + (end_value - start_value) * max(min(1.0, (current - start_step) / (end_step - start_step)), 0.0)
