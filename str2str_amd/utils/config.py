"""A small stand-in for the slice of Hydra/OmegaConf that ``eval.py`` uses, so the reference's YAML
files and CLI (`key=value` overrides, `defaults:` lists, `${oc.env:VAR}`, `${a.b}` interpolation,
`_target_` / `_partial_` instantiation) work where hydra is not installed.  When hydra IS
installed eval.py uses it instead."""
from __future__ import annotations

import datetime
import functools
import importlib
import os
import re
from typing import Any, List

import yaml


class Cfg(dict):
    """dict with attribute access (enough of DictConfig for this code base)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_plain(x):
    if isinstance(x, dict):
        return {k: to_plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_plain(v) for v in x]
    return x


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_group(cfg_dir: str, group: str, name: str) -> dict:
    with open(os.path.join(cfg_dir, group, f"{name}.yaml")) as f:
        data = yaml.safe_load(f) or {}
    out: dict = {}
    for d in data.pop("defaults", []) or []:
        if isinstance(d, str) and d != "_self_":
            _merge(out, _load_group(cfg_dir, group, d))
    return _merge(out, data)


def _set(cfg: dict, dotted: str, value):
    cur = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = value


def _get(cfg: dict, dotted: str):
    cur: Any = cfg
    for p in dotted.split("."):
        cur = cur[p]
    return cur


_INTERP = re.compile(r"\$\{([^${}]+)\}")
_SCI_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")  # PyYAML reads 1e-8 as a string; OmegaConf as a float


def _resolve_str(s: str, root: dict, now: datetime.datetime, runtime: dict):
    def one(expr: str):
        if expr.startswith("oc.env:"):
            var, _, default = expr[len("oc.env:"):].partition(",")
            if var in os.environ:
                return os.environ[var]
            if default != "" or "," in expr:
                return default
            raise KeyError(f"environment variable '{var}' not set (needed by the config, see .env.example)")
        if expr.startswith("now:"):
            return now.strftime(expr[4:])
        if expr.startswith("hydra:"):
            return runtime[expr[6:]]
        return _get(root, expr)

    for _ in range(20):
        m = _INTERP.fullmatch(s)
        if m:  # whole-string interpolation keeps the type
            v = one(m.group(1))
            if isinstance(v, str) and "${" in v:
                s = v
                continue
            return v
        if not _INTERP.search(s):
            return s
        s = _INTERP.sub(lambda mm: str(one(mm.group(1))), s)
    raise ValueError(f"unresolvable interpolation: {s}")


def _resolve(node, root, now, runtime):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(node[k], root, now, runtime)
        return node
    if isinstance(node, list):
        return [_resolve(v, root, now, runtime) for v in node]
    if isinstance(node, str) and "${" in node:
        return _resolve(_resolve_str(node, root, now, runtime), root, now, runtime)
    if isinstance(node, str) and _SCI_FLOAT.match(node):
        return float(node)
    if node == "null":
        return None
    return node


def compose(cfg_dir: str, config_name: str, overrides: List[str]) -> Cfg:
    with open(os.path.join(cfg_dir, config_name)) as f:
        primary = yaml.safe_load(f) or {}
    defaults = primary.pop("defaults", [])
    groups = {}
    for d in defaults:
        if isinstance(d, dict):
            (g, n), = d.items()
            groups[g] = n
    plain = []
    for ov in overrides:
        key, _, val = ov.lstrip("+").partition("=")
        if key in groups and "." not in key:
            groups[key] = None if val in ("null", "None", "") else val
        else:
            plain.append((key, yaml.safe_load(val) if val != "" else None))
    cfg: dict = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(cfg, primary)
            self_done = True
        elif isinstance(d, dict):
            (g, _), = d.items()
            if groups[g] is not None and g != "hydra":
                cfg[g] = _merge(cfg.get(g, {}), _load_group(cfg_dir, g, groups[g]))
            elif groups[g] is None:
                cfg.setdefault(g, None)
    if not self_done:
        _merge(cfg, primary)
    for k, v in plain:
        _set(cfg, k, v)
    now = datetime.datetime.now()
    root_dir = os.environ.get("PROJECT_ROOT", os.getcwd())
    os.environ.setdefault("PROJECT_ROOT", root_dir)
    log_dir = os.path.join(root_dir, "logs")
    out_dir = os.path.join(log_dir, str(cfg.get("task_name", "eval")), "runs", now.strftime("%Y-%m-%d_%H-%M-%S"))
    runtime = {"runtime.output_dir": out_dir, "runtime.cwd": os.getcwd()}
    cfg = _resolve(cfg, cfg, now, runtime)
    os.makedirs(out_dir, exist_ok=True)
    return _wrap(cfg)


def locate(path: str):
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def instantiate(node, **extra):
    """Recursive `_target_` instantiation (`_partial_: true` -> functools.partial)."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return Cfg({k: instantiate(v) for k, v in node.items()})
    kwargs = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
    kwargs.update(extra)
    fn = locate(node["_target_"])
    return functools.partial(fn, **kwargs) if node.get("_partial_") else fn(**kwargs)
