"""A small stand-in for the parts of Hydra the reference's entry points use (hydra/omegaconf are not
installed in this image): defaults-list composition, ``a.b=c`` overrides, ``${a.b}`` interpolation and
``_target_`` instantiation (reference: src/train.py:17, src/tasks/train_task.py:34-47,
src/models/networks/discrete_diffusion.py:11-12).  If the real hydra is importable the entry points use it."""
import copy
import importlib
import os
import re
import time

import yaml


class Cfg(dict):
    """dict with attribute access (enough of DictConfig for the task code)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")


def _wrap(x):
    if isinstance(x, str) and _FLOAT.match(x):        # PyYAML (YAML 1.1) reads 4e-4 as a string; OmegaConf does not
        return float(x)
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg, dotted, value):
    keys = dotted.split(".")
    cur = cfg
    for k in keys[:-1]:
        cur = cur.setdefault(k, {})
    cur[keys[-1]] = value


def _load(config_dir, rel):
    path = os.path.join(config_dir, rel if rel.endswith((".yaml", ".yml")) else rel + ".yaml")
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _compose_file(config_dir, rel, group_dir, group_choice):
    """Load one config file and expand its defaults list.  Returns a plain dict."""
    raw = _load(config_dir, rel)
    defaults = raw.pop("defaults", [])
    out, own = {}, raw
    self_done = False
    for item in defaults:
        if item == "_self_":
            _merge(out, own)
            self_done = True
            continue
        if isinstance(item, str) and (item.startswith("/") or "@" in item):
            item = {item: None}                       # `/group/file@package`: the path is the file
        if isinstance(item, str):                     # a bare file name: a sibling in this file's group, merged in place
            try:                                      # (configs/callbacks/default.yaml: `- model_checkpoint.yaml`)
                _merge(out, _compose_file(config_dir, os.path.join(group_dir, item), group_dir, {}))
            except FileNotFoundError:
                if not item.startswith("optional "):
                    raise
            continue
        (key, choice), = item.items()
        optional = key.startswith("optional ")
        key = key.replace("optional ", "").strip()
        pkg = None
        if "@" in key:
            key, pkg = key.split("@", 1)
        if key.startswith("/"):                       # absolute group path, value None -> the path is the file
            rel2 = key[1:] if choice is None else os.path.join(key[1:], str(choice))
            if pkg is None:
                pkg = os.path.basename(key)
        else:
            choice = group_choice.get(key, choice)
            if choice is None:
                continue
            rel2 = os.path.join(group_dir, key, str(choice))
            if pkg is None:
                pkg = key
        try:
            sub = _compose_file(config_dir, rel2, os.path.dirname(rel2), {})
        except FileNotFoundError:
            if optional:
                continue
            raise
        tgt = out
        for part in pkg.split("."):
            tgt = tgt.setdefault(part, {})
        _merge(tgt, sub)
    if not self_done:
        _merge(out, own)
    return out


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _resolve(root, node, where=()):
    """``where`` is the key path of the container that holds ``node`` (for ``${.sibling}`` / ``${..uncle}``)."""
    if isinstance(node, dict):
        for k in list(node):
            if not where and k == "hydra":            # hydra's own node (run/sweep dir patterns with ${hydra.job.*}): resolved only
                continue                              # where something points into it, and dropped from the result like hydra does
            node[k] = _resolve(root, node[k], where + (k,))
        return node
    if isinstance(node, list):
        return [_resolve(root, v, where) for v in node]
    if isinstance(node, str):
        holder = where[:-1]                            # a leaf's own key is the last element
        m = _INTERP.fullmatch(node)
        if m:
            tgt, at = _lookup(root, m.group(1), holder)
            return _resolve(root, tgt, at)
        return _INTERP.sub(lambda mm: str(_resolve(root, *_lookup(root, mm.group(1), holder))), node)
    return node


_NOW = [None]                                          # one clock reading per compose(), like hydra's ${now:...}


def _lookup(root, dotted, holder=()):
    """Value of an interpolation key and the key path it was found at.  Resolvers: oc.env, now, hydra:runtime.{cwd,output_dir}
    (the ones the reference's configs/paths/default.yaml and configs/hydra/default.yaml use)."""
    if dotted.startswith("oc.env:"):
        name, _, default = dotted[7:].partition(",")
        if name not in os.environ and not _:
            raise KeyError(f"environment variable '{name}' is not set (configs/paths/default.yaml needs PROJECT_ROOT; "
                           "src/train.py and src/eval.py set it)")
        return os.environ.get(name, default), ()
    if dotted.startswith("now:"):
        return time.strftime(dotted[4:], _NOW[0] or time.localtime()), ()
    if dotted.startswith("hydra:"):
        what = dotted[6:]
        if what == "runtime.cwd":
            return os.getcwd(), ()
        if what == "runtime.output_dir":
            return "${hydra.run.dir}", ("hydra", "run")
        raise KeyError(f"${{{dotted}}}: only hydra:runtime.cwd and hydra:runtime.output_dir are provided")
    base = ()
    if dotted.startswith("."):
        ups = len(dotted) - len(dotted.lstrip("."))
        dotted = dotted[ups:]
        base = holder[:len(holder) - (ups - 1)] if ups > 1 else holder
    path = base + tuple(dotted.split("."))
    cur = root
    for k in path:
        cur = cur[k]
    return cur, path


def compose(config_dir, config_name, overrides=()):
    groups = {d for d in os.listdir(config_dir) if os.path.isdir(os.path.join(config_dir, d))}
    group_choice, sets = {}, []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if k in groups and "." not in k:
            group_choice[k] = None if v in ("null", "None") else v
        else:
            sets.append((k, yaml.safe_load(v)))
    cfg = _compose_file(config_dir, config_name, "", group_choice)
    for k, v in sets:
        _set_path(cfg, k, v)
    _NOW[0] = time.localtime()
    cfg = _resolve(cfg, cfg, ())
    cfg.pop("hydra", None)
    return _wrap(cfg)


def get_class(path):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, *args, _recursive_=True, **kwargs):
    """``_target_`` instantiation with hydra's _recursive_ flag."""
    if cfg is None:
        return None
    if not isinstance(cfg, dict) or "_target_" not in cfg:
        return cfg
    params = {k: v for k, v in cfg.items() if k not in ("_target_", "_recursive_", "_partial_")}
    rec = cfg.get("_recursive_", _recursive_)
    if rec:
        params = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in params.items()}
    params.update(kwargs)
    return get_class(cfg["_target_"])(*args, **params)
