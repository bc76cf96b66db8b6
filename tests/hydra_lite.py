"""Just enough of Hydra / OmegaConf to read the reference's model configs in the tests (neither
library is installed here): composition of the ``defaults`` lists with ``# @package`` headers,
deep merge, and LAZY resolution of ``${a.b.c}`` and ``${eval:'<python>'}`` interpolations - only
the nodes a test asks for are resolved, so unrelated entries (paths, loggers, trainer) may stay
unresolvable.  TEST INFRASTRUCTURE ONLY."""
import os
import re

import yaml

ROOT = "/root/reference/configs"


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load(rel, cfg, seen=None):
    """Merge ``rel`` (path under configs/, with or without .yaml) into ``cfg`` after its defaults;
    a file listed by several defaults lists is merged once (Hydra's rule)."""
    seen = cfg.setdefault("__seen__", set()) if seen is None else seen
    rel = rel.lstrip("/")
    if not rel.endswith(".yaml"):
        rel += ".yaml"
    if rel in seen:
        return cfg
    seen.add(rel)
    text = open(os.path.join(ROOT, rel)).read()
    m = re.match(r"#\s*@package\s+(\S+)", text)
    package = m.group(1) if m else rel.split("/")[0]
    body = yaml.safe_load(text) or {}
    for d in body.pop("defaults", []) or []:
        if isinstance(d, str) and d != "_self_":
            _load(d if d.startswith("/") else os.path.join(os.path.dirname(rel), d), cfg, seen)
    node = cfg
    if package != "_global_":
        for part in package.split("."):
            node = node.setdefault(part, {})
    _merge(node, body)
    return cfg


def compose(datamodule, model, overrides=None):
    """``configs/datamodule/<datamodule>.yaml`` + ``configs/model/<model>.yaml`` (+ their defaults),
    then the experiment's plain overrides (a nested dict)."""
    cfg = {}
    _load("datamodule/" + datamodule, cfg)
    _load("model/" + model, cfg)
    cfg.pop("__seen__", None)
    _merge(cfg, overrides or {})
    return Config(cfg)


class Config:
    def __init__(self, root):
        self.root = root
        self._busy = set()

    def raw(self, path):
        node = self.root
        for part in path.split("."):
            node = node[int(part)] if isinstance(node, list) else node[part]
        return node

    def get(self, path):
        if path in self._busy:
            raise RecursionError(f"interpolation cycle at {path}")
        self._busy.add(path)
        try:
            return self.resolve(self.raw(path))
        finally:
            self._busy.discard(path)

    def resolve(self, v):
        if isinstance(v, dict):
            return {k: self.resolve(x) for k, x in v.items()}
        if isinstance(v, list):
            return [self.resolve(x) for x in v]
        if not isinstance(v, str) or "${" not in v:
            return v
        return self._resolve_str(v, v.lstrip().startswith("${eval:"))

    @staticmethod
    def _matching(s, start):
        """Index of the '}' closing the '${' at ``start`` (dict literals inside nest properly)."""
        depth, i = 0, start
        while i < len(s):
            if s[i] == "{":
                depth += 1
            elif s[i] == "}":
                depth -= 1
                if depth == 0:
                    return i
            i += 1
        raise ValueError(f"unbalanced interpolation in {s[:80]!r}")

    def _resolve_str(self, v, in_eval):
        while True:
            a = v.find("${")
            if a < 0:
                return v
            b = self._matching(v, a + 1)
            body = v[a + 2:b].strip()
            if body.startswith("eval:"):
                expr = body[5:].strip()
                if expr[:1] in "'\"" and expr[-1:] == expr[:1]:
                    expr = expr[1:-1]
                expr = self._resolve_str(expr, True)             # inner interpolations first
                val = eval(expr, {"__builtins__": __builtins__}, {})     # the configs' own python
            else:
                val = self.get(body)
            if a == 0 and b == len(v) - 1:
                return val
            v = v[:a] + (repr(val) if in_eval else str(val)) + v[b + 1:]
