"""Minimal `toml` stand-in (the real package is not installed here): load via tomli, dump via a tiny writer.
Only so the UNMODIFIED reference CLI (speech_enhance/tools/inference.py:5,30; base_inferencer.py:58-60) runs."""
import tomli


def load(f):
    if isinstance(f, (str, bytes)):
        with open(f, "rb") as fh:
            return tomli.load(fh)
    data = f.read()
    return tomli.loads(data if isinstance(data, str) else data.decode())


def loads(s):
    return tomli.loads(s)


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return repr(v)
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_fmt(x) for x in v) + "]"
    raise TypeError(type(v))


def dumps(d, _prefix=""):
    lines, tables = [], []
    for k, v in d.items():
        if isinstance(v, dict):
            tables.append((k, v))
        else:
            lines.append(f"{k} = {_fmt(v)}")
    out = "\n".join(lines) + ("\n" if lines else "")
    for k, v in tables:
        name = f"{_prefix}{k}"
        out += f"\n[{name}]\n" + dumps(v, name + ".")
    return out


def dump(d, f):
    f.write(dumps(d))
