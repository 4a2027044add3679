"""The drop-in boundary, pinned: every public function (and public method of a public class) under
vggsfm_amd/{models,utils,two_view_geo} that has a namesake in the same-named module of the reference must accept every call
the reference's own call sites can make -- the reference's parameter names, in the reference's order, with the reference's
defaults, as a PREFIX of ours; anything we add must be optional.  Compared on the syntax trees (nothing of the reference is
imported or executed).  Build container only: /root/reference does not exist on the GPU box."""
import ast
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/vggsfm"
PACKAGES = ("models", "utils", "two_view_geo")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")

# Documented, deliberate differences.  Key: "package/module.py:qualname".
ALLOWED = {}


def _functions(path):
    """qualname -> ast.FunctionDef for module-level functions and methods of module-level classes."""
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            out[node.name] = node
        elif isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    out[f"{node.name}.{sub.name}"] = sub
    return out


def _public(qualname):
    return all(not part.startswith("_") or part in ("__init__", "__call__") for part in qualname.split("."))


def _params(fn):
    """[(name, default-or-None as a normalised string)] for positional parameters, same for keyword-only ones."""
    a = fn.args
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    norm = lambda d: None if d is None else ast.dump(d)
    positional = [(p.arg, norm(d)) for p, d in zip(pos, defaults)]
    kwonly = {p.arg: norm(d) for p, d in zip(a.kwonlyargs, a.kw_defaults)}
    return positional, kwonly, a.vararg is not None, a.kwarg is not None


def _pairs():
    for pkg in PACKAGES:
        ours_dir = os.path.join(ROOT, "vggsfm_amd", pkg)
        for name in sorted(os.listdir(ours_dir)):
            if not name.endswith(".py") or name == "__init__.py":
                continue
            ref_path = os.path.join(REF, pkg, name)
            if os.path.isfile(ref_path):
                yield f"{pkg}/{name}", os.path.join(ours_dir, name), ref_path


def _compare(key, ours, ref):
    """List of human-readable incompatibilities of `ours` against `ref` (both ast.FunctionDef)."""
    (op, ok, ovar, okw), (rp, rk, rvar, rkw) = _params(ours), _params(ref)
    problems = []
    for i, (rname, rdef) in enumerate(rp):
        if i >= len(op):
            if not (okw and rdef is not None):
                problems.append(f"{key}: reference parameter #{i} '{rname}' is missing")
            continue
        oname, odef = op[i]
        if oname != rname:
            problems.append(f"{key}: parameter #{i} is '{oname}', the reference has '{rname}'")
        elif rdef is not None and odef is None:
            problems.append(f"{key}: '{rname}' is optional in the reference and required here")
        elif rdef is not None and odef != rdef:
            problems.append(f"{key}: default of '{rname}' differs from the reference's")
        elif rdef is None and odef is not None:
            pass  # more permissive than the reference: compatible
    for oname, odef in op[len(rp):]:
        if odef is None:
            problems.append(f"{key}: extra parameter '{oname}' has no default")
    for rname, rdef in rk.items():
        if rname not in ok and rname not in [n for n, _ in op] and not okw:
            problems.append(f"{key}: reference keyword-only parameter '{rname}' is missing")
    for oname, odef in ok.items():
        if odef is None and oname not in rk:
            problems.append(f"{key}: extra keyword-only parameter '{oname}' has no default")
    if rvar and not ovar:
        problems.append(f"{key}: the reference accepts *args")
    if rkw and not okw:
        problems.append(f"{key}: the reference accepts **kwargs")
    return problems


def test_every_namesake_accepts_the_reference_call_signature():
    problems, compared = [], 0
    for rel, ours_path, ref_path in _pairs():
        ours, ref = _functions(ours_path), _functions(ref_path)
        for q in sorted(set(ours) & set(ref)):
            if not _public(q):
                continue
            key = f"{rel}:{q}"
            compared += 1
            found = _compare(key, ours[q], ref[q])
            if key in ALLOWED:
                assert found, f"{key} is listed in ALLOWED ({ALLOWED[key]}) but no longer differs: remove the entry"
                continue
            problems += found
    assert compared >= 30, f"only {compared} namesake functions found: did a package move?"
    assert not problems, "boundary drift against /root/reference:\n  " + "\n  ".join(problems)


def test_reference_call_sites_keywords_exist():
    """Every keyword a reference call site passes to one of the namesake functions (by bare or attribute name) exists here.
    Catches a dropped keyword even when the positional prefix still matches."""
    ours_by_name = {}
    for rel, ours_path, ref_path in _pairs():
        for q, fn in _functions(ours_path).items():
            if "." not in q and _public(q) and q in _functions(ref_path):
                ours_by_name.setdefault(q, []).append((rel, fn))
    missing = []
    for dirpath, _, files in os.walk(REF):
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dirpath, f)
            try:
                tree = ast.parse(open(path).read())
            except SyntaxError:
                continue
            for node in ast.walk(tree):
                if not isinstance(node, ast.Call):
                    continue
                name = node.func.id if isinstance(node.func, ast.Name) else (
                    node.func.attr if isinstance(node.func, ast.Attribute) else None)
                if name not in ours_by_name:
                    continue
                for rel, fn in ours_by_name[name]:
                    pos, kwonly, _, has_kw = _params(fn)
                    names = {n for n, _ in pos} | set(kwonly)
                    for kw in node.keywords:
                        if kw.arg is not None and kw.arg not in names and not has_kw:
                            missing.append(f"{os.path.relpath(path, REF)}:{node.lineno} passes {name}({kw.arg}=...) "
                                           f"which {rel} does not accept")
    assert not missing, "\n".join(sorted(set(missing)))
