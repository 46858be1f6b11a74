#!/usr/bin/env python3
"""Undefined-name check for the package's modules (no pyflakes in this image): every global name a function loads must be defined at
module level, imported, or a builtin.  usage: python tools/check_names.py [files...]  (exit 1 on findings)"""
import ast
import builtins
import glob
import os
import sys


def names_of(n):
    if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
        return [n.name]
    if isinstance(n, ast.Assign):
        out = []
        for t in n.targets:
            for x in ast.walk(t):
                if isinstance(x, ast.Name):
                    out.append(x.id)
        return out
    if isinstance(n, (ast.AnnAssign, ast.AugAssign)) and isinstance(n.target, ast.Name):
        return [n.target.id]
    if isinstance(n, (ast.Import, ast.ImportFrom)):
        return [(a.asname or a.name).split(".")[0] for a in n.names]
    return []


def module_globals(tree):
    g = set()
    for n in ast.walk(tree):          # conservative: any top-level-or-nested-in-if/try assignment at module scope
        pass
    def walk(body):
        for n in body:
            g.update(names_of(n))
            for attr in ("body", "orelse", "finalbody", "handlers"):
                sub = getattr(n, attr, None)
                if isinstance(n, (ast.If, ast.Try, ast.With, ast.For, ast.While)) and sub:
                    walk([x for x in sub if isinstance(x, ast.stmt)] + [s for h in sub if isinstance(h, ast.ExceptHandler) for s in h.body])
            if isinstance(n, ast.For):
                for x in ast.walk(n.target):
                    if isinstance(x, ast.Name):
                        g.add(x.id)
    walk(tree.body)
    return g


class Check(ast.NodeVisitor):
    def __init__(self, g):
        self.g, self.stack, self.bad = g, [], []

    def _locals(self, fn):
        loc = set()
        a = fn.args
        for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
            loc.add(arg.arg)

        def walk(node):
            for ch in ast.iter_child_nodes(node):
                if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                    loc.add(ch.name)
                    continue
                if isinstance(ch, ast.Lambda):
                    continue
                if isinstance(ch, ast.Name) and isinstance(ch.ctx, (ast.Store, ast.Del)):
                    loc.add(ch.id)
                if isinstance(ch, (ast.Import, ast.ImportFrom)):
                    for al in ch.names:
                        loc.add((al.asname or al.name).split(".")[0])
                if isinstance(ch, ast.ExceptHandler) and ch.name:
                    loc.add(ch.name)
                if isinstance(ch, (ast.Global, ast.Nonlocal)):
                    continue
                walk(ch)
        body = fn.body if isinstance(fn.body, list) else [fn.body]
        for st in body:
            if isinstance(st, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                loc.add(st.name)
            else:
                if isinstance(st, ast.Name) and isinstance(st.ctx, ast.Store):
                    loc.add(st.id)
                walk(st)
                if isinstance(st, (ast.Import, ast.ImportFrom)):
                    for al in st.names:
                        loc.add((al.asname or al.name).split(".")[0])
        return loc

    def visit_FunctionDef(self, n):
        for d in n.decorator_list + n.args.defaults + [x for x in n.args.kw_defaults if x is not None]:
            self.visit(d)
        self.stack.append(self._locals(n))
        for st in n.body:
            self.visit(st)
        self.stack.pop()

    visit_AsyncFunctionDef = visit_FunctionDef

    def visit_Lambda(self, n):
        self.stack.append(self._locals(n))
        self.visit(n.body)
        self.stack.pop()

    def visit_ClassDef(self, n):
        for d in n.decorator_list + n.bases:
            self.visit(d)
        self.stack.append({x for st in n.body for x in names_of(st)})
        for st in n.body:
            self.visit(st)
        self.stack.pop()

    def _comp(self, n):
        loc = set()
        for gen in n.generators:
            for t in ast.walk(gen.target):
                if isinstance(t, ast.Name):
                    loc.add(t.id)
        self.stack.append(loc)
        self.generic_visit(n)
        self.stack.pop()

    visit_ListComp = visit_SetComp = visit_DictComp = visit_GeneratorExp = _comp

    def visit_arg(self, n):
        return

    def visit_AnnAssign(self, n):
        self.visit(n.target)
        if n.value:
            self.visit(n.value)

    def visit_Name(self, n):
        if isinstance(n.ctx, ast.Load) and not any(n.id in s for s in self.stack) and n.id not in self.g and n.id not in dir(builtins) and n.id not in ("__file__", "__name__"):
            self.bad.append((n.lineno, n.id))


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "tss-with-dprnn_amd", "*.py"))) + [os.path.join(root, "bench.py")]
    rc = 0
    for f in files:
        tree = ast.parse(open(f).read())
        c = Check(module_globals(tree))
        c.visit(tree)
        for ln, name in sorted(set(c.bad)):
            print(f"{os.path.relpath(f, root)}:{ln}: undefined name {name}")
            rc = 1
    # attributes read through the engine module object must exist there
    import importlib
    sys.path.insert(0, root)
    import tss_with_dprnn_amd  # noqa: F401
    eng = importlib.import_module("tss_with_dprnn_amd.engine")
    for f in files:
        tree = ast.parse(open(f).read())
        for n in ast.walk(tree):
            if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "E" and not hasattr(eng, n.attr):
                print(f"{os.path.relpath(f, root)}:{n.lineno}: engine has no attribute {n.attr}")
                rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
