"""Undefined-global check for the python sources (no pyflakes in the image): every LOAD_GLOBAL / LOAD_NAME of every code object of a
file must name a module-level binding of that file or a builtin.  `python tools/check_names.py [files]` (default: the package, bench.py,
__graft_entry__.py, oracle/, tests/, tools/); exit code 1 when something is unbound."""
import ast
import builtins
import dis
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def module_bindings(tree):
    names = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(node.name)
        elif isinstance(node, ast.Import):
            names.update((a.asname or a.name).split('.')[0] for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            names.update(a.asname or a.name for a in node.names)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            names.add(node.id)
        elif isinstance(node, ast.Global):
            names.update(node.names)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            names.add(node.name)
    return names


def code_objects(co):
    yield co
    for c in co.co_consts:
        if hasattr(c, 'co_code'):
            yield from code_objects(c)


def check(path):
    src = open(path).read()
    tree = ast.parse(src, path)
    bound = module_bindings(tree) | set(dir(builtins)) | {'__file__', '__name__', '__doc__', '__builtins__', '__spec__', '__package__'}
    bad = []
    for co in code_objects(compile(src, path, 'exec')):
        for ins in dis.get_instructions(co):
            if ins.opname in ('LOAD_GLOBAL', 'LOAD_NAME') and ins.argval not in bound:
                bad.append((ins.positions.lineno if hasattr(ins, 'positions') and ins.positions else co.co_firstlineno, co.co_name, ins.argval))
    return sorted(set(bad))


if __name__ == '__main__':
    files = sys.argv[1:]
    if not files:
        for pat in ('feabas_amd/*.py', 'bench.py', '__graft_entry__.py', 'oracle/*.py', 'tests/*.py', 'tools/*.py', 'tests/golden/*.py'):
            files += sorted(glob.glob(os.path.join(ROOT, pat)))
    total = 0
    for f in files:
        for line, fn, name in check(f):
            print(f'{os.path.relpath(f, ROOT)}:{line}: in {fn}: name {name!r} is not bound in the module')
            total += 1
    print(f'{len(files)} files, {total} unbound names')
    sys.exit(1 if total else 0)
