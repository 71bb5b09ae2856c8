"""tools: poor man's pyflakes (none in the image) — names loaded in a module that nothing binds.  usage: check_names.py file.py ..."""
import ast, builtins, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bad = 0
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {"__file__", "__name__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            bound.add(n.name)
            if not isinstance(n, ast.ClassDef):
                a = n.args
                for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                    bound.add(x.arg)
        elif isinstance(n, ast.Lambda):
            for x in n.args.args:
                bound.add(x.arg)
        elif isinstance(n, ast.Import):
            for a in n.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ImportFrom):
            for a in n.names:
                if a.name == "*":
                    pkg = os.path.relpath(os.path.dirname(os.path.abspath(path)), ROOT).replace(os.sep, ".")
                    m = importlib.import_module("." * n.level + (n.module or ""), pkg if n.level else None)
                    bound |= {k for k in dir(m) if not k.startswith("_")}
                else:
                    bound.add(a.asname or a.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            print(f"{path}:{n.lineno}: undefined name {n.id}")
            bad += 1
sys.exit(1 if bad else 0)
