"""TEST INFRASTRUCTURE (fixture generation only, runs in the build container where /root/reference exists).

Imports single modules of the reference tree (/root/reference/newton/...) WITHOUT executing the package __init__ files (which
pull in the whole simulator and its un-vendored dependencies): every package on the way is replaced by a bare module whose
attribute lookups are resolved lazily -- the package's __init__.py is parsed (not run) for `from .sub import name` lines, and
only the defining sub-module is imported.  Together with tests/golden/refshim/warp (a pure-Python stand-in for the warp-lang
API the kernels use) this lets the generator scripts EXECUTE the reference's own kernel source on small cases and record
golden input / output vectors.  Nothing here is shipped or used at test time."""
import ast
import importlib
import importlib.abc
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference"


class _LazyPackage(types.ModuleType):
    def __init__(self, name, path, dummies=()):
        super().__init__(name)
        self.__path__ = [path]
        self.__package__ = name
        self.__file__ = os.path.join(path, "__init__.py")
        self._dummies = set(dummies)
        self._where = None

    def _index(self):
        if self._where is None:
            self._where = {}
            init = os.path.join(self.__path__[0], "__init__.py")
            if os.path.exists(init):
                tree = ast.parse(open(init).read())
                for node in ast.walk(tree):
                    if isinstance(node, ast.ImportFrom) and node.level >= 1 and node.module:
                        for a in node.names:
                            self._where[a.asname or a.name] = (node.level, node.module, a.name)
        return self._where

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = os.path.join(self.__path__[0], name)
        if os.path.isdir(sub) or os.path.exists(sub + ".py"):
            return importlib.import_module(self.__name__ + "." + name)
        if name in self._dummies:
            return _Dummy(name)
        where = self._index().get(name)
        if where is None:
            raise AttributeError(f"{self.__name__}: no lazy source for {name!r}")
        level, module, attr = where
        base = self.__name__.split(".")
        base = base[: len(base) - (level - 1)]
        mod = importlib.import_module(".".join(base + [module]))
        value = getattr(mod, attr)
        setattr(self, name, value)
        return value


class _Dummy:
    """Stands for a reference class that is only used in annotations / isinstance-free code paths."""

    def __init__(self, name="dummy"):
        self._name = name

    def __call__(self, *a, **k):
        return _Dummy(self._name + "()")

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy(self._name + "." + k)

    def __getitem__(self, k):
        return _Dummy(self._name + "[]")

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self

    def __mro_entries__(self, bases):
        return (object,)


class _Finder:
    """Turns every reference PACKAGE into a _LazyPackage; plain modules are imported normally from the reference tree."""

    def __init__(self, dummies, execute, dummy_modules=()):
        self.dummies, self.execute, self.dummy_modules = dummies, set(execute), set(dummy_modules)

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "newton" and not fullname.startswith("newton."):
            return None
        if fullname in self.dummy_modules:  # modules that only contribute types to annotations on the paths executed here
            return importlib.util.spec_from_loader(fullname, _DummyModuleLoader())
        rel = fullname.split(".")
        p = os.path.join(REF_ROOT, *rel)
        if os.path.isdir(p) and fullname in self.execute:  # packages whose __init__ holds real definitions: run it
            init = os.path.join(p, "__init__.py")
            return importlib.util.spec_from_file_location(fullname, init, loader=_ValueSemanticsLoader(fullname, init),
                                                          submodule_search_locations=[p])
        if os.path.isdir(p):
            return importlib.util.spec_from_loader(fullname, _PkgLoader(p, self.dummies.get(fullname, ())), is_package=True)
        if os.path.exists(p + ".py"):
            return importlib.util.spec_from_file_location(fullname, p + ".py", loader=_ValueSemanticsLoader(fullname, p + ".py"))
        return None


class _CopyOnAssign(ast.NodeTransformer):
    """Warp's vector / matrix / quaternion / transform types are VALUE types: `b = a` copies, `b[0] = x` leaves `a` alone.
    Python names alias.  Every assignment whose right-hand side is a plain name / attribute / element read is rewritten to
    `b = __wp_val__(a)` (copies shim value types, passes everything else -- arrays, ints, floats -- through)."""

    def _wrap(self, node):
        if isinstance(node.value, (ast.Name, ast.Attribute, ast.Subscript)):
            node.value = ast.copy_location(ast.Call(func=ast.Name(id="__wp_val__", ctx=ast.Load()), args=[node.value], keywords=[]),
                                           node.value)
        return node

    def visit_Assign(self, node):
        self.generic_visit(node)
        return self._wrap(node)

    def visit_AnnAssign(self, node):
        self.generic_visit(node)
        return self._wrap(node) if node.value is not None else node


class _F32Literals(ast.NodeTransformer):
    """Warp compiles every float literal and every `float(x)` of kernel code to float32.  Python keeps them as 64-bit floats, and
    a chain of pure-literal locals (`golden = 0.38...; offset = 0.5 * golden; left = 0.5 - offset; m = 0.5 * (a + b)` in
    do_edge_sdf_collision) would silently run in double precision.  Opt-in per module (install(f32_literals=...)): literals become
    `__wp_f32__(c)` and `float(x)` calls become `__wp_f32__(x)`; the bare name `float` (annotations, dtype arguments) is left alone."""

    def visit_Constant(self, node):
        if isinstance(node.value, float):
            return ast.copy_location(ast.Call(func=ast.Name(id="__wp_f32__", ctx=ast.Load()), args=[node], keywords=[]), node)
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        if isinstance(node.func, ast.Name) and node.func.id == "float" and len(node.args) == 1 and not node.keywords:
            node.func = ast.copy_location(ast.Name(id="__wp_f32__", ctx=ast.Load()), node.func)
        return node


_F32_MODULES = set()


class _ValueSemanticsLoader(importlib.abc.Loader):
    def __init__(self, name, path):
        self.name, self.path = name, path

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        src = open(self.path).read()
        tree = ast.parse(src, self.path)
        if self.name in _F32_MODULES:
            tree = _F32Literals().visit(tree)
        tree = ast.fix_missing_locations(_CopyOnAssign().visit(tree))
        module.__file__ = self.path
        exec(compile(tree, self.path, "exec"), module.__dict__)


class _DummyModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy(self.__name__ + "." + k)


class _DummyModuleLoader:
    def create_module(self, spec):
        return _DummyModule(spec.name)

    def exec_module(self, module):
        pass


class _PkgLoader:
    def __init__(self, path, dummies):
        self.path, self.dummies = path, dummies

    def create_module(self, spec):
        return _LazyPackage(spec.name, self.path, self.dummies)

    def exec_module(self, module):
        pass


def install(dummies=None, execute=("newton._src.math",), dummy_modules=(), f32_literals=()):
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)  # makes `import warp` find tests/golden/refshim/warp
    import builtins

    import warp

    builtins.__wp_val__ = warp._val
    builtins.__wp_f32__ = warp.f32
    _F32_MODULES.update(f32_literals)
    sys.meta_path.insert(0, _Finder(dummies or {}, execute, dummy_modules))
