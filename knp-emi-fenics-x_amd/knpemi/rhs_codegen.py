"""From a membrane module's Python right-hand side to device source.

The reference binds a membrane model through `ode.rhs_numba.address` (`src/knpemi/odeSolver.py:96`): a numba
`cfunc(lsoda_sig)` whose body is the Gotran-generated list of assignments of `examples/*/mm_*.py`
(`rhs_numba(t, states, values, parameters)`: parameter unpacking, `math.` / `np.` expressions, stores into
`values[i]` and -- the side effect the PDEs read -- `parameters[i]`).  A host function cannot be called from a GPU
kernel, but such a body is plain arithmetic: this module translates its *source text* into the
`__device__ inline void rhs(double t, const double* states, double* values, double* parameters)` that
`knpemi_ode_bind_source` compiles with hipRTC, so a module written for the reference needs neither `MODEL_ID` nor a
hand-written `RHS_HIP`.  Nothing is imported or executed: the function is parsed with `ast` (numba and numbalsoda need
not be installed).

Supported: assignments to names / tuples of names, stores into `values[...]` / `parameters[...]`, `+ - * / **`,
unary `-`, comparisons (as 0.0 / 1.0 factors, the way `(t < 125e-3)` is used), conditional expressions, calls of
`math.` / `np.` / `numpy.` functions with a C counterpart, numeric literals, string statements (the commented-out blocks
Gotran modules carry) and a trailing `return`.  Anything else raises `NotImplementedError` naming the construct.
"""
from __future__ import annotations

import ast
import inspect
import textwrap

# exp / log go to the sweep's own device versions (csrc/lsoda_core.h: kn_exp, kn_log -- polynomials in Estrin's
# arrangement, within 4 ulp of the C library, a third of the dependent instructions of the library's): the right-hand
# side is the longest stretch of every trip of the integrator
_FUNCS = {
    "exp": "kn_exp", "log": "kn_log", "sqrt": "sqrt", "pow": "pow", "power": "pow", "fabs": "fabs", "abs": "fabs", "absolute": "fabs",
    "sin": "sin", "cos": "cos", "tan": "tan", "tanh": "tanh", "sinh": "sinh", "cosh": "cosh", "atan": "atan", "arctan": "atan",
    "log10": "log10", "log1p": "log1p", "expm1": "expm1", "floor": "floor", "ceil": "ceil", "fmod": "fmod", "mod": "fmod",
    "minimum": "fmin", "maximum": "fmax", "fmin": "fmin", "fmax": "fmax", "exp2": "exp2", "log2": "log2",
}
_CONSTS = {"pi": "3.141592653589793", "e": "2.718281828459045"}
_CMP = {ast.Lt: "<", ast.LtE: "<=", ast.Gt: ">", ast.GtE: ">=", ast.Eq: "==", ast.NotEq: "!="}
_ARRAYS = ("states", "values", "parameters")


class _Emitter:
    def __init__(self, argnames):
        self.t, self.states, self.values, self.parameters = argnames
        self.declared = set()
        self.lines = []

    # -- expressions --------------------------------------------------------------------------------------------
    def array(self, name):
        return {self.states: "states", self.values: "values", self.parameters: "parameters"}.get(name)

    def expr(self, e):
        if isinstance(e, ast.Constant):
            if isinstance(e.value, bool):
                return "1.0" if e.value else "0.0"
            if isinstance(e.value, (int, float)):
                return repr(float(e.value))
            raise NotImplementedError(f"constant {e.value!r} in an expression")
        if isinstance(e, ast.Name):
            if e.id == self.t:
                return "t"
            if self.array(e.id):
                raise NotImplementedError(f"array `{e.id}` used as a whole in an expression")
            if e.id not in self.declared:
                raise NotImplementedError(f"name `{e.id}` is read before it is assigned")
            return f"v_{e.id}"
        if isinstance(e, ast.Subscript):
            arr = self.array(e.value.id) if isinstance(e.value, ast.Name) else None
            idx = e.slice
            if arr is None or not (isinstance(idx, ast.Constant) and isinstance(idx.value, int)):
                raise NotImplementedError("only states[i] / values[i] / parameters[i] with a literal index are indexed")
            return f"{arr}[{idx.value}]"
        if isinstance(e, ast.UnaryOp):
            if isinstance(e.op, ast.USub):
                return f"(-{self.expr(e.operand)})"
            if isinstance(e.op, ast.UAdd):
                return self.expr(e.operand)
            raise NotImplementedError(f"unary operator {type(e.op).__name__}")
        if isinstance(e, ast.BinOp):
            a, b = self.expr(e.left), self.expr(e.right)
            if isinstance(e.op, ast.Add):
                return f"({a} + {b})"
            if isinstance(e.op, ast.Sub):
                return f"({a} - {b})"
            if isinstance(e.op, ast.Mult):
                return f"({a} * {b})"
            if isinstance(e.op, ast.Div):
                return f"({a} / {b})"
            if isinstance(e.op, ast.Pow):
                return f"pow({a}, {b})"
            if isinstance(e.op, ast.Mod):
                return f"fmod({a}, {b})"
            raise NotImplementedError(f"binary operator {type(e.op).__name__}")
        if isinstance(e, ast.Compare):
            if len(e.ops) != 1 or type(e.ops[0]) not in _CMP:
                raise NotImplementedError("chained or unusual comparison")
            return f"(({self.expr(e.left)} {_CMP[type(e.ops[0])]} {self.expr(e.comparators[0])}) ? 1.0 : 0.0)"
        if isinstance(e, ast.IfExp):
            return f"(({self.expr(e.test)}) != 0.0 ? {self.expr(e.body)} : {self.expr(e.orelse)})"
        if isinstance(e, ast.Attribute):          # math.pi, np.e
            if isinstance(e.value, ast.Name) and e.value.id in ("math", "np", "numpy") and e.attr in _CONSTS:
                return _CONSTS[e.attr]
            raise NotImplementedError(f"attribute `{ast.unparse(e)}`")
        if isinstance(e, ast.Call):
            f = e.func
            name = f.attr if isinstance(f, ast.Attribute) else (f.id if isinstance(f, ast.Name) else None)
            if isinstance(f, ast.Attribute) and not (isinstance(f.value, ast.Name) and f.value.id in ("math", "np", "numpy")):
                raise NotImplementedError(f"call of `{ast.unparse(f)}`")
            if name == "float" and len(e.args) == 1:
                return self.expr(e.args[0])
            if name not in _FUNCS or e.keywords:
                raise NotImplementedError(f"function `{ast.unparse(f)}` has no device counterpart here")
            return f"{_FUNCS[name]}({', '.join(self.expr(a) for a in e.args)})"
        raise NotImplementedError(f"expression `{ast.unparse(e)}`")

    # -- statements ---------------------------------------------------------------------------------------------
    def store(self, target, rhs):
        if isinstance(target, ast.Name):
            if target.id == self.t or self.array(target.id):
                raise NotImplementedError(f"assignment to the argument `{target.id}`")
            decl = "" if target.id in self.declared else "double "
            self.declared.add(target.id)
            self.lines.append(f"  {decl}v_{target.id} = {rhs};")
        elif isinstance(target, ast.Subscript):
            arr = self.array(target.value.id) if isinstance(target.value, ast.Name) else None
            if arr not in ("values", "parameters"):
                raise NotImplementedError(f"store into `{ast.unparse(target)}`")
            self.lines.append(f"  {self.expr(target)} = {rhs};")
        else:
            raise NotImplementedError(f"assignment target `{ast.unparse(target)}`")

    def statement(self, st):
        if isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant) and isinstance(st.value.value, str):
            return                                   # docstring / commented-out block
        if isinstance(st, ast.Return) or isinstance(st, ast.Pass):
            return
        if isinstance(st, ast.Assign):
            if len(st.targets) != 1:
                raise NotImplementedError("chained assignment")
            tgt, val = st.targets[0], st.value
            if isinstance(tgt, (ast.Tuple, ast.List)):
                if isinstance(val, ast.Name) and self.array(val.id):          # (a, b, c) = parameters
                    for i, el in enumerate(tgt.elts):
                        self.store(el, f"{self.array(val.id)}[{i}]")
                    return
                if isinstance(val, (ast.Tuple, ast.List)) and len(val.elts) == len(tgt.elts):
                    rhs = [self.expr(v) for v in val.elts]                    # evaluate all, then assign
                    tmp = []
                    for i, r in enumerate(rhs):
                        name = f"kn_tmp{len(self.lines)}_{i}"
                        self.lines.append(f"  const double {name} = {r};")
                        tmp.append(name)
                    for el, name in zip(tgt.elts, tmp):
                        self.store(el, name)
                    return
                raise NotImplementedError("tuple assignment from an expression")
            self.store(tgt, self.expr(val))
            return
        if isinstance(st, ast.AugAssign):
            op = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/"}.get(type(st.op))
            if op is None:
                raise NotImplementedError("augmented assignment operator")
            self.store(st.target, f"({self.expr(st.target)} {op} {self.expr(st.value)})")
            return
        raise NotImplementedError(f"statement `{type(st).__name__}` (line {getattr(st, 'lineno', '?')})")


def _function_def(tree, name):
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and (name is None or node.name == name):
            return node
    raise ValueError(f"no function `{name}` in the given source")


def hip_source_from_python(source, func=None):
    """HIP source of `rhs` from the Python source text `source` of a right-hand side with numbalsoda's signature
    `(t, states, values, parameters)`; `func` picks the function by name (first function definition otherwise)."""
    tree = ast.parse(textwrap.dedent(source))
    fd = _function_def(tree, func)
    args = [a.arg for a in fd.args.args]
    if len(args) != 4:
        raise NotImplementedError(f"`{fd.name}` does not have the signature (t, states, values, parameters)")
    em = _Emitter(args)
    for st in fd.body:
        em.statement(st)
    head = (f"// generated by knpemi.rhs_codegen from the Python function `{fd.name}`\n"
            "__device__ inline void rhs(double t, const double* states, double* values, double* parameters) {\n  (void)t;\n")
    return head + "\n".join(em.lines) + "\n}\n"


def hip_source_from_module(ode):
    """The module protocol's right-hand side (`rhs_numba`, a numba cfunc, or a plain `rhs`) as device source, or None if
    the module has neither.  The Python function behind a numba object is taken from its `_pyfunc` / `py_func`."""
    for name in ("rhs_numba", "rhs"):
        obj = getattr(ode, name, None)
        if obj is None:
            continue
        fn = getattr(obj, "_pyfunc", None) or getattr(obj, "py_func", None) or obj
        try:
            src = inspect.getsource(fn)
        except (OSError, TypeError):
            continue
        return hip_source_from_python(src, getattr(fn, "__name__", None))
    return None


def hip_source_from_file(path, func="rhs_numba"):
    """Device source of function `func` of the module file `path`, without importing it (the reference's modules
    import numba / numbalsoda at module level)."""
    with open(path) as f:
        return hip_source_from_python(f.read(), func)
