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
unary `-`, `%` / `np.mod` (Python's sign convention: a - b floor(a / b); `math.fmod` / `np.fmod` stay C's fmod),
comparisons (as 0.0 / 1.0 factors, the way `(t < 125e-3)` is used), conditional expressions, calls of
`math.` / `np.` / `numpy.` functions with a C counterpart and the matching number of arguments (`math.log(x, base)` becomes
a quotient of logarithms), numeric literals, string statements (the commented-out blocks
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
    "log10": "log10", "log1p": "log1p", "expm1": "expm1", "floor": "floor", "ceil": "ceil", "fmod": "fmod",
    "minimum": "fmin", "maximum": "fmax", "fmin": "fmin", "fmax": "fmax", "exp2": "exp2", "log2": "log2",
}
# number of arguments of each C counterpart (a call with another count is refused by name, not left to hipRTC)
_ARITY = {name: 2 if c in ("pow", "fmod", "fmin", "fmax") else 1 for name, c in _FUNCS.items()}


def _py_mod(a, b):
    """Python's `%` and np.mod take the sign of the DIVISOR, C's fmod that of the dividend: a - b floor(a / b) is the
    former (math.fmod / np.fmod are the latter and stay fmod)."""
    return f"({a} - {b} * floor({a} / {b}))"


_CONSTS = {"pi": "3.141592653589793", "e": "2.718281828459045"}
_CMP = {ast.Lt: "<", ast.LtE: "<=", ast.Gt: ">", ast.GtE: ">=", ast.Eq: "==", ast.NotEq: "!="}
_ARRAYS = ("states", "values", "parameters")


def _const_int(e):
    """Value of an integer literal or of `+ - *` arithmetic on integer literals (`parameters[15 + 1]`), else None."""
    if isinstance(e, ast.Constant) and isinstance(e.value, int) and not isinstance(e.value, bool):
        return e.value
    if isinstance(e, ast.UnaryOp) and isinstance(e.op, (ast.USub, ast.UAdd)):
        v = _const_int(e.operand)
        return None if v is None else (-v if isinstance(e.op, ast.USub) else v)
    if isinstance(e, ast.BinOp) and isinstance(e.op, (ast.Add, ast.Sub, ast.Mult)):
        a, b = _const_int(e.left), _const_int(e.right)
        if a is None or b is None:
            return None
        return a + b if isinstance(e.op, ast.Add) else a - b if isinstance(e.op, ast.Sub) else a * b
    return None


class _Emitter:
    def __init__(self, argnames, n_states=None, n_params=None):
        self.t, self.states, self.values, self.parameters = argnames
        self.sizes = {"states": n_states, "values": n_states, "parameters": n_params}
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
            idx = _const_int(e.slice)
            if arr is None or idx is None:
                raise NotImplementedError("only states[i] / values[i] / parameters[i] with a literal index are indexed")
            size = self.sizes[arr]
            if idx < 0 or (size is not None and idx >= size):
                raise NotImplementedError(f"index {idx} of `{arr}` is out of range" + (f" (0 .. {size - 1})" if size else ""))
            return f"{arr}[{idx}]"
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
                return _py_mod(a, b)
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
            if e.keywords:
                raise NotImplementedError(f"keyword arguments in the call of `{ast.unparse(f)}`")
            args = [self.expr(a) for a in e.args]
            if name in ("mod", "remainder") and len(args) == 2:      # np.mod / np.remainder: the sign of the divisor
                return _py_mod(args[0], args[1])
            if name == "log" and len(args) == 2:                      # math.log(x, base)
                return f"(kn_log({args[0]}) / kn_log({args[1]}))"
            if name not in _FUNCS:
                raise NotImplementedError(f"function `{ast.unparse(f)}` has no device counterpart here")
            if len(args) != _ARITY[name]:
                raise NotImplementedError(f"`{ast.unparse(f)}` called with {len(args)} argument(s), its device counterpart "
                                          f"`{_FUNCS[name]}` takes {_ARITY[name]}")
            return f"{_FUNCS[name]}({', '.join(args)})"
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
                    size = self.sizes[self.array(val.id)]
                    if size is not None and len(tgt.elts) != size:            # Python raises ValueError here
                        raise NotImplementedError(f"{len(tgt.elts)} names unpacked from `{val.id}`, which has {size} entries")
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


def hip_source_from_python(source, func=None, n_states=None, n_params=None):
    """HIP source of `rhs` from the Python source text `source` of a right-hand side with numbalsoda's signature
    `(t, states, values, parameters)`; `func` picks the function by name (first function definition otherwise).
    `n_states` / `n_params`, when known, are the lengths `(a, b, ...) = states` / `= parameters` must unpack and the bound
    of every literal index."""
    tree = ast.parse(textwrap.dedent(source))
    fd = _function_def(tree, func)
    args = [a.arg for a in fd.args.args]
    if len(args) != 4:
        raise NotImplementedError(f"`{fd.name}` does not have the signature (t, states, values, parameters)")
    em = _Emitter(args, n_states, n_params)
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
        sizes = {}
        for key, init in (("n_states", "init_state_values"), ("n_params", "init_parameter_values")):
            try:
                sizes[key] = len(getattr(ode, init)())
            except Exception:      # noqa: BLE001 -- a module without the tables: no bound to check against
                sizes[key] = None
        return hip_source_from_python(src, getattr(fn, "__name__", None), **sizes)
    return None


def hip_source_from_file(path, func="rhs_numba"):
    """Device source of function `func` of the module file `path`, without importing it (the reference's modules
    import numba / numbalsoda at module level)."""
    with open(path) as f:
        return hip_source_from_python(f.read(), func)
