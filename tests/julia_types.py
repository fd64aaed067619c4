"""A small, syntactic model of Julia's `invoke` method lookup — test infrastructure for tests/test_julia_dispatch.py.

`invoke(f, argtypes, args...)` runs the most specific method of `f` whose signature is a SUPERTYPE of `argtypes`
(`argtypes <: signature`); when no method covers the tuple it throws a MethodError.  Julia is not installed where this project is
built or tested, so the binding's fall-backs (`julia/ElPhGPU.jl`) are checked here by restating the part of the subtype relation they
need:

  * nominal types with declared parameters and supertypes (`DECLS`: the reference's struct / abstract type headers, the few Base types
    the signatures use), parameters invariant, `Tuple` covariant;
  * a method's type variables (`where {T1, T2<:Number}`) are existential: every INVARIANT occurrence of one variable must be bound to
    one and the same type, within the variable's bound;
  * a bare name (`SSHModel` for `SSHModel{T1,T2,T3,T4} where …`) stands for the union over all its parameters.  Such an unspecified
    parameter can meet a method's type variable only when that variable occurs ONCE in the whole signature (then
    `Tuple{SSHModel{T}} where T == Tuple{SSHModel}`), and only if the parameter's declared bound lies within the variable's bound;
    a variable that is shared between arguments (`y::AbstractVector{T2}, model::AbstractModel{T1,T2}`) cannot take "every value";
  * `typeof(x)` is the concrete type of the value bound to `x`: the declared type of `x` in the enclosing method, its unspecified
    trailing parameters being fixed-but-unknown ("opaque") types — they bind any variable, equal only to themselves.

Nothing here is imported by the product; it never sees a GPU.
"""
import re

# ---- declarations: name -> ([(parameter, bound or None)], supertype text or None) ------------------------------------------------
# Base / stdlib
DECLS = {
    "Any": ([], None),
    "Number": ([], "Any"), "Real": ([], "Number"), "AbstractFloat": ([], "Real"), "Float64": ([], "AbstractFloat"),
    "Integer": ([], "Real"), "Int": ([], "Integer"), "Int64": ([], "Integer"), "Bool": ([], "Integer"),
    "Complex": ([("T", "Real")], "Number"),
    "AbstractVector": ([("T", None)], "Any"), "Vector": ([("T", None)], "AbstractVector{T}"),
    "AbstractMatrix": ([("T", None)], "Any"), "Matrix": ([("T", None)], "AbstractMatrix{T}"),
    "UniformScaling": ([("T", "Number")], "Any"),
    "AbstractRNG": ([], "Any"),
    # the reference's own (headers re-derived from /root/reference/src when it is present: test_declarations_are_the_reference)
    "IterativeSolver": ([("Ttol", "AbstractFloat"), ("Tdata", "Continuous")], "Any"),                  # IterativeSolvers.jl:27
    "ConjugateGradient": ([("Ttol", None), ("Tdata", None)], "IterativeSolver{Ttol,Tdata}"),          # IterativeSolvers.jl:36
    "AbstractModel": ([("T1", "AbstractFloat"), ("T2", "Continuous"), ("T3", "IterativeSolver"), ("T4", "AbstractRNG")], "Any"),  # Models.jl:65
    "HolsteinModel": ([("T1", None), ("T2", None), ("T3", None), ("T4", None)], "AbstractModel{T1,T2,T3,T4}"),   # HolsteinModels.jl:22
    "SSHModel": ([("T1", None), ("T2", None), ("T3", None), ("T4", None)], "AbstractModel{T1,T2,T3,T4}"),        # SSHModels.jl:79
    "KPMPreconditioner": ([("T1", "AbstractFloat"), ("T2", "Continuous"), ("T3", "AbstractModel")], "Any"),     # KPMPreconditioners.jl:153
    "SymmetricKPMPreconditioner": ([("T1", None), ("T2", None), ("T3", None)], "KPMPreconditioner{T1,T2,T3}"),  # KPMPreconditioners.jl:219
    "KPMExpansion": ([("T1", "AbstractFloat"), ("T2", "Continuous"), ("T3", "AbstractModel")], "Any"),          # KPMPreconditioners.jl:21
    "HybridMonteCarlo": ([("T", "AbstractFloat")], "Any"),                                                      # HMC.jl:20
    "FourierAccelerator": ([("T", "AbstractFloat"), ("Tfft", None), ("Tifft", None)], "Any"),                   # FourierAcceleration.jl:11
}
ALIASES = {
    "Continuous": "Union{AbstractFloat,Complex{<:AbstractFloat}}",                                              # Models.jl:20
    # the binding's own
    "GPUModel": "Union{HolsteinModel{Float64,Float64},SSHModel{Float64,Float64}}",
    "GPUSymmetricKPM": "SymmetricKPMPreconditioner{Float64,Float64,<:GPUModel}",
}


# ---- parsing -------------------------------------------------------------------------------------------------------------------
def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def parse_type(s, tvars=()):
    """Type text -> ('any',) | ('var', T) | ('ub', type) | ('typeof', ident) | ('union', [types]) | ('app', name, [params] | None)."""
    s = s.strip()
    if s.startswith("<:"):
        return ("ub", parse_type(s[2:], tvars))
    m = re.match(r"^typeof\(\s*([^\s()]+)\s*\)$", s)
    if m:
        return ("typeof", m.group(1))
    m = re.match(r"^([^\s{}<>:,()]+)(\{(.*)\})?$", s, re.S)
    assert m, f"cannot parse type {s!r}"
    name, has, inner = m.group(1), m.group(2), m.group(3)
    if name in tvars:
        assert not has
        return ("var", name)
    if name in ALIASES and not has:
        return parse_type(ALIASES[name], tvars)
    if name == "Union":
        return ("union", [parse_type(p, tvars) for p in split_top(inner)])
    if name == "Any" and not has:
        return ("any",)
    assert name in DECLS or name == "Tuple", f"type {name!r} is not in the declaration table (tests/julia_types.py DECLS)"
    return ("app", name, [parse_type(p, tvars) for p in split_top(inner)] if has else None)


def parse_where(s):
    """`{T1<:AbstractFloat,T2}` or `T` -> {name: bound text or None}"""
    s = s.strip()
    if s.startswith("{"):
        s = s[1:s.rindex("}")]
    out = {}
    for p in split_top(s):
        m = re.match(r"^([^\s<:]+)\s*(?:<:\s*(.+))?$", p)
        out[m.group(1)] = m.group(2)
    return out


def parse_method(text):
    """`function f(a::A, b=I, c::T=1.0; kw...)::Ret where {…}` (one string, newlines allowed) ->
    [(name, [argument types], {tvar: bound}, [argument names])] — one entry per arity the default values generate."""
    m = re.match(r"^\s*function\s+([^\s(]+)\(", text)
    assert m, text
    name = m.group(1)
    depth, j = 1, m.end()
    while depth:
        depth += {"(": 1, ")": -1}.get(text[j], 0)
        j += 1
    args_txt, rest = text[m.end():j - 1], text[j:]
    wh = re.search(r"\bwhere\s+(\{[^}]*\}|[^\s]+)", rest)
    tv = parse_where(wh.group(1)) if wh else {}
    positional = split_top(args_txt, ";")[0] if args_txt.strip() else ""
    args = []
    for a in split_top(positional):
        default = None
        parts = split_top(a, "=")
        if len(parts) == 2:
            a, default = parts
        nm, _, ty = a.partition("::")
        args.append((nm.strip(), ty.strip() or "Any", default))
    out = []
    n_req = len([a for a in args if a[2] is None])
    for n in range(n_req, len(args) + 1):
        out.append((name, [a[1] for a in args[:n]], tv, [a[0] for a in args[:n]]))
    return out


# ---- the subtype check ---------------------------------------------------------------------------------------------------------
class NoMatch(Exception):
    pass


def _occurrences(t, name):
    if t[0] == "var":
        return int(t[1] == name)
    if t[0] in ("ub",):
        return _occurrences(t[1], name)
    if t[0] == "union":
        return sum(_occurrences(x, name) for x in t[1])
    if t[0] == "app" and t[2]:
        return sum(_occurrences(x, name) for x in t[2])
    return 0


def _supertype(t):
    """('app', name, params) -> its declared supertype with the parameters substituted, or None at the top."""
    name, params = t[1], t[2]
    decl, sup = DECLS[name]
    if sup is None or sup == "Any":
        return None
    names = [d[0] for d in decl]
    st = parse_type(sup, tuple(names))

    def subst(x):
        if x[0] == "var":
            i = names.index(x[1])
            if params is None or i >= len(params):
                return ("unspec", name, i)
            return params[i]
        if x[0] == "app" and x[2] is not None:
            return ("app", x[1], [subst(p) for p in x[2]])
        return x
    return subst(st)


def _declared_bound(name, i):
    b = DECLS[name][0][i][1]
    return parse_type(b) if b else ("any",)


class Matcher:
    def __init__(self, sig_types, tvars):
        self.tvars = tvars
        self.sig = [parse_type(t, tuple(tvars)) for t in sig_types]
        self.count = {v: sum(_occurrences(t, v) for t in self.sig) for v in tvars}
        self.bound = {v: (parse_type(b, tuple(tvars)) if b else ("any",)) for v, b in tvars.items()}
        self.binding = {}

    def covers(self, arg_types):
        """arg_types (parsed, typeof already resolved) <: this signature?  -> (True, None) | (False, reason)"""
        if len(arg_types) != len(self.sig):
            return False, f"arity {len(arg_types)} vs {len(self.sig)}"
        self.binding = {}
        try:
            for k, (a, b) in enumerate(zip(arg_types, self.sig)):
                try:
                    self.sub(a, b)
                except NoMatch as e:
                    raise NoMatch(f"argument {k + 1}: {e}")
        except NoMatch as e:
            return False, str(e)
        return True, None

    # a <: b, covariant position
    def sub(self, a, b):
        if b[0] == "any":
            return
        if a[0] == "union":
            for x in a[1]:
                self.sub(x, b)
            return
        if b[0] == "union":
            errs = []
            for x in b[1]:
                saved = dict(self.binding)
                try:
                    self.sub(a, x)
                    return
                except NoMatch as e:
                    self.binding = saved
                    errs.append(str(e))
            raise NoMatch(" / ".join(errs))
        if b[0] == "ub":
            return self.sub(a, b[1])
        if b[0] == "var":                       # `power::T`: a leaf type binds it, within the bound
            if a[0] in ("any",):
                raise NoMatch(f"Any is not within {b[1]}")
            self.bind(b[1], a)
            return
        if a[0] in ("opaque",):
            return                              # a parameter of a live value: it satisfied its bounds when the value was made
        if a[0] == "unspec":
            return self.sub(_declared_bound(a[1], a[2]), b)
        if a[0] == "any":
            raise NoMatch(f"Any is not a subtype of {show(b)}")
        assert a[0] == "app" and b[0] == "app", (a, b)
        t = a
        while t is not None and t[1] != b[1]:
            t = _supertype(t)
            if t is not None and t[0] != "app":
                t = None
        if t is None:
            raise NoMatch(f"{show(a)} has no supertype {b[1]}")
        if b[2] is None:
            return
        nparams = len(DECLS[b[1]][0])
        for i in range(nparams):
            pa = t[2][i] if (t[2] is not None and i < len(t[2])) else ("unspec", t[1], i)
            if i >= len(b[2]):
                break                           # trailing parameters of the signature's type are free
            self.inv(pa, b[2][i], f"{show(a)} vs {show(b)}, parameter {i + 1}")

    # invariant position
    def inv(self, pa, pb, where):
        if pb[0] == "var":
            v = pb[1]
            if pa[0] == "unspec":
                if self.count[v] != 1:
                    raise NoMatch(f"{where}: the parameter is unspecified (bare UnionAll) but the method ties `{v}` to "
                                  f"{self.count[v]} places of its signature — no single value of `{v}` covers every instance")
                try:
                    self.sub(_declared_bound(pa[1], pa[2]), self.bound[v])
                except NoMatch:
                    raise NoMatch(f"{where}: unspecified parameter (declared bound {show(_declared_bound(pa[1], pa[2]))}) "
                                  f"exceeds the bound of `{v}` ({show(self.bound[v])})")
                return
            if pa[0] == "ub":
                raise NoMatch(f"{where}: `<:` parameter against type variable `{v}`")
            self.bind(v, pa)
            return
        if pb[0] == "ub":
            if pa[0] == "unspec":
                return self.sub(_declared_bound(pa[1], pa[2]), pb[1])
            if pa[0] == "ub":
                return self.sub(pa[1], pb[1])
            return self.sub(pa, pb[1])
        if pa[0] in ("unspec", "ub"):
            raise NoMatch(f"{where}: unspecified parameter against the fixed {show(pb)}")
        if pa != pb:
            raise NoMatch(f"{where}: {show(pa)} is not {show(pb)} (parameters are invariant)")

    def bind(self, v, a):
        if v in self.binding:
            if self.binding[v] != a:
                raise NoMatch(f"type variable `{v}` would have to be both {show(self.binding[v])} and {show(a)}")
            return
        if a[0] != "opaque":
            try:
                self.sub(a, self.bound[v])
            except NoMatch:
                raise NoMatch(f"{show(a)} is outside the bound of `{v}` ({show(self.bound[v])})")
        self.binding[v] = a


def show(t):
    if t[0] == "any":
        return "Any"
    if t[0] == "var":
        return t[1]
    if t[0] == "ub":
        return "<:" + show(t[1])
    if t[0] == "union":
        return "Union{" + ",".join(show(x) for x in t[1]) + "}"
    if t[0] == "opaque":
        return f"<parameter {t[2] + 1} of typeof({t[1]})>"
    if t[0] == "unspec":
        return f"<any {DECLS[t[1]][0][t[2]][0]} of {t[1]}>"
    if t[0] == "typeof":
        return f"typeof({t[1]})"
    return t[1] + ("{" + ",".join(show(x) for x in t[2]) + "}" if t[2] is not None else "")


def concretise(t, var):
    """typeof(var) where `var` was declared `::t`: unspecified trailing parameters become fixed-but-unknown types."""
    if t[0] == "union":
        return ("union", [concretise(x, var) for x in t[1]])
    assert t[0] == "app", f"typeof({var}): declared type {show(t)} is not a nominal type"
    n = len(DECLS[t[1]][0])
    params = list(t[2] or [])
    for i in range(len(params), n):
        params.append(("opaque", var, i))
    return ("app", t[1], params)


def resolve_typeof(t, declared):
    """Replace ('typeof', x) by the concrete form of x's declared type (`declared`: {argument name: parsed type})."""
    if t[0] == "typeof":
        assert t[1] in declared, f"typeof({t[1]}): `{t[1]}` is not an argument of the enclosing method"
        return concretise(declared[t[1]], t[1])
    if t[0] == "app" and t[2] is not None:
        return ("app", t[1], [resolve_typeof(p, declared) for p in t[2]])
    if t[0] == "union":
        return ("union", [resolve_typeof(p, declared) for p in t[1]])
    return t


def expand_unions(types):
    """A tuple type whose elements contain unions of concrete alternatives (typeof of a `::GPUModel` argument) -> every alternative."""
    out = [[]]
    for t in types:
        alts = t[1] if t[0] == "union" and all(x[0] == "app" and x[2] and any(p[0] == "opaque" for p in x[2]) for x in t[1]) else [t]
        out = [o + [a] for o in out for a in alts]
    return out
