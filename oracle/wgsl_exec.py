"""wgsl_exec -- a small interpreter for the WGSL subset the reference's hot-path shaders are written in.

TEST INFRASTRUCTURE ONLY (like ws_oracle.c): used by tests/golden/gen_wgsl_golden.py, in the build container, to run the
reference's own shader SOURCE TEXT (read from the reference checkout at generation time, never copied into this
repository) on seeded inputs and to record the outputs as golden vectors under tests/golden/.  The oracle and the HIP
kernels are then compared with those vectors, so that K1 / K1c / the K6 fragment function are pinned to the reference's
source rather than to a reading of it.  Nothing in the product path imports this file.

What is implemented is the W3C WGSL semantics of exactly what those shaders use:
  * declarations: const, struct (with @align), var<uniform|storage>, fn; attributes are parsed and ignored otherwise;
  * statements: let / var / assignment / compound assignment / ++ / -- / if-else / for / while / loop / break / continue /
    return / discard / call statements;
  * types: f32 u32 i32 bool, vecN<T>, matCxR<f32>, array<T,N>, array<T>, atomic<T>, structs;
  * the memory layout rules (align / size / stride, section "Memory Layout") used to decode the bound buffers from raw bytes
    and to encode stores, so that the host-side uniform structs are checked against the shader's own declaration;
  * f32 arithmetic rounds after every operation (numpy float32 scalars), abstract numerics fold in double precision and
    convert when they meet a concrete type, u32 / i32 wrap;
  * built-ins: unpack2x16float pack2x16float unpack4x8snorm extractBits any all dot length distance normalize transpose
    smoothstep clamp min max abs sqrt exp log floor select bitcast atomicAdd atomicLoad atomicStore arrayLength and the
    value constructors.  length = sqrt(x*x + y*y [+ z*z]) summed left to right, normalize = v / length(v),
    smoothstep = t*t*(3 - 2t): the spec's definitions, without fused operations.
dispatch(): invocations run one after the other in global-invocation order, so atomicAdd hands out consecutive values in
that order.  dispatch_workgroups(): for entry points whose invocations cooperate -- var<workgroup> / var<private>,
workgroupBarrier(): workgroups in workgroup-id order, the invocations of one as threads that meet at the barriers, atomics
serialised, WebGPU's robust buffer access on request (out-of-bounds loads give zero, stores are dropped).  What it does NOT
model is subgroup lock-step: code that reads a neighbour lane's store without a barrier needs subgroup size 1.
"""
import re
import struct

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------------------------------
# values
# ----------------------------------------------------------------------------------------------------------------------
class U32(int):
    pass


class I32(int):
    pass


def u32(v):
    return U32(int(v) & 0xFFFFFFFF)


def i32(v):
    v = int(v) & 0xFFFFFFFF
    return I32(v - (1 << 32) if v & 0x80000000 else v)


class Vec:
    __slots__ = ("c",)

    def __init__(self, comps):
        self.c = list(comps)

    def __len__(self):
        return len(self.c)

    def __repr__(self):
        return "vec%d(%s)" % (len(self.c), ", ".join(repr(x) for x in self.c))


class Mat:
    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = list(cols)  # list of Vec (columns)

    def __repr__(self):
        return "mat(%r)" % (self.cols,)


class StructVal:
    def __init__(self, tname, fields):
        self.tname = tname
        self.f = dict(fields)

    def copy(self):
        return StructVal(self.tname, {k: copy_val(v) for k, v in self.f.items()})

    def __repr__(self):
        return "%s(%r)" % (self.tname, self.f)


def copy_val(v):
    if isinstance(v, Vec):
        return Vec(v.c)
    if isinstance(v, Mat):
        return Mat([Vec(c.c) for c in v.cols])
    if isinstance(v, StructVal):
        return v.copy()
    if isinstance(v, list):
        return [copy_val(x) for x in v]
    return v


class Discard(Exception):
    pass


class _Return(Exception):
    def __init__(self, v):
        self.v = v


class _Break(Exception):
    pass


class _Continue(Exception):
    pass


def is_abstract(v):
    return type(v) in (int, float)


def concretize_like(v, other):
    """An abstract numeric meeting a concrete scalar takes that scalar's type."""
    if type(v) in (int, float):
        if isinstance(other, F32):
            return F32(v)
        if isinstance(other, U32):
            if type(v) is float:
                raise TypeError("abstract float with u32")
            return u32(v)
        if isinstance(other, I32):
            if type(v) is float:
                raise TypeError("abstract float with i32")
            return i32(v)
    return v


def scalar_binop(op, a, b):
    a = concretize_like(a, b)
    b = concretize_like(b, a)
    ta, tb = type(a), type(b)
    if ta is bool or tb is bool:
        if op == "==":
            return a == b
        if op == "!=":
            return a != b
        if op in ("&", "&&"):
            return bool(a and b)
        if op in ("|", "||"):
            return bool(a or b)
        raise TypeError("bool op " + op)
    if ta is int and tb is float:
        a = float(a)
        ta = float
    if ta is float and tb is int:
        b = float(b)
        tb = float
    if ta is not tb:
        raise TypeError("type mismatch %s %s %s" % (ta.__name__, op, tb.__name__))
    if op in ("==", "!=", "<", ">", "<=", ">="):
        return bool({"==": a == b, "!=": a != b, "<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op])
    if ta is F32:
        with np.errstate(all="ignore"):
            if op == "+":
                return F32(a + b)
            if op == "-":
                return F32(a - b)
            if op == "*":
                return F32(a * b)
            if op == "/":
                return F32(a / b)
        raise TypeError("f32 op " + op)
    if ta is float:
        return {"+": a + b, "-": a - b, "*": a * b, "/": a / b if b != 0 else float("inf")}[op]
    wrap = u32 if ta is U32 else (i32 if ta is I32 else int)
    if op == "+":
        return wrap(int(a) + int(b))
    if op == "-":
        return wrap(int(a) - int(b))
    if op == "*":
        return wrap(int(a) * int(b))
    if op == "/":
        if int(b) == 0:
            return a
        q = abs(int(a)) // abs(int(b))
        return wrap(q if (int(a) < 0) == (int(b) < 0) else -q)  # truncating division
    if op == "%":
        if int(b) == 0:
            return wrap(0)
        r = abs(int(a)) % abs(int(b))
        return wrap(r if int(a) >= 0 else -r)
    if op == "&":
        return wrap(int(a) & int(b))
    if op == "|":
        return wrap(int(a) | int(b))
    if op == "^":
        return wrap(int(a) ^ int(b))
    if op == "<<":
        return wrap(int(a) << (int(b) & 31))
    if op == ">>":
        return wrap(int(a) >> (int(b) & 31))
    raise TypeError("int op " + op)


def dot(a, b):
    acc = scalar_binop("*", a.c[0], b.c[0])
    for x, y in zip(a.c[1:], b.c[1:]):
        acc = scalar_binop("+", acc, scalar_binop("*", x, y))
    return acc


def mat_vec(m, v):
    """Column-major M * v = sum_c M[c] * v[c], accumulated left to right."""
    rows = len(m.cols[0])
    out = []
    for r in range(rows):
        acc = scalar_binop("*", m.cols[0].c[r], v.c[0])
        for c in range(1, len(m.cols)):
            acc = scalar_binop("+", acc, scalar_binop("*", m.cols[c].c[r], v.c[c]))
        out.append(acc)
    return Vec(out)


def binop(op, a, b):
    va, vb = isinstance(a, Vec), isinstance(b, Vec)
    ma, mb = isinstance(a, Mat), isinstance(b, Mat)
    if ma or mb:
        if op == "*":
            if ma and mb:
                return Mat([mat_vec(a, col) for col in b.cols])
            if ma and vb:
                return mat_vec(a, b)
            if va and mb:  # row vector * matrix
                return Vec([dot(a, col) for col in b.cols])
            if ma:
                return Mat([Vec([scalar_binop("*", x, b) for x in col.c]) for col in a.cols])
            return Mat([Vec([scalar_binop("*", a, x) for x in col.c]) for col in b.cols])
        if op in ("+", "-") and ma and mb:
            return Mat([Vec([scalar_binop(op, x, y) for x, y in zip(ca.c, cb.c)]) for ca, cb in zip(a.cols, b.cols)])
        raise TypeError("matrix op " + op)
    if va and vb:
        if len(a) != len(b):
            raise TypeError("vector size mismatch")
        return Vec([scalar_binop(op, x, y) for x, y in zip(a.c, b.c)])
    if va:
        return Vec([scalar_binop(op, x, b) for x in a.c])
    if vb:
        return Vec([scalar_binop(op, a, y) for y in b.c])
    return scalar_binop(op, a, b)


def unary(op, a):
    if isinstance(a, Vec):
        return Vec([unary(op, x) for x in a.c])
    if op == "-":
        if isinstance(a, F32):
            return F32(-a)
        if isinstance(a, U32):
            raise TypeError("negating u32")
        if isinstance(a, I32):
            return i32(-int(a))
        return -a
    if op == "!":
        return not a
    if op == "~":
        return u32(~int(a)) if isinstance(a, U32) else i32(~int(a))
    raise TypeError("unary " + op)


def to_f32(v):
    if isinstance(v, Vec):
        return Vec([to_f32(x) for x in v.c])
    if type(v) is bool:
        return F32(1.0 if v else 0.0)
    return F32(float(v)) if not isinstance(v, F32) else v


def to_u32(v):
    if isinstance(v, Vec):
        return Vec([to_u32(x) for x in v.c])
    if isinstance(v, (F32, float)):
        x = float(v)
        if x != x:
            return u32(0)
        return u32(min(max(int(x), 0), 0xFFFFFFFF))  # truncation towards zero, clamped (value conversion)
    if type(v) is bool:
        return u32(1 if v else 0)
    return u32(int(v))  # i32 -> u32 reinterprets the bits


def to_i32(v):
    if isinstance(v, Vec):
        return Vec([to_i32(x) for x in v.c])
    if isinstance(v, (F32, float)):
        x = float(v)
        if x != x:
            return i32(0)
        return i32(min(max(int(x), -(1 << 31)), (1 << 31) - 1))
    if type(v) is bool:
        return i32(1 if v else 0)
    return i32(int(v))


def to_bool(v):
    if isinstance(v, Vec):
        return Vec([to_bool(x) for x in v.c])
    return bool(v != 0)


def f16_bits_to_f32(h):
    return F32(np.array([h], dtype=np.uint16).view(np.float16)[0])


def f32_to_f16_bits(x):
    with np.errstate(all="ignore"):
        return int(np.array([x], dtype=np.float32).astype(np.float16).view(np.uint16)[0])


# ----------------------------------------------------------------------------------------------------------------------
# lexer / parser
# ----------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F]+[ui]?|(?:[0-9]+\.[0-9]*|\.[0-9]+|[0-9]+)(?:[eE][+-]?[0-9]+)?[fuih]?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>->|\+\+|--|&&|\|\||==|!=|<=|>=|<<|>>|\+=|-=|\*=|/=|%=|&=|\|=|\^=|[-+*/%&|^~!<>=.,;:(){}\[\]@])
""", re.X | re.S)

TEMPLATED = {"vec2", "vec3", "vec4", "array", "atomic", "bitcast", "ptr",
             "mat2x2", "mat2x3", "mat2x4", "mat3x2", "mat3x3", "mat3x4", "mat4x2", "mat4x3", "mat4x4"}


def tokenize(src):
    out = []
    pos = 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError("cannot tokenize at %r" % src[pos:pos + 30])
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group(m.lastgroup)))
    out.append(("eof", ""))
    return out


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    # -- token helpers
    def peek(self, k=0):
        return self.t[self.i + k]

    def at(self, val):
        return self.t[self.i][1] == val and self.t[self.i][0] != "num"

    def eat(self, val=None):
        tok = self.t[self.i]
        if val is not None and tok[1] != val:
            raise SyntaxError("expected %r, got %r (token %d)" % (val, tok[1], self.i))
        self.i += 1
        return tok[1]

    def eat_gt(self):
        """Close a template list: '>>' / '>=' are split."""
        kind, v = self.t[self.i]
        if v == ">":
            self.i += 1
        elif v == ">>":
            self.t[self.i] = (kind, ">")
        elif v == ">=":
            self.t[self.i] = (kind, "=")
        else:
            raise SyntaxError("expected '>' got %r" % v)

    def attributes(self):
        attrs = {}
        while self.at("@"):
            self.eat("@")
            name = self.eat()
            args = []
            if self.at("("):
                self.eat("(")
                while not self.at(")"):
                    args.append(self.expr())
                    if self.at(","):
                        self.eat(",")
                self.eat(")")
            attrs[name] = args
        return attrs

    # -- types: ('T', name, [args])  (args: types or integer expressions)
    def type_(self):
        name = self.eat()
        args = []
        if self.at("<"):
            self.eat("<")
            while True:
                if name == "array" and args:          # array<T, N>: the element count is an expression
                    args.append(("expr", self.add()))
                elif self.peek()[0] == "id" and (self.peek()[1] in TEMPLATED or self.peek(1)[1] in (",", ">", ">>")):
                    args.append(self.type_())
                else:
                    args.append(("expr", self.add()))
                if self.at(","):
                    self.eat(",")
                    continue
                break
            self.eat_gt()
        return ("T", name, args)

    # -- module
    def module(self):
        decls = []
        while self.peek()[0] != "eof":
            attrs = self.attributes()
            tok = self.peek()[1]
            if tok == "const":
                self.eat()
                name = self.eat()
                ty = None
                if self.at(":"):
                    self.eat(":")
                    ty = self.type_()
                self.eat("=")
                e = self.expr()
                self.eat(";")
                decls.append(("const", name, ty, e))
            elif tok == "struct":
                self.eat()
                name = self.eat()
                self.eat("{")
                fields = []
                while not self.at("}"):
                    fa = self.attributes()
                    fname = self.eat()
                    self.eat(":")
                    fty = self.type_()
                    fields.append((fname, fty, fa))
                    if self.at(","):
                        self.eat(",")
                self.eat("}")
                if self.at(";"):
                    self.eat(";")
                decls.append(("struct", name, fields))
            elif tok == "var":
                self.eat()
                space = []
                if self.at("<"):
                    self.eat("<")
                    while not self.at(">"):
                        space.append(self.eat())
                        if self.at(","):
                            self.eat(",")
                    self.eat(">")
                name = self.eat()
                self.eat(":")
                ty = self.type_()
                self.eat(";")
                decls.append(("var", name, ty, space, attrs))
            elif tok == "fn":
                self.eat()
                name = self.eat()
                self.eat("(")
                params = []
                while not self.at(")"):
                    pa = self.attributes()
                    pname = self.eat()
                    self.eat(":")
                    params.append((pname, self.type_(), pa))
                    if self.at(","):
                        self.eat(",")
                self.eat(")")
                ret = None
                if self.at("->"):
                    self.eat("->")
                    self.attributes()
                    ret = self.type_()
                body = self.block()
                decls.append(("fn", name, params, ret, body, attrs))
            elif tok == ";":
                self.eat()
            else:
                raise SyntaxError("unexpected top-level token %r" % tok)
        return decls

    # -- statements
    def block(self):
        self.eat("{")
        stmts = []
        while not self.at("}"):
            stmts.append(self.statement())
        self.eat("}")
        return stmts

    def statement(self):
        tok = self.peek()[1]
        if self.peek()[0] == "num":
            tok = None
        if tok in ("let", "var"):
            self.eat()
            name = self.eat()
            ty = None
            if self.at(":"):
                self.eat(":")
                ty = self.type_()
            e = None
            if self.at("="):
                self.eat("=")
                e = self.expr()
            self.eat(";")
            return (tok, name, ty, e)
        if tok == "return":
            self.eat()
            e = None if self.at(";") else self.expr()
            self.eat(";")
            return ("return", e)
        if tok == "discard":
            self.eat()
            self.eat(";")
            return ("discard",)
        if tok == "if":
            self.eat()
            cond = self.expr()
            then = self.block()
            els = None
            if self.at("else"):
                self.eat()
                els = [self.statement()] if self.at("if") else self.block()
            return ("if", cond, then, els)
        if tok == "{":
            return ("block", self.block())
        if tok in ("break", "continue"):
            self.eat()
            self.eat(";")
            return (tok,)
        if tok == "while":
            self.eat()
            cond = self.expr()
            return ("loop", None, cond, None, self.block())
        if tok == "loop":
            self.eat()
            return ("loop", None, None, None, self.block())
        if tok == "for":
            self.eat()
            self.eat("(")
            init = None if self.at(";") else self.statement()      # (consumes its own ';')
            if init is None:
                self.eat(";")
            cond = None if self.at(";") else self.expr()
            self.eat(";")
            step = None if self.at(")") else self.simple_statement()
            self.eat(")")
            return ("loop", init, cond, step, self.block())
        st = self.simple_statement()
        self.eat(";")
        return st

    def simple_statement(self):
        """assignment / compound assignment / increment / call, without the trailing ';' (also a for-loop's step)."""
        lhs = self.expr()
        if self.at("++") or self.at("--"):
            op = self.eat()
            return ("assign", "+=" if op == "++" else "-=", lhs, ("num", 1))
        if self.at(";") or self.at(")"):
            return ("expr", lhs)
        op = self.eat()
        if op not in ("=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="):
            raise SyntaxError("unexpected %r in statement" % op)
        return ("assign", op, lhs, self.expr())

    # -- expressions
    def _left(self, sub, ops):
        e = sub()
        while self.peek()[0] == "op" and self.peek()[1] in ops:
            op = self.eat()
            e = ("bin", op, e, sub())
        return e

    def expr(self):
        return self._left(self.and_, ("||",))

    def and_(self):
        return self._left(self.bor, ("&&",))

    def bor(self):
        return self._left(self.bxor, ("|",))

    def bxor(self):
        return self._left(self.band, ("^",))

    def band(self):
        return self._left(self.eq, ("&",))

    def eq(self):
        return self._left(self.rel, ("==", "!="))

    def rel(self):
        return self._left(self.shift, ("<", ">", "<=", ">="))

    def shift(self):
        return self._left(self.add, ("<<", ">>"))

    def add(self):
        return self._left(self.mul, ("+", "-"))

    def mul(self):
        return self._left(self.unary, ("*", "/", "%"))

    def unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "!", "~", "&", "*"):
            op = self.eat()
            return ("un", op, self.unary())
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.at("("):
                self.eat("(")
                args = []
                while not self.at(")"):
                    args.append(self.expr())
                    if self.at(","):
                        self.eat(",")
                self.eat(")")
                e = ("call", e, args)
            elif self.at("["):
                self.eat("[")
                idx = self.expr()
                self.eat("]")
                e = ("index", e, idx)
            elif self.at("."):
                self.eat(".")
                e = ("member", e, self.eat())
            else:
                return e

    def primary(self):
        kind, v = self.peek()
        if kind == "num":
            self.eat()
            return ("num", parse_number(v))
        if v == "(":
            self.eat("(")
            e = self.expr()
            self.eat(")")
            return e
        if v in ("true", "false"):
            self.eat()
            return ("num", v == "true")
        if kind == "id":
            if v in TEMPLATED and self.peek(1)[1] == "<":
                return ("type", self.type_())
            self.eat()
            return ("id", v)
        raise SyntaxError("unexpected token %r" % v)


def parse_number(s):
    if s[:2] in ("0x", "0X"):
        suffix = s[-1] if s[-1] in "ui" else ""
        v = int(s[2:len(s) - len(suffix)], 16)
        return u32(v) if suffix == "u" else (i32(v) if suffix == "i" else v)
    suffix = s[-1] if s[-1] in "fuih" else ""
    body = s[:len(s) - len(suffix)]
    if suffix == "u":
        return u32(int(body))
    if suffix == "i":
        return i32(int(body))
    if suffix == "f":
        return F32(float(body))
    if any(ch in body for ch in ".eE"):
        return float(body)
    return int(body)


# ----------------------------------------------------------------------------------------------------------------------
# memory layout (WGSL "Alignment and Size") and buffer access
# ----------------------------------------------------------------------------------------------------------------------
def round_up(a, n):
    return (n + a - 1) // a * a


class Layout:
    """align / size of a type and decode / encode against a bytearray."""

    def __init__(self, module):
        self.m = module

    def align_size(self, ty):
        _, name, args = ty
        if name in ("f32", "u32", "i32"):
            return 4, 4
        if name == "atomic":
            return 4, 4
        if name in ("vec2", "vec3", "vec4"):
            n = int(name[3])
            return {2: 8, 3: 16, 4: 16}[n], 4 * n
        if name.startswith("mat"):
            c, r = int(name[3]), int(name[5])
            a, _ = self.align_size(("T", "vec%d" % r, [("T", "f32", [])]))
            return a, c * round_up(a, 4 * r)
        if name == "array":
            a, s = self.align_size(args[0])
            stride = round_up(a, s)
            if len(args) > 1:
                return a, stride * int(self.m.const_eval(args[1][1]))
            return a, stride  # runtime-sized: size of one element
        if name in self.m.structs:
            return self.struct_layout(name)[:2]
        raise TypeError("no layout for " + name)

    def struct_layout(self, name):
        off = 0
        align = 1
        members = []
        for fname, fty, fattrs in self.m.structs[name]:
            a, s = self.align_size(fty)
            if "align" in fattrs:
                a = int(self.m.const_eval(fattrs["align"][0]))
            if "size" in fattrs:
                s = int(self.m.const_eval(fattrs["size"][0]))
            off = round_up(a, off)
            members.append((fname, fty, off))
            off += s
            align = max(align, a)
        return align, round_up(align, off), members

    def decode(self, ty, buf, off):
        _, name, args = ty
        if name == "f32":
            return F32(struct.unpack_from("<f", buf, off)[0])
        if name == "u32" or (name == "atomic" and args[0][1] == "u32"):
            return u32(struct.unpack_from("<I", buf, off)[0])
        if name == "i32" or name == "atomic":
            return i32(struct.unpack_from("<i", buf, off)[0])
        if name in ("vec2", "vec3", "vec4"):
            return Vec([self.decode(args[0], buf, off + 4 * k) for k in range(int(name[3]))])
        if name.startswith("mat"):
            c, r = int(name[3]), int(name[5])
            a, _ = self.align_size(("T", "vec%d" % r, [("T", "f32", [])]))
            stride = round_up(a, 4 * r)
            return Mat([Vec([self.decode(("T", "f32", []), buf, off + stride * j + 4 * k) for k in range(r)])
                        for j in range(c)])
        if name == "array":
            a, s = self.align_size(args[0])
            stride = round_up(a, s)
            n = int(self.m.const_eval(args[1][1]))
            return [self.decode(args[0], buf, off + stride * k) for k in range(n)]
        if name in self.m.structs:
            _, _, members = self.struct_layout(name)
            return StructVal(name, {fname: self.decode(fty, buf, off + foff) for fname, fty, foff in members})
        raise TypeError("cannot decode " + name)

    def encode(self, ty, buf, off, v):
        _, name, args = ty
        if name == "f32":
            struct.pack_into("<f", buf, off, float(to_f32(v)))
        elif name in ("u32", "atomic") and (name == "u32" or args[0][1] == "u32"):
            struct.pack_into("<I", buf, off, int(v) & 0xFFFFFFFF)
        elif name in ("i32", "atomic"):
            struct.pack_into("<i", buf, off, int(i32(v)))
        elif name in ("vec2", "vec3", "vec4"):
            for k in range(int(name[3])):
                self.encode(args[0], buf, off + 4 * k, v.c[k])
        elif name == "array":
            a, s = self.align_size(args[0])
            stride = round_up(a, s)
            for k, x in enumerate(v):
                self.encode(args[0], buf, off + stride * k, x)
        elif name in self.m.structs:
            _, _, members = self.struct_layout(name)
            for fname, fty, foff in members:
                self.encode(fty, buf, off + foff, v.f[fname])
        else:
            raise TypeError("cannot encode " + name)


class Ref:
    """A reference into a bound buffer: (bytearray, type, byte offset).  Loads decode, stores encode."""

    def __init__(self, layout, buf, ty, off):
        self.layout, self.buf, self.ty, self.off = layout, buf, ty, off

    def load(self):
        if self.ty[1] == "array" and len(self.ty[2]) == 1:
            raise TypeError("loading a runtime-sized array")
        return self.layout.decode(self.ty, self.buf, self.off)

    def store(self, v):
        self.layout.encode(self.ty, self.buf, self.off, v)

    def member(self, name):
        _, _, members = self.layout.struct_layout(self.ty[1])
        for fname, fty, foff in members:
            if fname == name:
                return Ref(self.layout, self.buf, fty, self.off + foff)
        raise KeyError(name)

    def index(self, i):
        i = int(i)
        _, name, args = self.ty
        if name == "array":
            a, s = self.layout.align_size(args[0])
            stride = round_up(a, s)
            n = (len(self.buf) - self.off) // stride if len(args) == 1 else int(self.layout.m.const_eval(args[1][1]))
            if not 0 <= i < n:
                if self.layout.m.robust:
                    return OobRef(self.layout, args[0])
                raise IndexError("buffer index %d out of range (%d)" % (i, n))
            return Ref(self.layout, self.buf, args[0], self.off + stride * i)
        raise TypeError("indexing a reference to " + name)

    def array_length(self):
        a, s = self.layout.align_size(self.ty[2][0])
        return (len(self.buf) - self.off) // round_up(a, s)


class OobRef:
    """Out-of-bounds element under WebGPU's robust buffer access: loads give zero, stores are dropped."""

    def __init__(self, layout, ty):
        self.layout, self.ty = layout, ty

    def load(self):
        return self.layout.m.zero(self.ty if self.ty[1] != "atomic" else self.ty[2][0])

    def store(self, v):
        pass

    def member(self, name):
        _, _, members = self.layout.struct_layout(self.ty[1])
        for fname, fty, _ in members:
            if fname == name:
                return OobRef(self.layout, fty)
        raise KeyError(name)

    def index(self, i):
        return OobRef(self.layout, self.ty[2][0])


class LocalRef:
    """A reference to a function-scope variable (or a part of it): getter / setter closures."""

    def __init__(self, get, set_):
        self.load, self.store = get, set_


# ----------------------------------------------------------------------------------------------------------------------
# interpreter
# ----------------------------------------------------------------------------------------------------------------------
SWIZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


class Module:
    def __init__(self, source):
        self.decls = Parser(source).module()
        self.structs = {}
        self.fns = {}
        self.consts = {}
        self.vars = {}        # name -> (type, address space list)
        self.bindings = {}    # name -> bytearray
        self.wg = {}          # var<workgroup>: name -> value shared by the invocations of the running workgroup
        import threading
        self.tls = threading.local()   # .private: var<private> of the invocation this thread runs
        self.robust = False   # WebGPU robust buffer access: out-of-bounds loads give 0, stores are dropped
        self.barrier = None   # callable run by workgroupBarrier() (set by dispatch_workgroups)
        self.atomic_lock = None
        self.layout = Layout(self)
        for d in self.decls:
            if d[0] == "struct":
                self.structs[d[1]] = d[2]
        for d in self.decls:
            if d[0] == "const":
                v = self.eval(d[3], [{}])
                if d[2] is not None:
                    v = self.construct(d[2], [v]) if d[2][1] != "array" else v
                self.consts[d[1]] = v
            elif d[0] == "var":
                self.vars[d[1]] = (d[2], d[3])
            elif d[0] == "fn":
                self.fns[d[1]] = d

    def const_eval(self, e):
        return self.eval(e, [{}])

    def bind(self, name, data):
        """Bind a module-scope buffer variable to bytes (a bytearray is used in place so stores are visible)."""
        if name not in self.vars:
            raise KeyError("no such binding: " + name)
        self.bindings[name] = data if isinstance(data, bytearray) else bytearray(data)
        return self.bindings[name]

    def struct_size(self, name):
        return self.layout.struct_layout(name)[1]

    # -- construction / conversion
    def zero(self, ty):
        _, name, args = ty
        if name == "f32":
            return F32(0)
        if name == "u32":
            return u32(0)
        if name == "i32":
            return i32(0)
        if name == "bool":
            return False
        if name == "atomic":
            return self.zero(args[0])
        if name in ("vec2", "vec3", "vec4"):
            return Vec([self.zero(args[0]) for _ in range(int(name[3]))])
        if name.startswith("mat"):
            return Mat([Vec([F32(0)] * int(name[5])) for _ in range(int(name[3]))])
        if name == "array":
            return [self.zero(args[0]) for _ in range(int(self.const_eval(args[1][1])))]
        if name in self.structs:
            return StructVal(name, {f: self.zero(t) for f, t, _ in self.structs[name]})
        raise TypeError("zero of " + name)

    def convert_scalar(self, name, v):
        if name == "f32":
            return to_f32(v)
        if name == "u32":
            return to_u32(v)
        if name == "i32":
            return to_i32(v)
        if name == "bool":
            return to_bool(v)
        raise TypeError(name)

    def construct(self, ty, args):
        _, name, targs = ty
        if name in ("f32", "u32", "i32", "bool"):
            if not args:
                return self.zero(ty)
            return self.convert_scalar(name, args[0])
        if name in ("vec2", "vec3", "vec4"):
            n = int(name[3])
            el = targs[0][1] if targs else None
            flat = []
            for a in args:
                flat.extend(a.c if isinstance(a, Vec) else [a])
            if not flat:
                return self.zero(ty)
            if len(flat) == 1 and len(args) == 1 and not isinstance(args[0], Vec):
                flat = flat * n
            if len(flat) != n:
                raise TypeError("%s from %d components" % (name, len(flat)))
            if el:
                flat = [self.materialize(el, x) for x in flat]
            return Vec(flat)
        if name.startswith("mat"):
            c, r = int(name[3]), int(name[5])
            if all(isinstance(a, Vec) for a in args) and len(args) == c:
                return Mat([Vec([to_f32(x) for x in a.c]) for a in args])
            if len(args) == c * r:
                vals = [to_f32(x) for x in args]
                return Mat([Vec(vals[j * r:(j + 1) * r]) for j in range(c)])
            raise TypeError("matrix constructor")
        if name == "array":
            return [self.materialize_type(targs[0], a) for a in args]
        if name in self.structs:
            fields = self.structs[name]
            if not args:
                return self.zero(ty)
            if len(args) != len(fields):
                raise TypeError("struct constructor arity")
            return StructVal(name, {f: self.materialize_type(t, a) for (f, t, _), a in zip(fields, args)})
        raise TypeError("constructor " + name)

    def materialize(self, el, x):
        """Give an abstract numeric the concrete element type; concrete values must already match."""
        if is_abstract(x):
            return self.convert_scalar(el, x)
        want = {"f32": F32, "u32": U32, "i32": I32, "bool": bool}[el]
        if type(x) is not want:
            raise TypeError("component of type %s in a %s constructor" % (type(x).__name__, el))
        return x

    def materialize_type(self, ty, v):
        _, name, targs = ty
        if name in ("f32", "u32", "i32", "bool"):
            return self.materialize(name, v)
        if name in ("vec2", "vec3", "vec4") and isinstance(v, Vec):
            return Vec([self.materialize(targs[0][1], x) for x in v.c])
        return copy_val(v)

    # -- lvalues
    def ref(self, e, env):
        kind = e[0]
        if kind == "id":
            name = e[1]
            for scope in reversed(env):
                if name in scope:
                    def get(scope=scope, name=name):
                        return scope[name]

                    def set_(v, scope=scope, name=name):
                        scope[name] = v
                    return LocalRef(get, set_)
            priv = getattr(self.tls, "private", None)
            if priv is not None and name in priv:
                def getp(priv=priv, name=name):
                    return priv[name]

                def setp(v, priv=priv, name=name):
                    priv[name] = v
                return LocalRef(getp, setp)
            if name in self.wg:
                def getw(name=name):
                    return self.wg[name]

                def setw(v, name=name):
                    self.wg[name] = v
                return LocalRef(getw, setw)
            if name in self.bindings:
                return Ref(self.layout, self.bindings[name], self.vars[name][0], 0)
            raise NameError(name)
        if kind == "member":
            base = self.ref(e[1], env)
            name = e[2]
            if isinstance(base, OobRef):
                return base.member(name)
            if isinstance(base, Ref):
                if base.ty[1] in self.structs:
                    return base.member(name)
                if base.ty[1] in ("vec2", "vec3", "vec4") and len(name) == 1:
                    return Ref(self.layout, base.buf, base.ty[2][0], base.off + 4 * SWIZ[name])
                if base.ty[1] in ("vec2", "vec3", "vec4"):  # multi-component swizzle: a value, not a reference

                    def no_store(v):
                        raise TypeError("store through a swizzle")
                    return LocalRef(lambda: self.member(base.load(), name), no_store)
                raise TypeError("member of " + base.ty[1])

            def get():
                return self.member(base.load(), name)

            def set_(v):
                obj = base.load()
                if isinstance(obj, StructVal):
                    obj.f[name] = v
                elif isinstance(obj, Vec) and len(name) == 1:
                    obj.c[SWIZ[name]] = v
                else:
                    raise TypeError("cannot assign member " + name)
                base.store(obj)
            return LocalRef(get, set_)
        if kind == "index":
            base = self.ref(e[1], env)
            idx = int(self.eval(e[2], env))
            if isinstance(base, OobRef):
                return base.index(idx)
            if isinstance(base, Ref):
                if base.ty[1] == "array":
                    return base.index(idx)
                if base.ty[1].startswith("mat"):
                    r = int(base.ty[1][5])
                    a, _ = self.layout.align_size(("T", "vec%d" % r, [("T", "f32", [])]))
                    return Ref(self.layout, base.buf, ("T", "vec%d" % r, [("T", "f32", [])]),
                               base.off + round_up(a, 4 * r) * idx)
                if base.ty[1] in ("vec2", "vec3", "vec4"):
                    return Ref(self.layout, base.buf, base.ty[2][0], base.off + 4 * idx)
                raise TypeError("index of " + base.ty[1])

            def get():
                return self.index(base.load(), idx)

            def set_(v):
                obj = base.load()
                if isinstance(obj, list):
                    obj[idx] = v
                elif isinstance(obj, Vec):
                    obj.c[idx] = v
                elif isinstance(obj, Mat):
                    obj.cols[idx] = v
                base.store(obj)
            return LocalRef(get, set_)
        if kind == "un" and e[1] == "*":
            return self.eval(e[2], env)
        raise TypeError("not a reference expression: %r" % (e,))

    @staticmethod
    def member(obj, name):
        if isinstance(obj, StructVal):
            return obj.f[name]
        if isinstance(obj, Vec):
            if len(name) == 1:
                return obj.c[SWIZ[name]]
            return Vec([obj.c[SWIZ[ch]] for ch in name])
        raise TypeError("member %s of %r" % (name, type(obj)))

    @staticmethod
    def index(obj, idx):
        if isinstance(obj, list):
            if not 0 <= idx < len(obj):
                raise IndexError("array index %d out of range" % idx)
            return obj[idx]
        if isinstance(obj, Vec):
            return obj.c[idx]
        if isinstance(obj, Mat):
            return obj.cols[idx]
        raise TypeError("index of %r" % type(obj))

    # -- expressions
    def eval(self, e, env):
        kind = e[0]
        if kind == "num":
            return e[1]
        if kind == "id":
            name = e[1]
            for scope in reversed(env):
                if name in scope:
                    return scope[name]
            if name in self.consts:
                return self.consts[name]
            priv = getattr(self.tls, "private", None)
            if priv is not None and name in priv:
                return priv[name]
            if name in self.wg:
                return self.wg[name]
            if name in self.bindings:
                return Ref(self.layout, self.bindings[name], self.vars[name][0], 0).load()
            raise NameError(name)
        if kind == "bin":
            op = e[1]
            if op == "&&":
                return bool(self.eval(e[2], env)) and bool(self.eval(e[3], env))
            if op == "||":
                return bool(self.eval(e[2], env)) or bool(self.eval(e[3], env))
            return binop(op, self.eval(e[2], env), self.eval(e[3], env))
        if kind == "un":
            if e[1] == "&":
                return self.ref(e[2], env)
            if e[1] == "*":
                return self.eval(e[2], env).load()
            return unary(e[1], self.eval(e[2], env))
        if kind == "member":
            if self._rooted_in_buffer(e, env):
                return self.ref(e, env).load()
            return self.member(self.eval(e[1], env), e[2])
        if kind == "index":
            if self._rooted_in_buffer(e, env):
                return self.ref(e, env).load()
            return self.index(self.eval(e[1], env), int(self.eval(e[2], env)))
        if kind == "type":
            raise TypeError("type used as a value")
        if kind == "call":
            return self.call(e[1], e[2], env)
        raise TypeError("cannot evaluate %r" % (e,))

    def _rooted_in_buffer(self, e, env):
        """Member / index chains that start at a bound buffer are resolved by address (no whole-buffer decode)."""
        while e[0] in ("member", "index"):
            e = e[1]
        if e[0] != "id" or e[1] not in self.bindings:
            return False
        return not any(e[1] in scope for scope in env)

    def call(self, callee, arg_exprs, env):
        if callee[0] == "type":
            ty = callee[1]
            if ty[1] == "bitcast":
                v = self.eval(arg_exprs[0], env)
                return self.bitcast(ty[2][0][1], v)
            return self.construct(ty, [self.eval(a, env) for a in arg_exprs])
        if callee[0] != "id":
            raise TypeError("call of a non-identifier")
        name = callee[1]
        if name in self.structs or name in ("f32", "u32", "i32", "bool"):
            return self.construct(("T", name, []), [self.eval(a, env) for a in arg_exprs])
        if name in ("vec2", "vec3", "vec4"):  # element type inferred from the arguments
            return self.construct(("T", name, []), [self.eval(a, env) for a in arg_exprs])
        if name in self.fns:
            return self.invoke(name, [self.eval(a, env) for a in arg_exprs])
        args = [self.eval(a, env) for a in arg_exprs]
        return self.builtin(name, args)

    @staticmethod
    def bitcast(to, v):
        if isinstance(v, Vec):
            return Vec([Module.bitcast(to, x) for x in v.c])
        if isinstance(v, F32):
            bits = struct.unpack("<I", struct.pack("<f", float(v)))[0]
        else:
            bits = int(v) & 0xFFFFFFFF
        if to == "u32":
            return u32(bits)
        if to == "i32":
            return i32(bits)
        if to == "f32":
            return F32(struct.unpack("<f", struct.pack("<I", bits))[0])
        raise TypeError("bitcast to " + to)

    def builtin(self, name, a):
        def cw(fn, *vs):  # component-wise over vectors, scalars broadcast
            if any(isinstance(v, Vec) for v in vs):
                n = max(len(v) for v in vs if isinstance(v, Vec))
                return Vec([fn(*[(v.c[k] if isinstance(v, Vec) else v) for v in vs]) for k in range(n)])
            return fn(*vs)

        def f1(np_fn):
            def g(x):
                with np.errstate(all="ignore"):
                    return F32(np_fn(to_f32(x)))
            return g

        def fmax(x, y):
            x, y = concretize_like(x, y), concretize_like(y, x)
            return y if x < y else x

        def fmin(x, y):
            x, y = concretize_like(x, y), concretize_like(y, x)
            return y if y < x else x

        if name == "unpack2x16float":
            w = int(a[0])
            return Vec([f16_bits_to_f32(w & 0xFFFF), f16_bits_to_f32(w >> 16)])
        if name == "pack2x16float":
            v = a[0]
            return u32(f32_to_f16_bits(v.c[0]) | (f32_to_f16_bits(v.c[1]) << 16))
        if name == "unpack4x8snorm":
            w = int(a[0])
            out = []
            for k in range(4):
                b = (w >> (8 * k)) & 0xFF
                b = b - 256 if b & 0x80 else b
                out.append(fmax(scalar_binop("/", F32(b), F32(127.0)), F32(-1.0)))
            return Vec(out)
        if name == "extractBits":
            e, off, cnt = a[0], int(a[1]), int(a[2])
            off = min(off, 32)
            cnt = min(cnt, 32 - off)
            if cnt == 0:
                return i32(0) if isinstance(e, I32) else u32(0)
            bits = (int(e) & 0xFFFFFFFF) >> off & ((1 << cnt) - 1)
            if isinstance(e, I32):
                if bits & (1 << (cnt - 1)):
                    bits -= 1 << cnt
                return i32(bits)
            return u32(bits)
        if name == "any":
            return any(a[0].c) if isinstance(a[0], Vec) else bool(a[0])
        if name == "all":
            return all(a[0].c) if isinstance(a[0], Vec) else bool(a[0])
        if name == "dot":
            return dot(a[0], a[1])
        if name == "length":
            return f1(np.sqrt)(dot(a[0], a[0])) if isinstance(a[0], Vec) else cw(lambda x: F32(abs(x)), a[0])
        if name == "distance":
            d = binop("-", a[0], a[1])
            return f1(np.sqrt)(dot(d, d)) if isinstance(d, Vec) else F32(abs(d))
        if name == "normalize":
            return binop("/", a[0], f1(np.sqrt)(dot(a[0], a[0])))
        if name == "transpose":
            m = a[0]
            return Mat([Vec([m.cols[c].c[r] for c in range(len(m.cols))]) for r in range(len(m.cols[0]))])
        if name == "max":
            return cw(fmax, a[0], a[1])
        if name == "min":
            return cw(fmin, a[0], a[1])
        if name == "clamp":
            return cw(lambda x, lo, hi: fmin(fmax(x, lo), hi), a[0], a[1], a[2])
        if name == "abs":
            return cw(lambda x: F32(abs(x)) if isinstance(x, F32) else type(x)(abs(int(x))), a[0])
        if name == "sqrt":
            return cw(f1(np.sqrt), a[0])
        if name == "exp":
            return cw(f1(np.exp), a[0])
        if name == "log":
            return cw(f1(np.log), a[0])
        if name == "floor":
            return cw(f1(np.floor), a[0])
        if name == "select":
            return cw(lambda f, t, c: t if c else f, a[0], a[1], a[2])
        if name == "smoothstep":
            def ss(lo, hi, x):
                lo, hi, x = to_f32(lo), to_f32(hi), to_f32(x)
                t = scalar_binop("/", scalar_binop("-", x, lo), scalar_binop("-", hi, lo))
                t = fmin(fmax(t, F32(0.0)), F32(1.0))
                return scalar_binop("*", scalar_binop("*", t, t),
                                    scalar_binop("-", F32(3.0), scalar_binop("*", F32(2.0), t)))
            return cw(ss, a[0], a[1], a[2])
        if name in ("atomicAdd", "atomicLoad", "atomicStore"):
            lock = self.atomic_lock
            if lock is not None:
                lock.acquire()
            try:
                if name == "atomicLoad":
                    return a[0].load()
                if name == "atomicStore":
                    a[0].store(concretize_like(a[1], a[0].load()))
                    return None
                old = a[0].load()
                a[0].store(scalar_binop("+", old, concretize_like(a[1], old)))
                return old
            finally:
                if lock is not None:
                    lock.release()
        if name in ("workgroupBarrier", "storageBarrier"):
            if self.barrier is None:
                raise RuntimeError("workgroupBarrier() outside dispatch_workgroups(threads=True)")
            self.barrier()
            return None
        if name == "arrayLength":
            return u32(a[0].array_length())
        raise NameError("built-in not implemented: " + name)

    # -- statements / functions
    def invoke(self, name, args, builtins=None):
        _, _, params, ret, body, _ = self.fns[name]
        scope = {}
        for (pname, pty, pattrs), v in zip(params, args):
            scope[pname] = self.materialize_type(pty, v)
        try:
            self.run(body, [scope])
        except _Return as r:
            v = r.v
            return self.materialize_type(ret, v) if ret is not None and v is not None else v
        return None

    def run(self, stmts, env):
        self.run_here(stmts, env + [{}])

    def run_here(self, stmts, env):
        for s in stmts:
            kind = s[0]
            if kind in ("let", "var"):
                _, name, ty, e = s
                if e is None:
                    v = self.zero(ty)
                else:
                    v = copy_val(self.eval(e, env))
                    if ty is not None:
                        v = self.materialize_type(ty, v)
                    elif type(v) is float:  # a declaration without a type gives abstract values their default type
                        v = F32(v)
                    elif type(v) is int:
                        v = i32(v)
                env[-1][name] = v
            elif kind == "assign":
                _, op, lhs, rhs = s
                r = self.ref(lhs, env)
                v = self.eval(rhs, env)
                if op != "=":
                    v = binop(op[:-1], r.load(), v)
                else:
                    cur = None
                    if isinstance(r, LocalRef):
                        cur = r.load()
                    if is_abstract(v) and cur is not None:
                        v = concretize_like(v, cur)
                r.store(copy_val(v))
            elif kind == "expr":
                self.eval(s[1], env)
            elif kind == "if":
                _, cond, then, els = s
                if bool(self.eval(cond, env)):
                    self.run(then, env)
                elif els is not None:
                    self.run(els, env)
            elif kind == "block":
                self.run(s[1], env)
            elif kind == "return":
                raise _Return(None if s[1] is None else self.eval(s[1], env))
            elif kind == "discard":
                raise Discard()
            elif kind == "break":
                raise _Break()
            elif kind == "continue":
                raise _Continue()
            elif kind == "loop":
                _, init, cond, step, body = s
                lenv = env + [{}]
                if init is not None:
                    self.run_here([init], lenv)
                while cond is None or bool(self.eval(cond, lenv)):
                    try:
                        self.run(body, lenv)
                    except _Break:
                        break
                    except _Continue:
                        pass
                    if step is not None:
                        self.run_here([step], lenv)
            else:
                raise TypeError("statement " + kind)

    def dispatch_workgroups(self, entry, num_workgroups, threads=True):
        """Run a compute entry point whose workgroups cooperate: workgroups one after the other in workgroup-id order (a
        workgroup that looks back at its predecessors finds them finished), the invocations of a workgroup as threads that
        meet at workgroupBarrier(); var<workgroup> is zero-initialised per workgroup, var<private> per invocation; atomics
        are serialised.  threads=False runs the invocations one after the other (entry points without barriers)."""
        import threading
        _, _, params, _, body, attrs = self.fns[entry]
        wg = int(self.const_eval(attrs["workgroup_size"][0]))
        self.atomic_lock = threading.Lock() if threads else None
        errors = []

        def invocation(w, l, barrier):
            scope = {}
            self.tls.private = {vname: self.zero(vty) for vname, (vty, space) in self.vars.items() if "private" in space}
            for pname, pty, pattrs in params:
                which = pattrs["builtin"][0][1]
                scope[pname] = {"global_invocation_id": Vec([u32(w * wg + l), u32(0), u32(0)]),
                                "local_invocation_id": Vec([u32(l), u32(0), u32(0)]),
                                "workgroup_id": Vec([u32(w), u32(0), u32(0)]),
                                "num_workgroups": Vec([u32(num_workgroups), u32(1), u32(1)])}[which]
            try:
                self.run(body, [scope])
            except _Return:
                pass
            except BaseException as e:  # noqa: BLE001  (reported by the dispatching thread)
                errors.append(e)
                if barrier is not None:
                    barrier.abort()

        for w in range(num_workgroups):
            self.wg = {vname: self.zero(vty) for vname, (vty, space) in self.vars.items() if "workgroup" in space}
            if threads:
                barrier = threading.Barrier(wg)
                self.barrier = barrier.wait
                ts = [threading.Thread(target=invocation, args=(w, l, barrier)) for l in range(wg)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
            else:
                self.barrier = None
                for l in range(wg):
                    invocation(w, l, None)
            if errors:
                raise errors[0]
        self.barrier = None
        self.atomic_lock = None

    def dispatch(self, entry, num_workgroups, after_invocation=None):
        """Run a compute entry point: workgroups x workgroup_size invocations in global-invocation order.
        after_invocation(global_id) is called after each one (the generator uses it to see who stored what)."""
        _, _, params, _, body, attrs = self.fns[entry]
        wg = int(self.const_eval(attrs["workgroup_size"][0]))
        for g in range(num_workgroups * wg):
            scope = {}
            for pname, pty, pattrs in params:
                b = pattrs.get("builtin")
                which = b[0][1] if b else None
                if which == "global_invocation_id":
                    scope[pname] = Vec([u32(g), u32(0), u32(0)])
                elif which == "num_workgroups":
                    scope[pname] = Vec([u32(num_workgroups), u32(1), u32(1)])
                elif which == "local_invocation_id":
                    scope[pname] = Vec([u32(g % wg), u32(0), u32(0)])
                elif which == "workgroup_id":
                    scope[pname] = Vec([u32(g // wg), u32(0), u32(0)])
                else:
                    raise TypeError("entry-point parameter " + pname)
            try:
                self.run(body, [scope])
            except _Return:
                pass
            if after_invocation is not None:
                after_invocation(g)
