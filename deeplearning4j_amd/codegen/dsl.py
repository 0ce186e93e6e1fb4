"""The op-definition DSL (reference: codegen/op-codegen/src/main/kotlin/org/nd4j/codegen/dsl/OpBuilder.kt and api/*.kt).

    ns = Namespace("bitwise", doc="...")
    (ns.op("and", libnd4j="bitwise_and", doc="Bitwise AND")
       .input("x", DataType.INT, "first operand").input("y", DataType.INT, "second operand")
       .output("output", DataType.INT, "x & y"))

Arguments name the libnd4j argument vector they travel in (``slot``: ``i`` integer, ``t`` floating point, ``b`` boolean, ``d`` data type)
and are emitted in declaration order inside their slot; ``convert`` names a conversion applied in the generated call (``int`` for a
boolean carried as an integer argument, ``ord`` for a character code, ``dtype`` for a data-type name)."""
from __future__ import annotations

import keyword
from typing import Dict, List, Optional


class DataType:
    NUMERIC, FLOATING_POINT, INT, LONG, BOOL, DATA_TYPE, STRING, ENUM = "NUMERIC", "FLOATING_POINT", "INT", "LONG", "BOOL", "DATA_TYPE", "STRING", "ENUM"
    ALL = (NUMERIC, FLOATING_POINT, INT, LONG, BOOL, DATA_TYPE, STRING, ENUM)
    # which argument vector an Arg of this type uses unless told otherwise
    DEFAULT_SLOT = {INT: "i", LONG: "i", FLOATING_POINT: "t", NUMERIC: "t", BOOL: "b", DATA_TYPE: "d", ENUM: "i"}


class Count:
    """How many values an argument takes: Exactly(n), AtLeast(n) or Range(a, b) (api/Count.kt)."""

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    @staticmethod
    def Exactly(n):
        return Count(n, n)

    @staticmethod
    def AtLeast(n):
        return Count(n, None)

    @staticmethod
    def Range(a, b):
        return Count(a, b)

    def isScalar(self):
        return self.lo == 1 and self.hi == 1

    def describe(self):
        if self.lo == self.hi:
            return "exactly %d" % self.lo
        return "at least %d" % self.lo if self.hi is None else "%d to %d" % (self.lo, self.hi)


class _Param:
    def __init__(self, name, type_, doc):
        if type_ not in DataType.ALL:
            raise ValueError("unknown DSL data type %r" % type_)
        if not name.isidentifier():
            raise ValueError("parameter name %r is not an identifier" % name)
        self.name, self.type, self.doc = name, type_, doc

    def pyName(self):
        return self.name + "_" if keyword.iskeyword(self.name) else self.name


class Input(_Param):
    def __init__(self, name, type_, doc, count: Optional[Count] = None, optional=False, default=None):
        """``default``: scalar constant passed to the native op when an optional input is not given (the loss ops' weights = 1.0); without it
        an absent optional input is simply left out of the op's input list."""
        super().__init__(name, type_, doc)
        self.count, self.optional, self.default = count or Count.Exactly(1), optional, default
        if default is not None and not optional:
            raise ValueError("input %s: only an optional input can have a default" % name)


class Output(_Param):
    pass


_NO_DEFAULT = object()


class Arg(_Param):
    def __init__(self, name, type_, doc, default=_NO_DEFAULT, count: Optional[Count] = None, slot=None, convert=None, possibleValues=None,
                 pos: Optional[int] = None):
        """``pos``: position inside the argument vector when it differs from the declaration order (the signature lists required
        arguments before defaulted ones; a native op may want them the other way round: onehot takes [axis, depth])."""
        super().__init__(name, type_, doc)
        self.pos = pos
        self.default, self.count = default, count or Count.Exactly(1)
        self.slot = slot or DataType.DEFAULT_SLOT.get(type_)
        if self.slot not in ("i", "t", "b", "d", "dims", "scalar"):
            raise ValueError("argument %s: no argument vector for type %s (give slot=)" % (name, type_))
        if convert not in (None, "int", "ord", "dtype", "float"):
            raise ValueError("argument %s: unknown conversion %r" % (name, convert))
        self.convert = convert
        self.possibleValues = list(possibleValues) if possibleValues else None
        if type_ == DataType.ENUM and not self.possibleValues:
            raise ValueError("argument %s: an ENUM needs possibleValues" % name)

    def hasDefault(self):
        return self.default is not _NO_DEFAULT


class Op:
    def __init__(self, namespace: "Namespace", name, libnd4j=None, doc=""):
        self.namespace, self.name, self.libnd4j, self.doc = namespace, name, libnd4j or name, doc
        self.inputs: List[Input] = []
        self.args: List[Arg] = []
        self.outputs: List[Output] = []
        self.aliases: List[str] = []
        self.nOutExpr: Optional[str] = None
        self.order: Optional[List[str]] = None
        self.legacy = False

    # -- fluent definition
    def input(self, name, type_, doc, count=None, optional=False, default=None):
        self.inputs.append(Input(name, type_, doc, count, optional, default))
        return self

    def nativeOrder(self, *names):
        """Order in which the native op takes the inputs when it differs from the signature's (losses: predictions, weights, labels)."""
        self.order = list(names)
        return self

    def arg(self, name, type_, doc, default=_NO_DEFAULT, count=None, slot=None, convert=None, possibleValues=None, pos=None):
        self.args.append(Arg(name, type_, doc, default, count, slot, convert, possibleValues, pos))
        return self

    def output(self, name, type_, doc):
        self.outputs.append(Output(name, type_, doc))
        return self

    def alias(self, *names):
        self.aliases += list(names)
        return self

    def legacyOp(self, family: str, opName: str):
        """The op is one of the LEGACY families (transform_* / scalar / pairwise / reduce_* / reduce3 / index_reduce / summary_stats of
        libnd4j's loops, addressed by (family, opNum)) instead of a declarable custom op. Arguments may then use slot ``dims`` (the
        reduction dimensions, one variable-length argument) and slot ``scalar`` (the scalar operand of a scalar op); ``t`` arguments
        become the op's extra parameters."""
        self.legacy = (family, opName)
        self.libnd4j = opName
        return self

    def numOutputs(self, expr: str):
        """Python expression over the argument names giving the number of outputs when it depends on them."""
        self.nOutExpr = expr
        return self

    def pyName(self):
        return self.name + "_" if keyword.iskeyword(self.name) else self.name

    # -- validation (OpBuilder.kt checks + what the generators rely on)
    def validate(self):
        where = "%s.%s" % (self.namespace.name, self.name)
        if not self.outputs:
            raise ValueError("%s: an op needs at least one output" % where)
        if self.legacy:
            bad = [a.name for a in self.args if a.slot in ("i", "b", "d")]
            if bad:
                raise ValueError("%s: a legacy op takes t / dims / scalar arguments only (got %s)" % (where, bad))
        elif any(a.slot in ("dims", "scalar") for a in self.args):
            raise ValueError("%s: slots dims / scalar belong to legacy ops" % where)
        names = [p.name for p in self.inputs + self.args]
        if len(set(names)) != len(names):
            raise ValueError("%s: duplicate parameter names" % where)
        seen_default = False
        for a in self.args:
            if a.hasDefault():
                seen_default = True
            elif seen_default and a.count.isScalar():
                raise ValueError("%s: argument %s without default follows one with a default" % (where, a.name))
        varargs = [a for a in self.args if not a.count.isScalar()]
        if len(varargs) > 1:
            raise ValueError("%s: at most one variable-length argument is supported" % where)
        if self.order is not None:
            if sorted(self.order) != sorted(i.name for i in self.inputs):
                raise ValueError("%s: nativeOrder must name every input exactly once" % where)
            if any(not i.count.isScalar() or (i.optional and i.default is None) for i in self.inputs):
                raise ValueError("%s: nativeOrder needs single inputs, optional ones with a default" % where)
        opt = [i for i in self.inputs if i.optional]
        if opt and self.inputs.index(opt[0]) < len(self.inputs) - len(opt):
            raise ValueError("%s: optional inputs must come last" % where)
        return self


class Namespace:
    _ALL: "Dict[str, Namespace]" = {}

    def __init__(self, name, doc="", className=None):
        self.name, self.doc = name, doc
        self.className = className or name[0].upper() + name[1:]
        self.ops: List[Op] = []
        Namespace._ALL[name] = self

    def op(self, name, libnd4j=None, doc=""):
        o = Op(self, name, libnd4j, doc)
        self.ops.append(o)
        return o

    def validate(self):
        names = [o.name for o in self.ops] + [a for o in self.ops for a in o.aliases]
        if len(set(names)) != len(names):
            raise ValueError("namespace %s: duplicate op / alias names" % self.name)
        for o in self.ops:
            o.validate()
        return self


def all_namespaces() -> "Dict[str, Namespace]":
    """Load every definition module under ``codegen/ops`` and return the namespaces by name."""
    import importlib
    import pkgutil
    from . import ops as _ops
    for m in pkgutil.iter_modules(_ops.__path__):
        importlib.import_module("%s.%s" % (_ops.__name__, m.name))
    return dict(sorted(Namespace._ALL.items()))
