"""Reader/writer for the `.prt` job files (text pickles, protocol 0).

The reference stores a job as six `cPickle` protocol-0 files
(reference IOOperation.py:19-31 reads them, :42-53 and RayTracerGui.py
write them).  Job files come from other people's machines, so this module
never hands them to `pickle`: it interprets the handful of protocol-0
opcodes those files use with a small stack machine of its own that can only
build ints, floats, bools, None, strings, lists, tuples, dicts and plain
numeric numpy arrays.  A GLOBAL opcode never imports anything; the three
numpy constructors that appear in `Elements.prt` are recognised by name and
replaced by inert markers that the REDUCE/BUILD handlers below understand.
Anything else raises `PrtError`.
"""
from __future__ import annotations

import codecs
import io
import os

import numpy as np


class PrtError(ValueError):
    """The file uses a pickle feature outside the job-file subset."""


class _Mark:
    pass


class _Global:
    __slots__ = ("module", "name")

    def __init__(self, module, name):
        self.module, self.name = module, name

    def key(self):
        return self.module + "." + self.name


class _PendingArray:
    """Result of `_reconstruct(ndarray, (0,), 'b')` waiting for its BUILD."""


class _PendingDtype:
    def __init__(self, code):
        self.code = code
        self.order = "="


_NUMPY_RECONSTRUCT = {"numpy.core.multiarray._reconstruct",
                      "numpy._core.multiarray._reconstruct"}
_NUMPY_NDARRAY = {"numpy.ndarray"}
_NUMPY_DTYPE = {"numpy.dtype"}
_CODECS_ENCODE = {"_codecs.encode"}
_ALLOWED_DTYPES = {"f8", "f4", "i8", "i4", "i2", "i1", "u8", "u4", "u2", "u1", "b1"}


def _unescape_py2_str(token: bytes) -> bytes:
    """`S'...'` payload: a Python-2 `repr` of a byte string."""
    token = token.strip()
    if len(token) < 2 or token[0:1] not in (b"'", b'"') or token[-1:] != token[0:1]:
        raise PrtError("malformed STRING opcode")
    return codecs.escape_decode(token[1:-1])[0]


def _parse_int(token: bytes):
    token = token.strip()
    if token == b"00":
        return False
    if token == b"01":
        return True
    return int(token)


def loads(data: bytes):
    """Decode one protocol-0 job file held in `data`."""
    stream = io.BytesIO(data)
    stack: list = []
    memo: dict = {}

    def readline() -> bytes:
        line = stream.readline()
        if not line.endswith(b"\n"):
            raise PrtError("truncated file")
        return line[:-1]

    def pop_to_mark() -> list:
        for i in range(len(stack) - 1, -1, -1):
            if isinstance(stack[i], _Mark):
                items = stack[i + 1:]
                del stack[i:]
                return items
        raise PrtError("MARK not found")

    while True:
        op = stream.read(1)
        if not op:
            raise PrtError("no STOP opcode")
        if op == b".":
            if len(stack) != 1:
                raise PrtError("stack not reduced to one object")
            return stack[0]
        elif op == b"(":
            stack.append(_Mark())
        elif op == b"l":
            stack.append(list(pop_to_mark()))
        elif op == b"t":
            stack.append(tuple(pop_to_mark()))
        elif op == b"d":
            items = pop_to_mark()
            stack.append(dict(zip(items[0::2], items[1::2])))
        elif op == b"]":
            stack.append([])
        elif op == b")":
            stack.append(())
        elif op == b"}":
            stack.append({})
        elif op == b"a":
            value = stack.pop()
            stack[-1].append(value)
        elif op == b"s":
            value = stack.pop()
            key = stack.pop()
            stack[-1][key] = value
        elif op == b"p":
            memo[readline()] = stack[-1]
        elif op == b"g":
            stack.append(memo[readline()])
        elif op == b"I":
            stack.append(_parse_int(readline()))
        elif op == b"L":
            stack.append(int(readline().strip().rstrip(b"L")))
        elif op == b"F":
            stack.append(float(readline()))
        elif op == b"N":
            stack.append(None)
        elif op == b"S":
            stack.append(_unescape_py2_str(readline()))
        elif op == b"V":
            stack.append(readline().decode("raw-unicode-escape"))
        elif op == b"c":
            stack.append(_Global(readline().decode("ascii"), readline().decode("ascii")))
        elif op == b"R":
            args = stack.pop()
            func = stack.pop()
            stack.append(_reduce(func, args))
        elif op == b"b":
            state = stack.pop()
            pending = stack[-1]
            built = _build(pending, state)
            stack[-1] = built
            # memo slots that held the pending marker must now see the result
            for key, value in memo.items():
                if value is pending:
                    memo[key] = built
        else:
            raise PrtError("opcode %r is outside the job-file subset" % op)


def _reduce(func, args):
    if not isinstance(func, _Global):
        raise PrtError("REDUCE on a non-global")
    key = func.key()
    if key in _NUMPY_RECONSTRUCT:
        if not (isinstance(args, tuple) and len(args) == 3 and isinstance(args[0], _Global)
                and args[0].key() in _NUMPY_NDARRAY):
            raise PrtError("unexpected arguments to numpy _reconstruct")
        return _PendingArray()
    if key in _NUMPY_DTYPE:
        code = args[0]
        if isinstance(code, bytes):
            code = code.decode("ascii")
        if code not in _ALLOWED_DTYPES:
            raise PrtError("dtype %r not allowed in a job file" % (code,))
        return _PendingDtype(code)
    if key in _CODECS_ENCODE:
        text, encoding = args
        if encoding != "latin1":
            raise PrtError("unexpected codec")
        return text.encode("latin1")
    raise PrtError("global %s is not allowed in a job file" % key)


def _build(obj, state):
    if isinstance(obj, _PendingDtype):
        # (version, byteorder, subdescr, names, fields, itemsize, alignment, flags)
        order = state[1]
        if isinstance(order, bytes):
            order = order.decode("ascii")
        if order not in ("<", ">", "=", "|"):
            raise PrtError("bad byte order")
        obj.order = order
        return obj
    if isinstance(obj, _PendingArray):
        _version, shape, dt, fortran, raw = state
        if not isinstance(dt, _PendingDtype):
            raise PrtError("array without a plain dtype")
        if isinstance(raw, str):
            raw = raw.encode("latin1")
        dtype = np.dtype(dt.order.replace("|", "=") + dt.code if dt.order != "=" else dt.code)
        arr = np.frombuffer(raw, dtype=dtype)
        if int(np.prod(shape)) != arr.size:
            raise PrtError("array payload does not match its shape")
        arr = arr.reshape(shape, order="F" if fortran else "C")
        return np.array(arr, dtype=dtype.newbyteorder("="))
    raise PrtError("BUILD on an unsupported object")


def load(path):
    with open(path, "rb") as fh:
        return loads(fh.read())


# ---------------------------------------------------------------------------
# writer: emits the same protocol-0 subset (arrays are written as nested lists,
# which is what the reference's `np.array(cPickle.load(f))` accepts)
# ---------------------------------------------------------------------------

def _emit(obj, out: list):
    if isinstance(obj, (bool, np.bool_)):
        out.append(b"I01\n" if obj else b"I00\n")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"I%d\n" % int(obj))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"F" + repr(float(obj)).encode("ascii") + b"\n")
    elif obj is None:
        out.append(b"N")
    elif isinstance(obj, str):
        out.append(b"V" + obj.encode("raw-unicode-escape") + b"\n")
    elif isinstance(obj, np.ndarray):
        _emit(obj.tolist(), out)
    elif isinstance(obj, (list, tuple)):
        out.append(b"(")
        for item in obj:
            _emit(item, out)
        out.append(b"l" if isinstance(obj, list) else b"t")
    elif isinstance(obj, dict):
        out.append(b"(")
        for k, v in obj.items():
            _emit(k, out)
            _emit(v, out)
        out.append(b"d")
    else:
        # material objects inside a parsed ElementProperties are saved by name,
        # the form ParsePath expects to read back
        name = getattr(obj, "name", None)
        if name is None:
            raise PrtError("cannot store %r in a job file" % type(obj))
        _emit(str(name), out)


def dumps(obj) -> bytes:
    out: list = []
    _emit(obj, out)
    out.append(b".")
    return b"".join(out)


def dump(obj, path):
    with open(path, "wb") as fh:
        fh.write(dumps(obj))


JOB_FILES = ("Elements", "ElementProperties", "ElementGeometry", "ElementSize",
             "Materials", "Photon")


def load_job(folder):
    """Read the six files of a job folder into a dict keyed by JOB_FILES."""
    return {name: load(os.path.join(folder, name + ".prt")) for name in JOB_FILES}


def dump_job(folder, job):
    os.makedirs(folder, exist_ok=True)
    for name in JOB_FILES:
        dump(job[name], os.path.join(folder, name + ".prt"))
