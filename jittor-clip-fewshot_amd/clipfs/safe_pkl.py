"""Inert reader for the pickle files the reference writes with ``jt.save``.

``jt.save`` (lora_train_vlp.py:592, jclip/clip.py:182 reads the same format)
stores a plain pickle whose only non-primitive objects are numpy arrays.  Running
such a file through ``pickle.load`` would execute whatever callables the file
names, so this module does NOT use the pickle VM at all: it walks the opcode
stream with ``pickletools.genops`` and rebuilds dict / list / tuple / str / int /
float / bool / None / bytes and numpy arrays itself.  The three numpy globals a
numpy array pickle references are mapped to local inert constructors; any other
global, and any opcode outside the small set below, raises ``UnsafePickleError``.
Nothing from the file is ever called.
"""
from __future__ import annotations

import pickletools
from typing import Any

import numpy as np


class UnsafePickleError(ValueError):
    pass


class _Global:
    def __init__(self, module: str, name: str):
        self.key = (module, name)


class _PendingArray:
    """Result of ``numpy.core.multiarray._reconstruct(ndarray, (0,), b'b')``."""


class _DType:
    def __init__(self, code: str):
        self.code = code
        self.byteorder = "="


_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
}
_MARK = object()


def _reduce(fn: Any, args: tuple):
    if not isinstance(fn, _Global):
        raise UnsafePickleError("REDUCE on a non-whitelisted callable")
    mod, name = fn.key
    if name == "_reconstruct":
        return _PendingArray()
    if (mod, name) == ("numpy", "dtype"):
        code = args[0]
        if not isinstance(code, str) or len(code) > 4:
            raise UnsafePickleError(f"unsupported dtype spec {code!r}")
        return _DType(code)
    raise UnsafePickleError(f"REDUCE on {mod}.{name} not allowed")


def _build(obj: Any, state: Any):
    if isinstance(obj, _DType):
        # (version, byteorder, subarray, names, fields, elsize, alignment, flags)
        if state[2] is not None or state[3] is not None or state[4] is not None:
            raise UnsafePickleError("structured dtypes not supported")
        obj.byteorder = state[1]
        return obj
    if isinstance(obj, _PendingArray):
        # (version, shape, dtype, is_fortran, rawdata)
        _, shape, dt, fortran, raw = state
        if not isinstance(dt, _DType) or not isinstance(raw, (bytes, bytearray)):
            raise UnsafePickleError("malformed ndarray state")
        bo = dt.byteorder if dt.byteorder in "<>" else "="
        npdt = np.dtype(dt.code).newbyteorder(bo) if dt.code[0] not in "<>|=" else np.dtype(dt.code)
        if npdt.hasobject:
            raise UnsafePickleError("object arrays not supported")
        arr = np.frombuffer(bytes(raw), dtype=npdt)
        arr = arr.reshape(tuple(shape), order="F" if fortran else "C")
        return np.array(arr)  # own, writable copy
    raise UnsafePickleError("BUILD on unsupported object")


def loads(data: bytes) -> Any:
    stack: list = []
    memo: dict = {}
    next_memo = 0

    def pop_mark() -> list:
        for i in range(len(stack) - 1, -1, -1):
            if stack[i] is _MARK:
                items = stack[i + 1:]
                del stack[i:]
                return items
        raise UnsafePickleError("MARK not found")

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            return stack.pop()
        elif n == "MARK":
            stack.append(_MARK)
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n in ("BININT", "BININT1", "BININT2", "LONG1", "LONG4", "BINFLOAT"):
            stack.append(arg)
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8"):
            stack.append(arg)
        elif n in ("SHORT_BINBYTES", "BINBYTES", "BINBYTES8", "BYTEARRAY8"):
            stack.append(bytes(arg))
        elif n == "MEMOIZE":
            memo[next_memo] = stack[-1]
            next_memo += 1
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
            next_memo = max(next_memo, int(arg) + 1)
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            items = pop_mark()
            d = stack[-1]
            for i in range(0, len(items), 2):
                d[items[i]] = items[i + 1]
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            items = pop_mark()
            stack[-1].extend(items)
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            mod = stack.pop()
            if (mod, name) not in _ALLOWED:
                raise UnsafePickleError(f"global {mod}.{name} is not on the whitelist")
            stack.append(_Global(mod, name))
        elif n == "GLOBAL":
            mod, name = arg.split(" ")
            if (mod, name) not in _ALLOWED:
                raise UnsafePickleError(f"global {mod}.{name} is not on the whitelist")
            stack.append(_Global(mod, name))
        elif n == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            stack.append(_reduce(fn, args))
        elif n == "BUILD":
            state = stack.pop()
            stack[-1] = _build(stack[-1], state)
        else:
            raise UnsafePickleError(f"opcode {n} not supported by the inert reader")
    raise UnsafePickleError("pickle stream ended without STOP")


def load(path: str) -> Any:
    with open(path, "rb") as f:
        return loads(f.read())
