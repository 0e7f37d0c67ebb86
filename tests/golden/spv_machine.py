"""A small SPIR-V interpreter: enough of the instruction set to EXECUTE the shader modules the reference commits
under shaders/*.spv (glslang output, SPIR-V 1.0, logical addressing + PhysicalStorageBuffer references), one
invocation at a time, in IEEE binary32 with every operation rounded on its own.

Test infrastructure of ours (no reference text in here): tests/golden/make_spv_exec.py feeds it the reference's
modules inside the container and stores inputs + outputs as tests/golden/spv_exec.npz; the modules themselves never
leave /root/reference.

Two arithmetic modes, because the modules carry no NoContraction decoration and SPIR-V leaves the association of
OpDot / OpMatrixTimesVector / OpMatrixTimesMatrix to the implementation:

  strict   every multiply and every add rounds once; dot products and matrix products accumulate left to right
           (x, then y, then z, then w).
  fused    what a contracting compiler makes of the same module: dot / matrix products are a multiply followed by a
           chain of fused multiply-adds in the same order, OpFAdd whose first (else second) operand is the result of a
           multiply of this invocation becomes one fma, FMix(x, y, a) = fma(y, a, x * (1 - a)).

Neither is "the" Vulkan result (drivers differ); together they bracket it.  Values: scalars are numpy float32 /
int32 / uint32 / bool, vectors / matrices (columns) / arrays / structs are Python lists.
"""
import ctypes
import ctypes.util
import math
import struct

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fmaf.restype = ctypes.c_float
_libm.fmaf.argtypes = [ctypes.c_float] * 3

F32 = np.float32


def fmaf(a, b, c):
    return F32(_libm.fmaf(float(a), float(b), float(c)))


class Box:
    """storage of one OpVariable (or of a buffer a PhysicalStorageBuffer pointer refers to)"""

    def __init__(self, value=None):
        self.value = value


class Pointer:
    def __init__(self, box, path=()):
        self.box, self.path = box, tuple(path)


def _copy(v):
    return [_copy(x) for x in v] if isinstance(v, list) else v


def _string(ws):
    b = b"".join(struct.pack("<I", x) for x in ws)
    return b.split(b"\0", 1)[0].decode()


class Module:
    def __init__(self, path):
        data = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(data) // 4), data)
        if w[0] != 0x07230203:
            raise ValueError("not a SPIR-V module: " + path)
        self.version = (w[1] >> 16 & 0xff, w[1] >> 8 & 0xff)
        self.ins = []
        i = 5
        while i < len(w):
            n, op = w[i] >> 16, w[i] & 0xffff
            self.ins.append((op, list(w[i + 1:i + n])))
            i += n
        self.names, self.member_names, self.types, self.consts = {}, {}, {}, {}
        self.decor, self.member_decor = {}, {}
        self.globals = {}       # id -> (type id, storage class)
        self.functions = {}     # id -> dict(params, blocks, first)
        self.entry = None
        self.ext_sets = {}
        self.local_size = None
        self._scan()

    def _scan(self):
        T, fn, label = self.types, None, None
        for op, a in self.ins:
            if op == 5:
                self.names[a[0]] = _string(a[1:])
            elif op == 6:
                self.member_names.setdefault(a[0], {})[a[1]] = _string(a[2:])
            elif op == 11:
                self.ext_sets[a[0]] = _string(a[1:])
            elif op == 15:
                self.entry = a[1]
            elif op == 16 and a[1] == 17:
                self.local_size = tuple(a[2:5])
            elif op == 71:
                self.decor.setdefault(a[0], {})[a[1]] = a[2:]
            elif op == 72:
                self.member_decor.setdefault(a[0], {}).setdefault(a[1], {})[a[2]] = a[3:]
            elif op == 19:
                T[a[0]] = ("void",)
            elif op == 20:
                T[a[0]] = ("bool",)
            elif op == 21:
                T[a[0]] = ("int", a[1], a[2])
            elif op == 22:
                T[a[0]] = ("float", a[1])
            elif op == 23:
                T[a[0]] = ("vec", a[1], a[2])
            elif op == 24:
                T[a[0]] = ("mat", a[1], a[2])
            elif op == 25:
                T[a[0]] = ("image",)
            elif op == 27:
                T[a[0]] = ("sampledimage", a[1])
            elif op == 28:
                T[a[0]] = ("array", a[1], a[2])
            elif op == 29:
                T[a[0]] = ("rtarray", a[1])
            elif op == 30:
                T[a[0]] = ("struct", a[1:])
            elif op == 32:
                T[a[0]] = ("ptr", a[1], a[2])
            elif op == 33:
                T[a[0]] = ("func", a[1], a[2:])
            elif op == 39:
                pass  # OpTypeForwardPointer: the OpTypePointer follows
            elif op == 43:
                t = T[a[0]]
                if t[0] == "float":
                    self.consts[a[1]] = F32(struct.unpack("<f", struct.pack("<I", a[2]))[0])
                elif t[0] == "int":
                    self.consts[a[1]] = np.int32(struct.unpack("<i", struct.pack("<I", a[2]))[0]) if t[2] else np.uint32(a[2])
                else:
                    raise NotImplementedError("OpConstant of type %r" % (t,))
            elif op == 44:
                self.consts[a[1]] = [self.consts[x] for x in a[2:]]
            elif op in (41, 42):
                self.consts[a[1]] = op == 41
            elif op == 59 and fn is None:
                self.globals[a[1]] = (a[0], a[2])
            elif op == 54:
                fn = {"params": [], "blocks": {}, "first": None, "result_type": a[0]}
                self.functions[a[1]] = fn
            elif op == 55:
                fn["params"].append(a[1])
            elif op == 248:
                label = a[0]
                fn["blocks"][label] = []
                if fn["first"] is None:
                    fn["first"] = label
            elif op == 56:
                fn, label = None, None
            elif fn is not None and label is not None:
                fn["blocks"][label].append((op, a))

    # ---- reflection helpers the harness uses
    def global_by_name(self, name):
        for gid in self.globals:
            if self.names.get(gid) == name:
                return gid
        raise KeyError(name)

    def globals_by_storage(self, storage):
        return [g for g, (_, sc) in self.globals.items() if sc == storage]

    def pointee(self, gid):
        return self.types[self.globals[gid][0]][2]

    def builtin_of(self, gid):
        d = self.decor.get(gid, {})
        return d[11][0] if 11 in d else None

    def location_of(self, gid):
        d = self.decor.get(gid, {})
        return d[30][0] if 30 in d else None

    def struct_member_index(self, type_id, member):
        for k, v in self.member_names.get(type_id, {}).items():
            if v == member:
                return k
        raise KeyError(member)

    def zero(self, tid):
        t = self.types[tid]
        k = t[0]
        if k == "float":
            return F32(0)
        if k == "int":
            return np.int32(0) if t[2] else np.uint32(0)
        if k == "bool":
            return False
        if k in ("vec", "mat"):
            return [self.zero(t[1]) for _ in range(t[2])]
        if k == "array":
            return [self.zero(t[1]) for _ in range(int(self.consts[t[2]]))]
        if k == "struct":
            return [self.zero(m) for m in t[1]]
        if k == "rtarray":
            return []
        if k in ("image", "sampledimage"):
            return k  # opaque handle: the harness answers samples / reads / writes
        return None


class Machine:
    """one invocation of a module's entry point"""

    def __init__(self, module, mode="strict", sample=None, image_size=None, image_write=None):
        assert mode in ("strict", "fused")
        self.m, self.mode = module, mode
        self.sample, self.image_size, self.image_write = sample, image_size, image_write
        self.boxes = {gid: Box(module.zero(module.pointee(gid))) for gid in module.globals}
        self.op_count = {}

    def set_global(self, name, value):
        self.boxes[self.m.global_by_name(name)].value = _copy(value)

    def get_global(self, name):
        return _copy(self.boxes[self.m.global_by_name(name)].value)

    def set_builtin(self, builtin, value, storage=1):
        """inputs (storage class 1) that are bare variables decorated BuiltIn"""
        for gid, (_, sc) in self.m.globals.items():
            if sc == storage and self.m.builtin_of(gid) == builtin:
                self.boxes[gid].value = _copy(value)
                return
        raise KeyError("no builtin %d" % builtin)

    def set_location(self, location, value, storage=1):
        for gid, (_, sc) in self.m.globals.items():
            if sc == storage and self.m.location_of(gid) == location:
                self.boxes[gid].value = _copy(value)
                return
        raise KeyError("no variable at location %d" % location)

    def get_location(self, location, storage=3):
        for gid, (_, sc) in self.m.globals.items():
            if sc == storage and self.m.location_of(gid) == location:
                return _copy(self.boxes[gid].value)
        raise KeyError("no variable at location %d" % location)

    def get_builtin_member(self, builtin, storage=3):
        """gl_Position & co.: members of the gl_PerVertex block decorated BuiltIn"""
        for gid, (_, sc) in self.m.globals.items():
            if sc != storage:
                continue
            st = self.m.pointee(gid)
            for k, d in self.m.member_decor.get(st, {}).items():
                if d.get(11, [None])[0] == builtin:
                    return _copy(self.boxes[gid].value[k])
        raise KeyError("no builtin member %d" % builtin)

    def run(self):
        with np.errstate(all="ignore"):
            self._call(self.m.entry, [])

    # ---- arithmetic
    def _count(self, op):
        self.op_count[op] = self.op_count.get(op, 0) + 1

    def _dot(self, a, b):
        if self.mode == "strict":
            r = a[0] * b[0]
            for x, y in zip(a[1:], b[1:]):
                r = r + x * y
            return r
        r = a[0] * b[0]
        for x, y in zip(a[1:], b[1:]):
            r = fmaf(x, y, r)
        return r

    def _mat_vec(self, m, v):
        # result[r] = sum_c m[c][r] * v[c], accumulated over the columns in order
        rows = len(m[0])
        return [self._dot([m[c][r] for c in range(len(m))], v) for r in range(rows)]

    def _map(self, f, *xs):
        if isinstance(xs[0], list):
            return [self._map(f, *[x[i] for x in xs]) for i in range(len(xs[0]))]
        return f(*xs)

    def _ext(self, n, args):
        if n == 8:
            return self._map(lambda x: F32(np.floor(x)), args[0])
        if n == 10:
            return self._map(lambda x: x - F32(np.floor(x)), args[0])
        if n == 14:  # Cos: correctly rounded (the specification promises far less)
            return self._map(lambda x: F32(math.cos(float(x))) if np.isfinite(x) else F32(np.nan), args[0])
        if n == 26:
            return self._map(lambda x, y: F32(math.pow(float(x), float(y))) if x >= 0 else F32(np.nan), args[0], args[1])
        if n == 40:  # FMax
            return self._map(lambda x, y: y if x < y else x, args[0], args[1])
        if n == 46:  # FMix: x * (1 - a) + y * a
            if self.mode == "strict":
                return self._map(lambda x, y, a: x * (F32(1) - a) + y * a, *args)
            return self._map(lambda x, y, a: fmaf(y, a, x * (F32(1) - a)), *args)
        raise NotImplementedError("GLSL.std.450 instruction %d" % n)

    # ---- memory
    def _load(self, p):
        v = p.box.value
        for i in p.path:
            v = v[i]
        if v is None:
            raise RuntimeError("load of an undefined value")
        return _copy(v)

    def _store(self, p, val):
        val = _copy(val)
        if not p.path:
            p.box.value = val
            return
        v = p.box.value
        for i in p.path[:-1]:
            v = v[i]
        v[p.path[-1]] = val

    # ---- execution
    def _call(self, fid, args):
        fn = self.m.functions[fid]
        m, consts = self.m, self.m.consts
        vals, prod = {}, {}   # prod: result id -> (a, b) of the multiply that made it (fused mode)
        var_prod = {}         # function variable id -> prod of the value last stored whole
        for pid, arg in zip(fn["params"], args):
            vals[pid] = arg

        def V(i):
            if i in vals:
                return vals[i]
            if i in consts:
                return consts[i]
            if i in self.boxes:
                return Pointer(self.boxes[i])
            raise KeyError("id %d" % i)

        label, prev = fn["first"], None
        while True:
            nxt = None
            for op, a in fn["blocks"][label]:
                self._count(op)
                if op == 59:      # OpVariable (Function)
                    box = Box(m.zero(m.types[a[0]][2]))
                    if len(a) > 3:
                        box.value = _copy(V(a[3]))
                    vals[a[1]] = Pointer(box)
                elif op == 61:    # OpLoad
                    p = V(a[2])
                    vals[a[1]] = self._load(p)
                    if not p.path and id(p.box) in var_prod:
                        prod[a[1]] = var_prod[id(p.box)]
                elif op == 62:    # OpStore
                    p = V(a[0])
                    self._store(p, V(a[1]))
                    if not p.path:
                        if a[1] in prod:
                            var_prod[id(p.box)] = prod[a[1]]
                        else:
                            var_prod.pop(id(p.box), None)
                    else:
                        var_prod.pop(id(p.box), None)
                elif op == 65:    # OpAccessChain
                    p = V(a[2])
                    vals[a[1]] = Pointer(p.box, p.path + tuple(int(V(x)) for x in a[3:]))
                elif op == 79:    # OpVectorShuffle
                    both = list(V(a[2])) + list(V(a[3]))
                    vals[a[1]] = [both[k] for k in a[4:]]
                elif op == 80:    # OpCompositeConstruct
                    t = m.types[a[0]]
                    parts = []
                    for x in a[2:]:
                        v = V(x)
                        if t[0] == "vec" and isinstance(v, list):
                            parts.extend(v)
                        else:
                            parts.append(_copy(v))
                    vals[a[1]] = parts
                elif op == 81:    # OpCompositeExtract
                    v = V(a[2])
                    for k in a[3:]:
                        v = v[k]
                    vals[a[1]] = _copy(v)
                elif op == 12:    # OpExtInst
                    vals[a[1]] = self._ext(a[3], [V(x) for x in a[4:]])
                elif op == 87:    # OpImageSampleImplicitLod
                    vals[a[1]] = [F32(x) for x in self.sample(V(a[3]))]
                elif op == 104:   # OpImageQuerySize
                    vals[a[1]] = [np.int32(x) for x in self.image_size]
                elif op == 99:    # OpImageWrite
                    self.image_write([int(x) for x in V(a[1])], V(a[2]))
                elif op == 111:   # OpConvertSToF
                    vals[a[1]] = self._map(lambda x: F32(int(x)), V(a[2]))
                elif op == 124:   # OpBitcast (uvec -> ivec here)
                    t = m.types[a[0]]
                    et = m.types[t[1]] if t[0] == "vec" else t
                    conv = (lambda x: np.uint32(x).astype(np.int32)) if et[0] == "int" and et[2] else (lambda x: np.int32(x).astype(np.uint32))
                    vals[a[1]] = self._map(conv, V(a[2]))
                elif op == 129:   # OpFAdd
                    x, y = V(a[2]), V(a[3])
                    if self.mode == "fused" and a[2] in prod:
                        pa, pb = prod[a[2]]
                        vals[a[1]] = self._map(fmaf, pa, pb, y)
                    elif self.mode == "fused" and a[3] in prod:
                        pa, pb = prod[a[3]]
                        vals[a[1]] = self._map(fmaf, pa, pb, x)
                    else:
                        vals[a[1]] = self._map(lambda p, q: p + q, x, y)
                elif op == 131:   # OpFSub
                    vals[a[1]] = self._map(lambda p, q: p - q, V(a[2]), V(a[3]))
                elif op == 133:   # OpFMul
                    x, y = V(a[2]), V(a[3])
                    vals[a[1]] = self._map(lambda p, q: p * q, x, y)
                    prod[a[1]] = (x, y)
                elif op == 136:   # OpFDiv
                    vals[a[1]] = self._map(lambda p, q: p / q, V(a[2]), V(a[3]))
                elif op == 142:   # OpVectorTimesScalar
                    x, s = V(a[2]), V(a[3])
                    vals[a[1]] = [c * s for c in x]
                    prod[a[1]] = (x, [s] * len(x))
                elif op == 145:   # OpMatrixTimesVector
                    vals[a[1]] = self._mat_vec(V(a[2]), V(a[3]))
                elif op == 146:   # OpMatrixTimesMatrix: column j of the result = left * right[j]
                    left, right = V(a[2]), V(a[3])
                    vals[a[1]] = [self._mat_vec(left, col) for col in right]
                elif op == 148:   # OpDot
                    vals[a[1]] = self._dot(V(a[2]), V(a[3]))
                elif op == 167:   # OpLogicalAnd
                    vals[a[1]] = bool(V(a[2])) and bool(V(a[3]))
                elif op == 177:   # OpSLessThan
                    vals[a[1]] = self._map(lambda p, q: bool(p < q), V(a[2]), V(a[3]))
                elif op == 171:   # OpINotEqual
                    vals[a[1]] = self._map(lambda p, q: bool(p != q), V(a[2]), V(a[3]))
                elif op == 190:   # OpFOrdGreaterThanEqual
                    vals[a[1]] = self._map(lambda p, q: bool(p >= q), V(a[2]), V(a[3]))
                elif op == 245:   # OpPhi
                    for k in range(2, len(a), 2):
                        if a[k + 1] == prev:
                            vals[a[1]] = V(a[k])
                            break
                    else:
                        raise RuntimeError("OpPhi without an edge from block %r" % prev)
                elif op == 57:    # OpFunctionCall
                    vals[a[1]] = self._call(a[2], [V(x) for x in a[3:]])
                elif op in (246, 247):
                    pass          # merge declarations
                elif op == 249:
                    nxt = a[0]
                elif op == 250:
                    nxt = a[1] if V(a[0]) else a[2]
                elif op == 253:
                    return None
                elif op == 254:
                    return V(a[0])
                else:
                    raise NotImplementedError("opcode %d" % op)
                if nxt is not None:
                    break
            if nxt is None:
                raise RuntimeError("block %d fell off its end" % label)
            prev, label = label, nxt
