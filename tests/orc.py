"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference encoder.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (draco-oxide_amd/) never touches this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

# wire ids (reference core/attribute/mod.rs:568-582, :648-661, :705-710)
POSITION, NORMAL, COLOR, TEXCOORD, CUSTOM = 0, 1, 2, 3, 4
DOM_POSITION, DOM_CORNER = 0, 1
U8, I8, U16, I16, U32, I32, U64, I64, F32, F64 = range(1, 11)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        build()
    L = C.CDLL(LIB_PATH)
    L.orc_last_error.restype = C.c_char_p
    L.orc_session_new.restype = C.c_void_p
    L.orc_session_free.argtypes = [C.c_void_p]
    L.orc_load_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.orc_builder_add_attribute.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int]
    L.orc_builder_set_faces.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_build.argtypes = [C.c_void_p, C.c_int]
    L.orc_num_faces.argtypes = [C.c_void_p]
    L.orc_num_faces.restype = C.c_uint32
    L.orc_faces.argtypes = [C.c_void_p]
    L.orc_faces.restype = C.c_void_p
    L.orc_num_attributes.argtypes = [C.c_void_p]
    L.orc_num_attributes.restype = C.c_uint32
    L.orc_attribute_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_attribute_data.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_attribute_data.restype = C.c_void_p
    L.orc_attribute_map.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_attribute_map.restype = C.c_void_p
    L.orc_attribute_num_parents.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_attribute_num_parents.restype = C.c_uint32
    L.orc_attribute_parents.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_attribute_parents.restype = C.c_void_p
    L.orc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_last_encode_seconds.argtypes = [C.c_void_p]
    L.orc_last_encode_seconds.restype = C.c_double
    L.orc_last_stage_seconds.argtypes = [C.c_void_p]
    L.orc_drc.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_drc.restype = C.c_void_p
    L.orc_blob.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    L.orc_blob.restype = C.c_void_p
    L.orc_leb128.argtypes = [C.c_uint64, C.c_void_p]
    L.orc_leb128.restype = C.c_uint64
    L.orc_bitwriter.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.orc_bitwriter.restype = C.c_uint64
    L.orc_rans_encode_raw.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.orc_rans_encode_raw.restype = C.c_int64
    L.orc_rans_decode_raw.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]
    L.orc_rabs_encode.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.orc_rabs_encode.restype = C.c_int64
    L.orc_rabs_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
    L.orc_encode_symbols.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.orc_encode_symbols.restype = C.c_int64
    L.orc_decode_symbols.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_decode_attributes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.orc_decoded_count.argtypes = [C.c_void_p]
    L.orc_decoded_count.restype = C.c_uint32
    L.orc_decoded_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_decoded_array.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.orc_decoded_array.restype = C.c_void_p
    L.orc_oct_orthogonal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


class OracleError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_NP_OF = {F32: np.float32, U32: np.uint32, I32: np.int32, U8: np.uint8, I8: np.int8, U16: np.uint16, I16: np.int16, F64: np.float64}


class Session:
    """One mesh + the result of encoding it with the restated reference encoder."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_session_new())

    def __del__(self):
        try:
            self.L.orc_session_free(self.h)
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise OracleError(self.L.orc_last_error().decode())

    @classmethod
    def from_obj(cls, path, faithful=False):
        s = cls()
        s._check(s.L.orc_load_obj(s.h, path.encode(), int(faithful)))
        return s

    @classmethod
    def from_arrays(cls, faces, attributes, faithful=False):
        """attributes: list of dicts {data: ndarray [P, N], type, domain, parents: [ids]} in add order
        (ids are 0,1,2..., like MeshBuilder::add_attribute)."""
        s = cls()
        s.L.orc_builder_reset()
        for a in attributes:
            d = np.ascontiguousarray(a["data"])
            if d.ndim == 1:
                d = d.reshape(-1, 1)
            ct = {np.dtype(np.float32): F32, np.dtype(np.uint32): U32, np.dtype(np.int32): I32}[d.dtype]
            par = np.asarray(a.get("parents", []), dtype=np.uint32)
            s.L.orc_builder_add_attribute(_ptr(d), d.shape[0], a["type"], a.get("domain", DOM_POSITION), ct, d.shape[1], _ptr(par), len(par), int(faithful))
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        s.L.orc_builder_set_faces(_ptr(f), f.shape[0])
        s._check(s.L.orc_build(s.h, int(faithful)))
        return s

    # ---- mesh accessors ----
    def faces(self):
        n = self.L.orc_num_faces(self.h)
        if n == 0:
            return np.zeros((0, 3), np.uint32)
        p = self.L.orc_faces(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n, 3)).copy()

    def attributes(self):
        out = []
        for i in range(self.L.orc_num_attributes(self.h)):
            info = np.zeros(8, np.uint32)
            self.L.orc_attribute_info(self.h, i, _ptr(info))
            aid, ty, dom, ct, nc, nu, ln, hm = [int(x) for x in info]
            dt = _NP_OF[ct]
            if nu:
                p = self.L.orc_attribute_data(self.h, i)
                data = np.frombuffer(C.string_at(p, nu * nc * np.dtype(dt).itemsize), dtype=dt).reshape(nu, nc).copy()
            else:
                data = np.zeros((0, nc), dt)
            p2v = None
            if hm:
                p = self.L.orc_attribute_map(self.h, i)
                p2v = np.frombuffer(C.string_at(p, ln * 4), dtype=np.uint32).copy()
            npar = self.L.orc_attribute_num_parents(self.h, i)
            parents = []
            if npar:
                p = self.L.orc_attribute_parents(self.h, i)
                parents = [int(x) for x in np.frombuffer(C.string_at(p, npar * 4), dtype=np.uint32)]
            out.append(dict(id=aid, type=ty, domain=dom, comp_type=ct, ncomp=nc, num_unique=nu, len=ln, data=data, p2v=p2v, parents=parents))
        return out

    # ---- encode ----
    def encode(self, faithful=False, pos_bits=11, uv_bits=10, generic_bits=11, positions_delta=False, dump=True):
        opts = np.array([int(faithful), pos_bits, uv_bits, generic_bits, int(positions_delta)], dtype=np.int32)
        self._check(self.L.orc_encode(self.h, _ptr(opts), int(dump)))
        n = C.c_uint64(0)
        p = self.L.orc_drc(self.h, C.byref(n))
        return C.string_at(p, n.value)

    def encode_seconds(self):
        return float(self.L.orc_last_encode_seconds(self.h))

    def stage_seconds(self):
        """(connectivity, attribute section, sequencer part of the attribute section) of the last encode."""
        out = np.zeros(3, np.float64)
        self.L.orc_last_stage_seconds(_ptr(out))
        return tuple(float(x) for x in out)

    def stage_split(self):
        """BASELINE.md §3 per-stage split of the last encode (seconds on one host core)."""
        out = np.zeros(11, np.float64)
        self.L.orc_last_stage_split(_ptr(out))
        conn, atts, seq, tables, quant, pred, trans, hist_tab, coders, rans_only, n_rans = (float(x) for x in out)
        return {"corner_tables_s": tables, "edgebreaker_s": conn - tables, "sequencer_s": seq, "quantize_s": quant, "predict_s": pred, "transform_s": trans,
                "histogram_and_tables_s": hist_tab, "rans_rabs_coders_s": coders, "attribute_section_s": atts, "connectivity_s": conn,
                "rans_only_msym_per_s": (n_rans / rans_only / 1e6) if rans_only > 0 else 0.0, "rans_symbols": n_rans}

    def decode_attributes(self, section):
        """The attribute section `section` (bytes) read backwards against this mesh's connectivity stage (oracle/orc_decode.cpp).
        Returns [dict(id, type, ncomp, ncomp_port, port, scheme, transform, portable [n, ncomp_port] i32, values [n, ncomp] f32 (u32 for
        ToBits), points [n] u32 = the point behind every sequence entry)], and the number of bytes consumed."""
        b = np.frombuffer(section, dtype=np.uint8)
        used = C.c_uint64(0)
        self._check(self.L.orc_decode_attributes(self.h, _ptr(b), len(b), C.byref(used)))
        out = []
        for i in range(self.L.orc_decoded_count(self.h)):
            info = np.zeros(8, np.uint32)
            self.L.orc_decoded_info(self.h, i, _ptr(info))
            aid, ty, nc, ncp, port, scheme, transform, n = [int(x) for x in info]

            def arr(which, dtype, cols):
                cnt = C.c_uint64(0)
                p = self.L.orc_decoded_array(self.h, i, which, C.byref(cnt))
                if not p or cnt.value == 0:
                    return np.zeros((0, cols) if cols else 0, dtype)
                a = np.frombuffer(C.string_at(p, cnt.value * 4), dtype=dtype).copy()
                return a.reshape(-1, cols) if cols else a
            vals = arr(1, np.uint32 if port == 1 else np.float32, nc)
            out.append(dict(id=aid, type=ty, ncomp=nc, ncomp_port=ncp, port=port, scheme=scheme, transform=transform,
                            portable=arr(0, np.int32, ncp), values=vals, points=arr(2, np.uint32, 0)))
        return out, used.value

    def blob(self, key, dtype=np.uint8):
        n = C.c_uint64(0)
        p = self.L.orc_blob(self.h, key.encode(), C.byref(n))
        if not p or n.value == 0:
            return np.zeros(0, dtype)
        return np.frombuffer(C.string_at(p, n.value), dtype=dtype).copy()


# ---- small KAT hooks ----
def leb128(v):
    out = np.zeros(16, np.uint8)
    n = lib().orc_leb128(v, _ptr(out))
    return bytes(out[:n])


def bitwriter(ops, msb):
    a = np.array([x for op in ops for x in op], dtype=np.uint64)
    out = np.zeros(8 * len(ops) + 8, np.uint8)
    n = lib().orc_bitwriter(_ptr(a), len(ops), int(msb), _ptr(out))
    return bytes(out[:n])


def rans_encode_raw(dist, precision, syms):
    d = np.asarray(dist, dtype=np.uint64)
    s = np.asarray(syms, dtype=np.uint32)
    out = np.zeros(4 * len(s) + 16, np.uint8)
    n = lib().orc_rans_encode_raw(_ptr(d), len(d), precision, _ptr(s), len(s), _ptr(out), len(out))
    if n < 0:
        raise OracleError(lib().orc_last_error().decode())
    return bytes(out[:n])


def rans_decode_raw(data, dist, precision, n):
    d = np.asarray(dist, dtype=np.uint64)
    b = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(n, np.uint32)
    if lib().orc_rans_decode_raw(_ptr(b), len(b), _ptr(d), len(d), precision, n, _ptr(out)) != 0:
        raise OracleError(lib().orc_last_error().decode())
    return out


def rabs_encode(zero_prob, bits):
    b = np.asarray(bits, dtype=np.uint8)
    out = np.zeros(len(b) + 16, np.uint8)
    n = lib().orc_rabs_encode(zero_prob, _ptr(b), len(b), _ptr(out), len(out))
    if n < 0:
        raise OracleError(lib().orc_last_error().decode())
    return bytes(out[:n])


def rabs_decode(data, zero_prob, n):
    b = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(n, np.uint8)
    if lib().orc_rabs_decode(_ptr(b), len(b), zero_prob, n, _ptr(out)) != 0:
        raise OracleError(lib().orc_last_error().decode())
    return out


def encode_symbols(syms):
    s = np.asarray(syms, dtype=np.uint32)
    out = np.zeros(4 * len(s) + (1 << 20), np.uint8)
    n = lib().orc_encode_symbols(_ptr(s), len(s), _ptr(out), len(out))
    if n < 0:
        raise OracleError(lib().orc_last_error().decode())
    return bytes(out[:n])


def decode_symbols(data, n):
    b = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(n, np.uint32)
    used = C.c_uint64(0)
    if lib().orc_decode_symbols(_ptr(b), len(b), n, _ptr(out), C.byref(used)) != 0:
        raise OracleError(lib().orc_last_error().decode())
    return out, used.value


def oct_orthogonal_round_trip(orig, pred):
    """(corr, back): the oct-orthogonal prediction transform of `orig` against `pred` and its inverse applied to the result."""
    o = np.asarray(orig, np.int32); p = np.asarray(pred, np.int32)
    corr = np.zeros(2, np.int32); back = np.zeros(2, np.int32)
    lib().orc_oct_orthogonal(_ptr(o), _ptr(p), _ptr(corr), _ptr(back))
    return corr, back
