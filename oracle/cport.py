"""ctypes binding of oracle/_build/liborc.so (oracle/pcc_oracle.c).

TEST INFRASTRUCTURE ONLY -- the checker, never the product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PCCX_ORACLE_SO") or os.path.join(_HERE, "_build", "liborc.so")      # PCCX_ORACLE_SO: the sanitizer build (`make asan`)


def build(force=False):
    src = os.path.join(_HERE, "pcc_oracle.c")
    if os.environ.get("PCCX_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        f32p, u8p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
        i32p, i16p = C.POINTER(C.c_int32), C.POINTER(C.c_int16)
        L = _lib
        L.orc_get_decode_from_pc.argtypes = [f32p, C.c_int, C.c_double, C.c_int, f32p]
        L.orc_octree_encode.argtypes = [f32p, C.c_int, C.c_double, C.c_int, u8p, C.c_int]
        L.orc_encode_sampled.argtypes = [f32p, C.c_int, C.c_double, C.c_int, C.c_double, u8p, C.c_int,
                                         C.POINTER(C.c_int)]
        L.orc_octree_decode_reference.argtypes = [u8p, C.c_int, C.c_double, f32p]
        L.orc_octree_decode_full.argtypes = [u8p, C.c_int, C.c_double, f32p, C.c_int, C.POINTER(C.c_int)]
        L.orc_pack_bits.argtypes = [u8p, C.c_int, u8p]
        L.orc_unpack_bits.argtypes = [u8p, C.c_int, u8p]
        L.orc_fps.argtypes = [f32p, C.c_int, C.c_int, C.c_int, i64p]
        L.orc_fps.restype = None
        L.orc_knn.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, f32p, i64p]
        L.orc_knn.restype = None
        L.orc_ball_query.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_float, f32p, i64p]
        L.orc_ball_query.restype = None
        L.orc_nn_dist.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, i32p]
        L.orc_nn_dist.restype = None
        L.orc_range_encode.argtypes = [i32p, C.c_int, C.c_int, i16p, u8p, C.c_int]
        L.orc_range_decode.argtypes = [i32p, C.c_int, C.c_int, u8p, C.c_int, i16p]
        L.orc_range_decode.restype = None
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def get_decode_from_pc(pc, resolution, depth):
    pc = _f32(pc).reshape(-1, 3)
    out = np.empty_like(pc)
    n = lib().orc_get_decode_from_pc(_p(pc, C.c_float), pc.shape[0], float(resolution), int(depth), _p(out, C.c_float))
    return out[:n].copy()


def octree_encode(pc, resolution, depth):
    pc = _f32(pc).reshape(-1, 3)
    cap = 1 + 8 * max(pc.shape[0], 1) * (depth + 1) + 64
    bits = np.empty(cap, dtype=np.uint8)
    n = lib().orc_octree_encode(_p(pc, C.c_float), pc.shape[0], float(resolution), int(depth), _p(bits, C.c_uint8), cap)
    assert n >= 0
    return bits[:n].copy()


def encode_sampled(pc, scale, N, min_bpp):
    """One cloud of pn_kit.encode_sampled_np -> (bits, depth)."""
    pc = _f32(pc).reshape(-1, 3)
    cap = 1 + 8 * max(pc.shape[0], 1) * 18 + 64
    bits = np.empty(cap, dtype=np.uint8)
    d = C.c_int(0)
    n = lib().orc_encode_sampled(_p(pc, C.c_float), pc.shape[0], float(scale), int(N), float(min_bpp),
                                 _p(bits, C.c_uint8), cap, C.byref(d))
    assert n >= 0
    return bits[:n].copy(), d.value


def encode_sampled_np(sampled_xyz, scale, N, min_bpp):
    """pn_kit.encode_sampled_np (pn_kit.py:380-401): (B,S,3) -> (list of bit arrays, total bits)."""
    codes, total = [], 0
    for pc in sampled_xyz:
        b, _ = encode_sampled(pc, scale, N, min_bpp)
        codes.append(b)
        total += b.shape[0]
    return codes, total


def octree_decode_reference(bits, resolution=1.0):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.empty((64, 3), dtype=np.float32)
    n = lib().orc_octree_decode_reference(_p(bits, C.c_uint8), bits.shape[0], float(resolution), _p(out, C.c_float))
    return out, n


def octree_decode_full(bits, resolution=1.0, cap=4096):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.empty((cap, 3), dtype=np.float32)
    d = C.c_int(0)
    n = lib().orc_octree_decode_full(_p(bits, C.c_uint8), bits.shape[0], float(resolution), _p(out, C.c_float), cap,
                                     C.byref(d))
    assert n >= 0
    return out[:n].copy(), d.value


def decode_sampled_np(codes, scale, mode="reference"):
    """pn_kit.decode_sampled_np (pn_kit.py:424-431)."""
    if mode == "reference":
        return np.stack([octree_decode_reference(c, scale)[0] for c in codes], axis=0)
    return np.stack([octree_decode_full(c, scale)[0] for c in codes], axis=0)


def pack_bits(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.empty((bits.shape[0] + 7) // 8, dtype=np.uint8)
    n = lib().orc_pack_bits(_p(bits, C.c_uint8), bits.shape[0], _p(out, C.c_uint8))
    return bytearray(out[:n].tobytes())


def unpack_bits(byte_stream):
    b = np.frombuffer(bytes(byte_stream), dtype=np.uint8).copy()
    out = np.empty(b.shape[0] * 8, dtype=np.uint8)
    lib().orc_unpack_bits(_p(b, C.c_uint8), b.shape[0], _p(out, C.c_uint8))
    return out.astype(np.int32)


def fps(xyz, npoint, start):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty(npoint, dtype=np.int64)
    lib().orc_fps(_p(xyz, C.c_float), xyz.shape[0], int(npoint), int(start), _p(out, C.c_int64))
    return out


def knn(q, ref, K):
    q, ref = _f32(q).reshape(-1, 3), _f32(ref).reshape(-1, 3)
    d = np.empty((q.shape[0], K), dtype=np.float32)
    i = np.empty((q.shape[0], K), dtype=np.int64)
    lib().orc_knn(_p(q, C.c_float), q.shape[0], _p(ref, C.c_float), ref.shape[0], int(K), _p(d, C.c_float), _p(i, C.c_int64))
    return d, i


def ball_query(q, ref, K, radius):
    q, ref = _f32(q).reshape(-1, 3), _f32(ref).reshape(-1, 3)
    d = np.empty((q.shape[0], K), dtype=np.float32)
    i = np.empty((q.shape[0], K), dtype=np.int64)
    lib().orc_ball_query(_p(q, C.c_float), q.shape[0], _p(ref, C.c_float), ref.shape[0], int(K), float(radius),
                         _p(d, C.c_float), _p(i, C.c_int64))
    return d, i


def nn_dist(X, Y):
    X, Y = _f32(X).reshape(-1, 3), _f32(Y).reshape(-1, 3)
    d = np.empty(X.shape[0], dtype=np.float32)
    i = np.empty(X.shape[0], dtype=np.int32)
    lib().orc_nn_dist(_p(X, C.c_float), X.shape[0], _p(Y, C.c_float), Y.shape[0], _p(d, C.c_float), _p(i, C.c_int32))
    return d, i


def range_encode(cdf_int, sym):
    """cdf_int: (nsym, Lp) int32 holding uint16 values; sym: (nsym,) int16 -> bytes."""
    cdf_int = np.ascontiguousarray(cdf_int, dtype=np.int32)
    sym = np.ascontiguousarray(sym, dtype=np.int16).reshape(-1)
    nsym, Lp = cdf_int.reshape(-1, cdf_int.shape[-1]).shape
    cap = nsym * 4 + 16
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_range_encode(_p(cdf_int, C.c_int32), nsym, Lp, _p(sym, C.c_int16), _p(out, C.c_uint8), cap)
    assert n >= 0
    return out[:n].tobytes()


def range_decode(cdf_int, byte_stream):
    cdf_int = np.ascontiguousarray(cdf_int, dtype=np.int32)
    nsym, Lp = cdf_int.reshape(-1, cdf_int.shape[-1]).shape
    b = np.frombuffer(bytes(byte_stream), dtype=np.uint8).copy()
    out = np.empty(nsym, dtype=np.int16)
    lib().orc_range_decode(_p(cdf_int, C.c_int32), nsym, Lp, _p(b, C.c_uint8), b.shape[0], _p(out, C.c_int16))
    return out
