"""Minimal PLY point-cloud I/O (x,y,z vertices), replacing the reference's plyfile / pyntcloud use
(pn_kit.py:25-42: read_point_cloud, save_point_cloud).  Reads ASCII and binary_little_endian files
with arbitrary extra vertex properties; writes binary_little_endian float32 x,y,z."""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def read_point_cloud(path):
    """-> (N,3) float32, columns x,y,z (or X,Y,Z, as pn_kit.py:27-30 accepts)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vertex, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vertex = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property on vertex element unsupported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        cols = [names.index(c) if c in names else names.index(c.upper()) for c in ("x", "y", "z")]
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n_vertex, ndmin=2)
            return np.ascontiguousarray(data[:, cols], dtype=np.float32)
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        raw = np.frombuffer(f.read(n_vertex * dt.itemsize), dtype=dt, count=n_vertex)
        return np.stack([raw[names[c]].astype(np.float32) for c in cols], axis=1)


def save_point_cloud(pc, path):
    pc = np.ascontiguousarray(pc, dtype="<f4").reshape(-1, 3)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % pc.shape[0])
        f.write(b"property float x\nproperty float y\nproperty float z\nend_header\n")
        f.write(pc.tobytes())
