"""CPU: the host-side file writer / reader of the C ABI (pccx_write_streams_host / pccx_read_streams_host / pccx_stream_sizes_host;
csrc/hostio.hip -- host threads, no GPU call) against the reference's own file code, statement for statement:
compress.py:139-152 (open(... + '.p.bin', 'wb').write(bytes), '.s.bin' likewise, np.float32[4].tofile('.c.bin')) and
decompress.py:80-91,113 (open(...).read(), np.fromfile(dtype=float32))."""
import os

import numpy as np
import pytest
import torch

from pccx import _lib, codec


def _random_batch(B, s_stride, p_cap, seed):
    rng = np.random.default_rng(seed)
    host = torch.from_numpy(rng.integers(0, 256, codec.packed_layout(B, s_stride, p_cap)[-1], dtype=np.uint8))
    h = codec.Compressed.from_packed(host, B, s_stride, p_cap, 8192)
    h.s_nbytes.copy_(torch.from_numpy(rng.integers(0, s_stride + 1, B).astype(np.int32)))
    h.p_nbytes.copy_(torch.from_numpy(rng.integers(0, p_cap + 1, B).astype(np.int32)))
    h.c.copy_(torch.from_numpy(rng.standard_normal((B, 4)).astype(np.float32)))
    if B > 2:
        h.s_nbytes[0], h.p_nbytes[0] = 0, 0                     # empty streams
        h.s_nbytes[1], h.p_nbytes[1] = s_stride, p_cap          # full rows
    return host, h


@pytest.mark.parametrize("B,s_stride,p_cap,threads", [(1, 7, 5, 1), (37, 309, 1154, 3), (300, 299, 700, 0)])
def test_files_equal_the_reference_writer_and_read_back(tmp_path, B, s_stride, p_cap, threads):
    host, h = _random_batch(B, s_stride, p_cap, B)
    names = [f"cloud {i:04d}.ply" for i in range(B)]            # the reference names files <input file name> + ext, dots and spaces included
    ours, ref = tmp_path / "ours", tmp_path / "ref"
    ours.mkdir(), ref.mkdir()
    codec.write_streams(host, B, s_stride, p_cap, str(ours), names, threads)
    for b, n in enumerate(names):                               # the reference's writer (compress.py:139-152)
        with open(ref / (n + ".p.bin"), "wb") as f:
            f.write(bytes(h.p_bytes[b, :int(h.p_nbytes[b])].numpy()))
        with open(ref / (n + ".s.bin"), "wb") as f:
            f.write(bytes(h.s_bytes[b, :int(h.s_nbytes[b])].numpy()))
        arr = np.zeros(4)
        arr[:3], arr[3] = h.c[b, :3].numpy(), float(h.c[b, 3])
        arr.astype(np.float32).tofile(ref / (n + ".c.bin"))
    assert sorted(os.listdir(ours)) == sorted(os.listdir(ref)) and len(os.listdir(ours)) == 3 * B
    for f in os.listdir(ref):
        assert open(ours / f, "rb").read() == open(ref / f, "rb").read(), f
    ss, ps = codec.stream_sizes(str(ours), names + ["missing"])
    assert ss[:B].tolist() == h.s_nbytes.tolist() and ps[:B].tolist() == h.p_nbytes.tolist() and ss[B] == -1 and ps[B] == -1
    # read back into a buffer full of garbage: counts, centres and the filled parts of the rows return, the tails are cleared
    back = torch.full_like(host, 0xAB)
    codec.read_streams(back, B, s_stride, p_cap, str(ours), names, threads)
    g = codec.Compressed.from_packed(back, B, s_stride, p_cap, 8192)
    assert torch.equal(g.s_nbytes, h.s_nbytes) and torch.equal(g.p_nbytes, h.p_nbytes)
    assert g.c.numpy().tobytes() == h.c.numpy().tobytes()
    for b in range(B):
        sn, pn = int(h.s_nbytes[b]), int(h.p_nbytes[b])
        assert torch.equal(g.s_bytes[b, :sn], h.s_bytes[b, :sn]) and not g.s_bytes[b, sn:].any()
        assert torch.equal(g.p_bytes[b, :pn], h.p_bytes[b, :pn]) and not g.p_bytes[b, pn:].any()
    # ... and what the reference's reader sees (decompress.py:80-91,113)
    for b in (0, B - 1):
        assert np.array_equal(np.fromfile(ours / (names[b] + ".c.bin"), dtype=np.float32), h.c[b].numpy())
    # Compressed.read_files sizes its rows from the files
    r = codec.Compressed.read_files(str(ours), names, n_points=8192, threads=threads)
    assert r.s_bytes.shape[1] == max(1, int(h.s_nbytes.max())) and torch.equal(r.p_nbytes, h.p_nbytes)
    assert int(r.bits().sum()) == int(h.bits().sum())


def test_writer_and_reader_refuse_bad_input(tmp_path):
    B, s_stride, p_cap = 4, 16, 32
    host, h = _random_batch(B, s_stride, p_cap, 1)
    names = [f"n{i}" for i in range(B)]
    h.p_nbytes[2] = -40                                          # the range coder's "buffer too small" mark (rangecoder.hip)
    with pytest.raises(_lib.PccxError, match="cloud 2"):
        codec.write_streams(host, B, s_stride, p_cap, str(tmp_path), names)
    assert os.listdir(tmp_path) == []                            # refused before touching the file system
    h.p_nbytes[2] = 3
    with pytest.raises(_lib.PccxError, match="open"):
        codec.write_streams(host, B, s_stride, p_cap, str(tmp_path / "no such dir"), names)
    with pytest.raises(ValueError):
        codec.write_streams(host, B, s_stride, p_cap, str(tmp_path), ["a/b"] * B)
    with pytest.raises(ValueError):
        codec.write_streams(host[:-1], B, s_stride, p_cap, str(tmp_path), names)
    codec.write_streams(host, B, s_stride, p_cap, str(tmp_path), names)
    back = torch.zeros_like(host)
    os.remove(tmp_path / "n1.p.bin")
    with pytest.raises(_lib.PccxError, match="n1.p.bin"):
        codec.read_streams(back, B, s_stride, p_cap, str(tmp_path), names)
    with pytest.raises(_lib.PccxError, match="missing"):
        codec.Compressed.read_files(str(tmp_path), names)
    open(tmp_path / "n1.p.bin", "wb").write(b"x" * (p_cap + 1))  # longer than its row
    with pytest.raises(_lib.PccxError, match="longer"):
        codec.read_streams(back, B, s_stride, p_cap, str(tmp_path), names)
    open(tmp_path / "n1.p.bin", "wb").write(b"x")
    open(tmp_path / "n3.c.bin", "wb").write(b"\0" * 12)          # a centre file that is not 4 floats
    with pytest.raises(_lib.PccxError, match="n3.c.bin"):
        codec.read_streams(back, B, s_stride, p_cap, str(tmp_path), names)
    assert _lib.load().pccx_streams_packed_bytes(B, s_stride, p_cap) == codec.packed_layout(B, s_stride, p_cap)[-1]


def test_pool_survives_repeated_calls_with_changing_thread_counts(tmp_path):
    """the pool grows on demand and is reused: many short calls with 1..8 threads, files always complete"""
    B, s_stride, p_cap = 64, 40, 90
    host, h = _random_batch(B, s_stride, p_cap, 9)
    names = [f"{i:03d}" for i in range(B)]
    for it in range(40):
        codec.write_streams(host, B, s_stride, p_cap, str(tmp_path), names, 1 + it % 8)
        back = torch.zeros_like(host)
        codec.read_streams(back, B, s_stride, p_cap, str(tmp_path), names, 1 + (it * 3) % 8)
        assert torch.equal(codec.Compressed.from_packed(back, B, s_stride, p_cap, 0).p_nbytes, h.p_nbytes)
