import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def stacker():
    """One stk_ctx on cuda:0 for the whole GPU session (fails loudly without the HIP library)."""
    from libstacker_rs_amd import Stacker
    s = Stacker(0)
    yield s
    s.close()


@pytest.fixture(scope="session")
def small_stack():
    """4 x 320x240 synthetic BGR u8 frames + ground-truth homographies (seeded)."""
    from libstacker_rs_amd import synth
    frames, G = synth.make_stack(4, 320, 240)
    return frames.numpy(), G


def _write_png(path, img):
    """Minimal PNG writer (zlib + CRC): 8-bit grey (HxW), RGB (HxWx3, given as BGR) or RGBA (HxWx4), no gamma chunk."""
    import struct
    import zlib
    a = np.asarray(img, np.uint8)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[..., ::-1]                                     # BGR in memory -> RGB on disk
    h, w = a.shape[:2]
    ctype = {2: 0, 3: 2, 4: 6}[2 if a.ndim == 2 else a.shape[2]]
    raw = b"".join(b"\x00" + np.ascontiguousarray(a[y]).tobytes() for y in range(h))
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.fixture(scope="session")
def write_png():
    """Writer of small test PNGs (no Pillow in the main interpreter)."""
    return _write_png


def _write_tiff(path, img):
    """Minimal baseline TIFF writer: little-endian, uncompressed, one strip; 8/16-bit grey (HxW) or RGB (HxWx3 given as BGR)."""
    import struct
    a = np.asarray(img)
    assert a.dtype in (np.uint8, np.uint16)
    if a.ndim == 3:
        a = a[..., ::-1]
    h, w = a.shape[:2]
    spp = 1 if a.ndim == 2 else 3
    bps = a.dtype.itemsize * 8
    data = np.ascontiguousarray(a).astype("<u%d" % a.dtype.itemsize).tobytes()
    n_tags = 10
    ifd_ofs = 8
    bps_ofs = ifd_ofs + 2 + 12 * n_tags + 4
    data_ofs = bps_ofs + 8
    def tag(t, typ, count, value):
        return struct.pack("<HHII", t, typ, count, value)
    ifd = struct.pack("<H", n_tags)
    ifd += tag(256, 4, 1, w) + tag(257, 4, 1, h)
    ifd += tag(258, 3, spp, bps_ofs if spp == 3 else bps)
    ifd += tag(259, 3, 1, 1) + tag(262, 3, 1, 2 if spp == 3 else 1)
    ifd += tag(273, 4, 1, data_ofs) + tag(277, 3, 1, spp) + tag(278, 4, 1, h) + tag(279, 4, 1, len(data)) + tag(284, 3, 1, 1)
    ifd += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_ofs) + ifd + struct.pack("<HHHH", bps, bps, bps, 0) + data)


@pytest.fixture(scope="session")
def write_tiff():
    """Writer of small test TIFFs (no Pillow in the main interpreter)."""
    return _write_tiff
