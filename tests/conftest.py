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
