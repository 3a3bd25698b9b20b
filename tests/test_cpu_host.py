"""CPU tests of the host side: C-ABI exports, parameter/error mirrors, sharding, loud failure without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import libstacker_rs_amd as ls
from libstacker_rs_amd import _ffi, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "stacker.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(stk_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()                                   # raises if the .so is missing or a symbol is not exported
    syms = _header_symbols()
    assert len(syms) >= 19
    assert sorted(_ffi.SIGNATURES) == syms               # ctypes table == header
    nm = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (stk_[a-z0-9_]+)", nm))
    assert set(syms) <= exported
    assert lib.stk_version().startswith(b"libstacker_rs_amd")


def test_library_is_gfx950_only_and_has_no_cpu_fallback():
    out = subprocess.run(["strings", "-n", "6", _ffi.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    src = "".join(open(os.path.join(ROOT, "libstacker_rs_amd", f)).read() for f in ("api.py", "_ffi.py", "__init__.py", "shard.py"))
    assert "import oracle" not in src and "from oracle" not in src      # the product never touches the oracle


def test_struct_layouts_match_the_header():
    assert C.sizeof(_ffi.KeypointParams) == 4 + 4 + 8 + 4 + 4 + 4 + 4 + 32      # natural C layout of stk_keypoint_params
    assert C.sizeof(_ffi.EccParams) == 32
    assert C.sizeof(_ffi.FrameStats) == 8 + 8 + 16 + 72
    assert C.sizeof(_ffi.Frames) == 8 + 6 * 4 + 8
    assert C.sizeof(_ffi.Timing) == 16 * 8


def test_no_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ls.HipError):
        ls.Stacker(0)
    lib = _ffi.load()
    h = C.c_void_p()
    assert lib.stk_create(0, C.byref(h)) == 6 and not h.value        # STK_HIP_ERROR, no context
    assert lib.stk_last_error(None) == b"null context"


def test_term_criteria_mapping_doctest():
    # utils.rs:148-158: max_count None, epsilon 0.1 -> typ == EPS, epsilon 0.1
    typ, mc, eps = ls.EccMatchParameters(ls.MotionType.Euclidean, None, 0.1, 3).term_criteria()
    assert (typ, eps) == (2, 0.1)
    assert ls.EccMatchParameters(ls.MotionType.Homography, 5000, 1e-5, 5).term_criteria() == (3, 5000, 1e-5)
    assert ls.EccMatchParameters(ls.MotionType.Affine, 10, None, 5).term_criteria()[0] == 1
    p = ls.EccMatchParameters(ls.MotionType.Homography, None, None, 5)._c()
    assert (p.has_max_count, p.has_epsilon, p.motion_type) == (0, 0, 3)


def test_parameter_defaults_and_constants():
    d = ls.KeyPointMatchParameters()                     # utils.rs:250-261
    assert (d.method, d.ransac_reproj_threshold, d.match_keep_ratio, d.match_ratio, d.border_mode) == (8, 3.0, 0.75, 0.8, 0)
    assert tuple(d.border_value) == (0, 0, 0, 0)
    c = d._c()
    assert c.method == 8 and abs(c.match_ratio - 0.8) < 1e-7 and list(c.border_value) == [0, 0, 0, 0]
    assert [int(m) for m in (ls.MotionType.Translation, ls.MotionType.Euclidean, ls.MotionType.Affine, ls.MotionType.Homography)] == [0, 1, 2, 3]
    assert (ls.RANSAC, ls.LMEDS, ls.RHO) == (8, 4, 16)
    for exc in (ls.NotEnoughFiles, ls.InvalidParams, ls.ProcessingError, ls.OpenCvError, ls.IoError):
        assert issubclass(exc, ls.StackerError)


def test_frame_marshalling_rejects_mixed_stacks():
    from libstacker_rs_amd.api import _Marshalled
    a = np.zeros((4, 5, 3), np.uint8)
    m = _Marshalled([a, a.copy()])
    assert (m.n, m.w, m.h, m.c, m.depth, m.location) == (2, 5, 4, 3, 8, 0)
    assert _Marshalled(np.zeros((3, 4, 5, 3), np.uint16)).depth == 16
    with pytest.raises(ls.InvalidParams):
        _Marshalled([a, np.zeros((4, 6, 3), np.uint8)])
    with pytest.raises(ls.InvalidParams):
        _Marshalled([a.astype(np.float64)])


def test_shard_partition_properties():
    for n in (1, 2, 9, 64, 256, 257):
        for world in (1, 2, 3, 8):
            parts = [shard.shard_moving_frames(n, world, r) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(1, n))                                 # contiguous, ordered, complete, disjoint
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard.shard_moving_frames(256, 8, 0) == list(range(1, 33)) and len(shard.shard_moving_frames(256, 8, 7)) == 31
    assert shard.shard_frame_list(list("abcdefg"), 2, 1) == ["a", "e", "f", "g"]
    with pytest.raises(ValueError):
        shard.shard_moving_frames(4, 2, 2)


def test_synthetic_generator_is_seeded_and_sharded_consistently():
    from libstacker_rs_amd import synth
    a, Ga = synth.make_stack(3, 96, 64)
    b, Gb = synth.make_stack(0, 96, 64, indices=[0, 2])
    assert np.array_equal(a[0].numpy(), b[0].numpy()) and np.array_equal(a[2].numpy(), b[1].numpy())
    assert np.array_equal(Ga[2], Gb[1]) and np.array_equal(Ga[0], np.eye(3))
    assert a.dtype.is_floating_point is False and tuple(a.shape) == (3, 64, 96, 3)
    assert synth.corner_error(np.eye(3), np.eye(3), 96, 64) == 0
    u16, _ = synth.make_stack(1, 32, 24, depth=16)
    assert u16.numpy().dtype == np.uint16


def test_orb_bit_pattern_is_pinned():
    """ORB's 256 BRIEF test pairs (OpenCV `bit_pattern_31_`): pinned by hash, and row by row against scikit-image's
    verbatim copy of the OpenCV table where that package happens to be installed (it is in the build container)."""
    import hashlib
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "libstacker_rs_amd", "csrc", "orb_pattern.h")).read()
    body = txt[txt.index("{", txt.index("ORB_BIT_PATTERN_31")) + 1: txt.rindex("}")]
    table = np.array([int(v) for v in re.findall(r"-?\d+", body)], dtype=np.int8)
    assert table.size == 1024 and np.abs(table).max() <= 13          # 31x31 patch, rotated radius <= 15*sqrt(2)/... fits
    assert hashlib.sha256(table.tobytes()).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    sk = "/opt/conda/lib/python3.9/site-packages/skimage/feature/orb_descriptor_positions.txt"
    if os.path.exists(sk):
        assert np.array_equal(np.loadtxt(sk).astype(np.int8).reshape(-1), table)


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """include/stacker.h must be consumable from C (the Rust shim's bindgen input): compile a C
    translation unit against it, link it with the shared library and run the one call that needs no GPU."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "caller.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "stacker.h"\n'
        "int main(void) {\n"
        "    stk_keypoint_params kp; stk_ecc_params ep; stk_frames fr; stk_image_f32 im; stk_frame_stats fs; stk_timing tm;\n"
        "    memset(&kp, 0, sizeof kp); memset(&ep, 0, sizeof ep); memset(&fr, 0, sizeof fr); memset(&im, 0, sizeof im);\n"
        "    (void)fs; (void)tm; ep.motion_type = STK_MOTION_HOMOGRAPHY; kp.method = STK_METHOD_RANSAC;\n"
        '    printf("%s %d %d\\n", stk_version(), (int)sizeof(stk_frame_stats), (int)STK_SHARPNESS_GLVN);\n'
        "    return stk_ecc_match(NULL, &fr, &ep, 0.0f, &im, NULL) == STK_INVALID_PARAMS ? 0 : 1;\n"
        "}\n")
    exe = tmp_path / "caller"
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                         _ffi.LIB_PATH, "-Wl,-rpath," + os.path.dirname(_ffi.LIB_PATH)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    # torch's bundled HIP runtime is not on the loader path of a plain C program: point it at /opt/rocm like a Rust caller would
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.startswith("libstacker_rs_amd")


def test_imread_pnm_without_a_gpu(tmp_path):
    """stk_imread is pure host code and accepts a NULL context: binary PNM round trip, RGB on disk -> BGR in memory."""
    lib = _ffi.load()
    img = np.random.default_rng(0).integers(0, 256, (7, 9, 3), dtype=np.uint8)
    p = tmp_path / "x.ppm"
    p.write_bytes(b"P6\n# c\n9 7\n255\n" + np.ascontiguousarray(img[..., ::-1]).tobytes())
    w, h, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.stk_imread(None, os.fsencode(p), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
    assert (w.value, h.value, c.value, d.value) == (9, 7, 3, 8)
    out = np.empty((7, 9, 3), np.uint8)
    assert lib.stk_imread(None, os.fsencode(p), C.c_void_p(out.ctypes.data), out.nbytes, None, None, None, None) == 0
    assert np.array_equal(out, img)
    assert lib.stk_imread(None, os.fsencode(p), C.c_void_p(out.ctypes.data), 10, None, None, None, None) == 2      # INVALID_PARAMS
    assert lib.stk_imread(None, os.fsencode(tmp_path / "nope.ppm"), None, 0, None, None, None, None) == 4         # BACKEND_ERROR
    assert lib.stk_imread(None, os.fsencode(tmp_path / "a.jpg"), None, 0, None, None, None, None) == 4            # missing JPEG: BACKEND_ERROR
    assert lib.stk_imread(None, os.fsencode(tmp_path / "a.exr"), None, 0, None, None, None, None) == 7            # no codec: NOT_IMPLEMENTED


def test_imread_png_through_runtime_libpng(tmp_path, write_png):
    """PNG via libpng's row API loaded at run time (no headers in the image): 8- and 16-bit grey / RGB, palette, low-bit
    grey, RGBA -> BGRA; a missing libpng -> NOT_IMPLEMENTED."""
    import ctypes.util
    lib = _ffi.load()
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (13, 21, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (8, 5), dtype=np.uint8)
    write_png(tmp_path / "c.png", bgr)
    write_png(tmp_path / "g.png", grey)
    write_png(tmp_path / "a.png", rng.integers(0, 256, (4, 4, 4), dtype=np.uint8))
    w, h, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    st = lib.stk_imread(None, os.fsencode(tmp_path / "c.png"), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d))
    try:
        have_png = C.CDLL("libpng16.so.16") is not None
    except OSError:
        have_png = False
    if not have_png:
        assert st == 7
        return
    assert st == 0 and (w.value, h.value, c.value, d.value) == (21, 13, 3, 8)
    out = np.empty_like(bgr)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "c.png"), C.c_void_p(out.ctypes.data), out.nbytes, None, None, None, None) == 0
    assert np.array_equal(out, bgr)                              # RGB on disk -> BGR in memory, like imread
    og = np.empty_like(grey)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "g.png"), C.c_void_p(og.ctypes.data), og.nbytes, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
    assert c.value == 1 and np.array_equal(og, grey)
    # alpha: IMREAD_UNCHANGED keeps it — four channels, B G R A (round 4; the reference then stacks all four, utils.rs:132-142)
    rgba = rng.integers(0, 256, (4, 4, 4), dtype=np.uint8)
    write_png(tmp_path / "a.png", rgba)                                        # (four-channel arrays go to disk as given: R G B A)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "a.png"), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
    assert (w.value, h.value, c.value, d.value) == (4, 4, 4, 8)
    oa = np.empty_like(rgba)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "a.png"), C.c_void_p(oa.ctypes.data), oa.nbytes, None, None, None, None) == 0
    assert np.array_equal(oa, rgba[..., [2, 1, 0, 3]])
    rgba16 = rng.integers(0, 65536, (3, 5, 4), dtype=np.uint16)
    write_png(tmp_path / "a16.png", rgba16)
    oa16 = np.empty_like(rgba16)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "a16.png"), C.c_void_p(oa16.ctypes.data), oa16.nbytes, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
    assert (c.value, d.value) == (4, 16) and np.array_equal(oa16, rgba16[..., [2, 1, 0, 3]])
    # 16-bit PNG (utils.rs:110-117: imread(UNCHANGED) keeps the depth -> 16UC3 / 16UC1): how a 16-bit stack (BASELINE
    # configs[4]) arrives besides TIFF; samples byte-swapped to native order, never gamma-converted
    c16 = rng.integers(0, 65536, (9, 14, 3), dtype=np.uint16)
    g16 = rng.integers(0, 65536, (6, 4), dtype=np.uint16)
    for name, img in (("c16.png", c16), ("g16.png", g16)):
        write_png(tmp_path / name, img)
        assert lib.stk_imread(None, os.fsencode(tmp_path / name), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
        assert (h.value, w.value) == img.shape[:2] and c.value == (1 if img.ndim == 2 else 3) and d.value == 16
        o16 = np.empty_like(img)
        assert lib.stk_imread(None, os.fsencode(tmp_path / name), C.c_void_p(o16.ctypes.data), o16.nbytes, None, None, None, None) == 0
        assert np.array_equal(o16, img)
    # palette images decode to BGR, 1/2/4-bit grey expands to 8 bit (value as stored << nothing: expand_gray scales to 0..255)
    pal = rng.integers(0, 256, (5, 3), dtype=np.uint8)
    idx = rng.integers(0, 5, (7, 11), dtype=np.uint8)
    write_png(tmp_path / "p.png", idx, palette=pal)
    op = np.empty((7, 11, 3), np.uint8)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "p.png"), C.c_void_p(op.ctypes.data), op.nbytes, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
    assert (c.value, d.value) == (3, 8) and np.array_equal(op, pal[idx][..., ::-1])
    g2 = rng.integers(0, 4, (5, 9), dtype=np.uint8)
    write_png(tmp_path / "g2.png", g2, bits=2)
    og2 = np.empty((5, 9), np.uint8)
    assert lib.stk_imread(None, os.fsencode(tmp_path / "g2.png"), C.c_void_p(og2.ctypes.data), og2.nbytes, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
    assert (c.value, d.value) == (1, 8) and np.array_equal(og2, g2 * 85)                                  # 0..3 -> 0, 85, 170, 255
    data = (tmp_path / "c16.png").read_bytes()
    (tmp_path / "trunc.png").write_bytes(data[: len(data) - 40])
    assert lib.stk_imread(None, os.fsencode(tmp_path / "trunc.png"), C.c_void_p(o16.ctypes.data), 10 ** 6, None, None, None, None) == 4   # read error: BACKEND_ERROR
    (tmp_path / "bad.png").write_bytes(b"\x89PNG\r\n\x1a\n garbage")
    assert lib.stk_imread(None, os.fsencode(tmp_path / "bad.png"), None, 0, None, None, None, None) == 4     # BACKEND_ERROR


def test_imread_tiff_through_runtime_libtiff(tmp_path, write_tiff):
    """Stripped 8/16-bit grey / RGB TIFF via libtiff loaded at run time: what a 16-bit stack (BASELINE configs[4]) arrives as."""
    lib = _ffi.load()
    try:
        C.CDLL("libtiff.so.5")
    except OSError:
        pytest.skip("libtiff.so.5 is not installed here")
    rng = np.random.default_rng(2)
    for name, img in (("c16.tif", rng.integers(0, 65536, (9, 14, 3), dtype=np.uint16)), ("c8.tiff", rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)),
                      ("g16.tif", rng.integers(0, 65536, (6, 4), dtype=np.uint16)), ("g8.tif", rng.integers(0, 256, (3, 11), dtype=np.uint8))):
        write_tiff(tmp_path / name, img)
        w, h, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        assert lib.stk_imread(None, os.fsencode(tmp_path / name), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d)) == 0
        assert (h.value, w.value) == img.shape[:2] and c.value == (1 if img.ndim == 2 else 3) and d.value == img.dtype.itemsize * 8
        out = np.empty_like(img)
        assert lib.stk_imread(None, os.fsencode(tmp_path / name), C.c_void_p(out.ctypes.data), out.nbytes, None, None, None, None) == 0
        assert np.array_equal(out, img)
    (tmp_path / "bad.tif").write_bytes(b"II*\x00 not really a tiff")
    assert lib.stk_imread(None, os.fsencode(tmp_path / "bad.tif"), None, 0, None, None, None, None) == 4


def test_shard_arithmetic_of_the_library_matches_shard_py():
    """stk_shard_moving_frames (multi.cpp: the cut a multi-device context makes) == shard.py (the cut bench.py's
    one-process-per-GPU ranks make): contiguous, complete, sizes within one, earlier ranks larger. No GPU needed."""
    import ctypes as C
    from libstacker_rs_amd import _ffi, shard
    lib = _ffi.load()
    for n in list(range(1, 40)) + [64, 256, 257, 1024]:
        for world in (1, 2, 3, 4, 7, 8, 16):
            covered = []
            for rank in range(world):
                first, count = C.c_int32(-1), C.c_int32(-1)
                assert lib.stk_shard_moving_frames(n, world, rank, C.byref(first), C.byref(count)) == 0
                mine = shard.shard_moving_frames(n, world, rank)
                assert list(range(first.value, first.value + count.value)) == mine
                covered += mine
            assert covered == list(range(1, n))
    f, c = C.c_int32(), C.c_int32()
    assert lib.stk_shard_moving_frames(0, 2, 0, C.byref(f), C.byref(c)) == 2          # STK_INVALID_PARAMS
    assert lib.stk_shard_moving_frames(5, 2, 2, C.byref(f), C.byref(c)) == 2


def _conda_pillow_jpegs(tmp_path):
    """JPEG files written by the Pillow of the image's conda interpreter (the main interpreter has none) and Pillow's own
    decode of them; None where that interpreter is absent."""
    import subprocess
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        return None
    code = (
        "import numpy as np, sys\nfrom PIL import Image\nd = sys.argv[1]\nh, w = 123, 211\n"
        "yy, xx = np.mgrid[0:h, 0:w]\n"
        "img = np.stack([xx * 255 // w, yy * 255 // h, (xx + yy) * 3 % 256], -1).astype(np.uint8)\n"
        "img[30:60, 40:90] = [250, 10, 30]\n"
        "Image.fromarray(img).save(d + '/c444.jpg', quality=92, subsampling=0)\n"
        "Image.fromarray(img).save(d + '/c420.jpg', quality=85, subsampling=2)\n"
        "Image.fromarray(img).save(d + '/prog.jpg', quality=85, progressive=True, subsampling=0)\n"
        "Image.fromarray(img[..., 0]).save(d + '/grey.jpg', quality=90)\n"
        "np.save(d + '/src.npy', img)\n"
        "for n in ('c444', 'c420', 'prog', 'grey'):\n    np.save(d + '/' + n + '.npy', np.asarray(Image.open(d + '/' + n + '.jpg')))\n")
    r = subprocess.run([py, "-c", code, str(tmp_path)], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    return {n: np.load(tmp_path / (n + ".npy")) for n in ("src", "c444", "c420", "prog", "grey")}


def test_imread_jpeg_through_runtime_libjpeg(tmp_path):
    """JPEG (the reference's own data set: README.md:18) through libjpeg-turbo's libjpeg.so.8 loaded at run time, with the
    struct-size handshake and the SOF cross-check of imread.cpp. Pixels: equal to Pillow's decode where the two decoder
    families agree by construction (grey; colour without chroma subsampling: <= 1 level), same content otherwise (4:2:0:
    libjpeg-turbo — OpenCV's decoder — and Pillow's IJG 9 upsample chroma differently)."""
    lib = _ffi.load()
    ref = _conda_pillow_jpegs(tmp_path)
    if ref is None:
        pytest.skip("no Pillow to write test JPEGs with")

    def read(name, expect=0):
        w, h, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        st = lib.stk_imread(None, os.fsencode(tmp_path / name), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d))
        assert st == expect, (name, st)
        if st:
            return None
        out = np.zeros((h.value, w.value, c.value), np.uint8)
        assert lib.stk_imread(None, os.fsencode(tmp_path / name), C.c_void_p(out.ctypes.data), out.nbytes, None, None, None, None) == 0
        assert d.value == 8
        return out
    g = read("grey.jpg")
    assert g.shape == (123, 211, 1) and np.array_equal(g[..., 0], ref["grey"])                 # 8UC1, like imread(UNCHANGED)
    for name in ("c444", "prog"):
        c = read(name + ".jpg")
        assert c.shape == (123, 211, 3)
        assert np.abs(c[..., ::-1].astype(int) - ref[name].astype(int)).max() <= 1               # BGR in memory
    c420 = read("c420.jpg")
    assert np.abs(c420[..., ::-1].astype(int) - ref["src"].astype(int)).mean() < 4               # same picture, JPEG noise
    # a truncated file decodes with the missing part filled in (libjpeg warns and pads; so does OpenCV); garbage does not
    data = (tmp_path / "c444.jpg").read_bytes()
    (tmp_path / "trunc.jpg").write_bytes(data[: len(data) // 2])
    t = read("trunc.jpg")
    assert t.shape == (123, 211, 3) and np.array_equal(t[:8], read("c444.jpg")[:8])
    (tmp_path / "garbage.jpg").write_bytes(b"\xff\xd8" + bytes(range(256)) * 4)
    read("garbage.jpg", expect=4)                                                              # BACKEND_ERROR
    (tmp_path / "hdr_only.jpg").write_bytes(data[:30])
    read("hdr_only.jpg", expect=4)
    (tmp_path / "empty.jpg").write_bytes(b"")
    read("empty.jpg", expect=4)


def test_rust_ffi_is_generated_from_the_header():
    """rust/src/amd_ffi.rs (the `extern "C"` block, structs and constants of the Rust crate) is what tools/gen_rust_ffi.py
    prints for include/stacker.h today, and it declares every symbol of the header — the Rust side cannot be compiled in
    this container (no rustc), so this is the check that it has not drifted from the ABI."""
    import re
    import subprocess
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py")], capture_output=True, text=True, check=True).stdout
    committed = open(os.path.join(ROOT, "rust", "src", "amd_ffi.rs")).read()
    assert gen == committed, "run: python tools/gen_rust_ffi.py > rust/src/amd_ffi.rs"
    declared = sorted(re.findall(r"pub fn (stk_\w+)\(", committed))
    assert declared == _header_symbols()
    # the hand-written half only calls what the generated half declares
    used = set(re.findall(r"\b(stk_[a-z0-9_]+)\(", open(os.path.join(ROOT, "rust", "src", "amd.rs")).read()))
    assert used and used <= set(declared), used - set(declared)
    # struct layouts: same field order as the ctypes mirror (whose sizes test_struct_layouts_match_the_header checks against C)
    from libstacker_rs_amd import _ffi
    for cname, ctype in (("stk_keypoint_params", _ffi.KeypointParams), ("stk_ecc_params", _ffi.EccParams), ("stk_frames", _ffi.Frames),
                         ("stk_image_f32", _ffi.ImageF32)):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % cname, committed, re.S).group(1)
        assert re.findall(r"pub (\w+):", body) == [f[0] for f in ctype._fields_], cname


def _imread(lib, path):
    w, h, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    st = lib.stk_imread(None, os.fsencode(path), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d))
    if st:
        return st, None
    out = np.empty((h.value, w.value, c.value), np.uint8 if d.value == 8 else np.uint16)
    st = lib.stk_imread(None, os.fsencode(path), C.c_void_p(out.ctypes.data), out.nbytes, None, None, None, None)
    return st, (out[..., 0] if c.value == 1 else out)


def test_imread_tiled_and_rgba_tiff(tmp_path, write_tiff_tiled):
    """Round 4: tiled TIFF (TIFFReadTile; tiles overhanging the right and bottom edges) and RGBA TIFF (four channels, B G R A)."""
    try:
        C.CDLL("libtiff.so.5")
    except OSError:
        try:
            C.CDLL("libtiff.so.6")
        except OSError:
            pytest.skip("libtiff is not installed here")
    lib = _ffi.load()
    rng = np.random.default_rng(4)
    for name, img in (("t8.tif", rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)), ("t16.tif", rng.integers(0, 65536, (20, 45, 3), dtype=np.uint16)),
                      ("tg.tif", rng.integers(0, 256, (33, 17), dtype=np.uint8)), ("ta.tif", rng.integers(0, 256, (19, 40, 4), dtype=np.uint8))):
        write_tiff_tiled(tmp_path / name, img)
        st, got = _imread(lib, tmp_path / name)
        assert st == 0 and got.dtype == img.dtype and np.array_equal(got, img), name


def test_imread_bmp(tmp_path, write_bmp):
    """Round 4: BMP without a library — 24-bit (bottom-up and top-down, padded rows), 32-bit -> B G R A, 8-bit colour and grey palettes."""
    lib = _ffi.load()
    rng = np.random.default_rng(6)
    c24 = rng.integers(0, 256, (11, 13, 3), dtype=np.uint8)
    for td in (False, True):
        write_bmp(tmp_path / "c.bmp", c24, top_down=td)
        st, got = _imread(lib, tmp_path / "c.bmp")
        assert st == 0 and np.array_equal(got, c24)
    c32 = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    write_bmp(tmp_path / "a.bmp", c32)
    st, got = _imread(lib, tmp_path / "a.bmp")
    assert st == 0 and np.array_equal(got, c32)
    idx = rng.integers(0, 6, (9, 10), dtype=np.uint8)
    pal = rng.integers(0, 256, (6, 3), dtype=np.uint8)
    write_bmp(tmp_path / "p.bmp", idx, palette=pal)
    st, got = _imread(lib, tmp_path / "p.bmp")
    assert st == 0 and np.array_equal(got, pal[idx])
    gpal = np.repeat(rng.integers(0, 256, (6, 1), dtype=np.uint8), 3, axis=1)
    write_bmp(tmp_path / "g.bmp", idx, palette=gpal)
    st, got = _imread(lib, tmp_path / "g.bmp")
    assert st == 0 and got.ndim == 2 and np.array_equal(got, gpal[idx][..., 0])
    data = (tmp_path / "c.bmp").read_bytes()
    (tmp_path / "short.bmp").write_bytes(data[:100])
    assert _imread(lib, tmp_path / "short.bmp")[0] == 4                   # truncated: BACKEND_ERROR
    (tmp_path / "rle.bmp").write_bytes(data[:30] + (1).to_bytes(4, "little") + data[34:])
    assert _imread(lib, tmp_path / "rle.bmp")[0] == 7                     # RLE8 flag on a 24-bit file: a flavour not taken


def _webp_encoder():
    try:
        w = C.CDLL("libwebp.so.7")
    except OSError:
        pytest.skip("libwebp.so.7 is not on this machine")
    for name in ("WebPEncodeLosslessBGR", "WebPEncodeLosslessBGRA"):
        f = getattr(w, name)
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    w.WebPEncodeBGR.restype = C.c_size_t
    w.WebPEncodeBGR.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_void_p)]
    w.WebPDecodeBGR.restype = C.c_void_p
    w.WebPDecodeBGR.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    w.WebPFree.argtypes = [C.c_void_p]
    return w


def test_imread_webp_through_runtime_libwebp(tmp_path):
    """Round 4 (VERDICT r3 'missing' 6): still WebP. Lossless files must come back as the very pixels that were encoded (BGR in
    memory; four channels when the bitstream has alpha, like imread(UNCHANGED)); a lossy file as libwebp's own
    WebPDecodeBGR gives it — the call OpenCV's decoder makes."""
    lib = _ffi.load()
    w = _webp_encoder()
    rng = np.random.default_rng(5)

    def encode(fn, img, *extra):
        out = C.c_void_p()
        n = fn(C.c_void_p(img.ctypes.data), img.shape[1], img.shape[0], img.strides[0], *extra, C.byref(out))
        assert n > 0
        data = C.string_at(out, n)
        w.WebPFree(out)
        return data

    bgr = np.ascontiguousarray(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8))
    (tmp_path / "c.webp").write_bytes(encode(w.WebPEncodeLosslessBGR, bgr))
    st, got = _imread(lib, tmp_path / "c.webp")
    assert st == 0 and got.shape == bgr.shape and np.array_equal(got, bgr)
    bgra = np.ascontiguousarray(rng.integers(1, 256, (29, 41, 4), dtype=np.uint8))      # alpha >= 1: a lossless encoder may rewrite invisible pixels
    (tmp_path / "a.webp").write_bytes(encode(w.WebPEncodeLosslessBGRA, bgra))
    st, got = _imread(lib, tmp_path / "a.webp")
    assert st == 0 and got.shape == bgra.shape and np.array_equal(got, bgra)
    smooth = np.ascontiguousarray((np.add.outer(np.arange(64), np.arange(96))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8))
    lossy = encode(w.WebPEncodeBGR, smooth, C.c_float(80.0))
    (tmp_path / "l.webp").write_bytes(lossy)
    ww, hh = C.c_int(), C.c_int()
    ptr = w.WebPDecodeBGR(lossy, len(lossy), C.byref(ww), C.byref(hh))
    ref = np.frombuffer(C.string_at(ptr, ww.value * hh.value * 3), np.uint8).reshape(hh.value, ww.value, 3).copy()
    w.WebPFree(ptr)
    st, got = _imread(lib, tmp_path / "l.webp")
    assert st == 0 and np.array_equal(got, ref) and np.abs(got.astype(int) - smooth).mean() < 8
    (tmp_path / "short.webp").write_bytes(lossy[:40])
    assert _imread(lib, tmp_path / "short.webp")[0] == 4                  # truncated: BACKEND_ERROR
    (tmp_path / "bad.webp").write_bytes(b"RIFF\x10\x00\x00\x00WEBPnope" + bytes(16))
    assert _imread(lib, tmp_path / "bad.webp")[0] == 4


def _cmyk_jpeg(blocks):
    """A baseline JPEG by hand: four components (Adobe APP14, transform 0 = CMYK), no subsampling, quantisation table of ones,
    DC-only 8 x 8 blocks. `blocks`: (by, bx, 4) uint8 — the C, M, Y, K sample every pixel of a block decodes to (DC = 8 (v - 128),
    libjpeg's IDCT returns (DC + 4) >> 3 + 128 for a DC-only block). Huffman tables: twelve 4-bit DC codes, one 1-bit AC code (EOB)."""
    by, bx, _ = blocks.shape
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
    out += b"\xff\xdb\x00\x43\x00" + bytes([1] * 64)
    out += b"\xff\xc0\x00\x14\x08" + (8 * by).to_bytes(2, "big") + (8 * bx).to_bytes(2, "big") + b"\x04" + b"".join(bytes([c, 0x11, 0]) for c in (1, 2, 3, 4))
    out += b"\xff\xc4\x00\x1f\x00" + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12))
    out += b"\xff\xc4\x00\x14\x10" + bytes([1] + [0] * 15) + b"\x00"
    out += b"\xff\xda\x00\x0e\x04" + b"".join(bytes([c, 0x00]) for c in (1, 2, 3, 4)) + b"\x00\x3f\x00"
    bits = []
    pred = [0, 0, 0, 0]
    for y in range(by):
        for x in range(bx):
            for c in range(4):
                dc = 8 * (int(blocks[y, x, c]) - 128)
                diff, pred[c] = dc - pred[c], dc
                cat = abs(diff).bit_length()
                bits += [(cat >> k) & 1 for k in (3, 2, 1, 0)]                      # symbol `cat`: the cat-th 4-bit code
                extra = diff if diff >= 0 else diff + (1 << cat) - 1
                bits += [(extra >> k) & 1 for k in range(cat - 1, -1, -1)]
                bits.append(0)                                                      # EOB
    bits += [1] * (-len(bits) % 8)
    for i in range(0, len(bits), 8):
        b = int("".join(map(str, bits[i:i + 8])), 2)
        out.append(b)
        if b == 0xff:
            out.append(0)
    return bytes(out + b"\xff\xd9")


def test_imread_cmyk_jpeg(tmp_path):
    """Round 4 (VERDICT r3 'missing' 6): a four-component JPEG comes out as B G R through OpenCV's own CMYK conversion [OCV-RECALL:
    JpegDecoder asks the library for JCS_CMYK, then icvCvt_CMYK2BGR_8u_C4C3R: x' = k - ((255 - x) * k >> 8)] — closed form on a
    hand-made file whose blocks decode to known C, M, Y, K samples (no encoder in this environment writes CMYK)."""
    lib = _ffi.load()
    rng = np.random.default_rng(8)
    blocks = rng.integers(0, 256, (3, 5, 4), dtype=np.uint8)
    blocks[0, 0] = (255, 255, 255, 255); blocks[0, 1] = (0, 0, 0, 0); blocks[0, 2] = (0, 128, 255, 200)
    (tmp_path / "cmyk.jpg").write_bytes(_cmyk_jpeg(blocks))
    st, got = _imread(lib, tmp_path / "cmyk.jpg")
    if st == 7:
        pytest.skip("libjpeg.so.8 is not on this machine")
    assert st == 0 and got.shape == (24, 40, 3) and got.dtype == np.uint8
    b = blocks.astype(np.int64)
    k = b[..., 3:4]
    cmy = k - (((255 - b[..., :3]) * k) >> 8)                                       # c', m', y'
    want = np.repeat(np.repeat(cmy[..., ::-1], 8, axis=0), 8, axis=1).astype(np.uint8)   # B G R = y' m' c'
    assert np.array_equal(got, want)
    assert tuple(got[0, 0]) == (255, 255, 255) and tuple(got[0, 8]) == (0, 0, 0)    # "no ink" (inverted CMYK) is white, full ink black
