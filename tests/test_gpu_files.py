"""The file front-end: imread of binary PNM and the path-based entry points, against the frame-based ones."""
import numpy as np
import pytest

from libstacker_rs_amd import (EccMatchParameters, KeyPointMatchParameters, MotionType, NotEnoughFiles, NotImplementedYet,
                               OpenCvError, RANSAC, synth)

pytestmark = pytest.mark.gpu


def write_pnm(path, img):
    """P6 (HxWx3 BGR in, RGB on disk) or P5 (HxW), 8 or 16 bit (big-endian), with a comment line in the header."""
    a = np.asarray(img)
    maxval = 255 if a.dtype == np.uint8 else 65535
    if a.ndim == 3:
        a = a[..., ::-1]
        head = f"P6\n# written by the test\n{a.shape[1]} {a.shape[0]}\n{maxval}\n"
    else:
        head = f"P5\n{a.shape[1]} {a.shape[0]}\n{maxval}\n"
    raster = np.ascontiguousarray(a).astype(">u2").tobytes() if maxval == 65535 else np.ascontiguousarray(a).tobytes()
    with open(path, "wb") as f:
        f.write(head.encode() + raster)


def test_imread_pnm_variants(stacker, tmp_path):
    rng = np.random.default_rng(0)
    for name, img in (("c8.ppm", rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)),
                      ("g8.pgm", rng.integers(0, 256, (20, 31), dtype=np.uint8)),
                      ("c16.ppm", rng.integers(0, 65536, (11, 17, 3), dtype=np.uint16)),
                      ("g16.pgm", rng.integers(0, 65536, (9, 5), dtype=np.uint16))):
        p = tmp_path / name
        write_pnm(p, img)
        got = stacker.imread(p)
        assert got.dtype == img.dtype and np.array_equal(got, img)
    with pytest.raises(OpenCvError):
        stacker.imread(tmp_path / "missing.ppm")                   # empty Mat -> cvtColor raises in the reference
    (tmp_path / "junk.ppm").write_bytes(b"not an image at all")
    with pytest.raises(OpenCvError):
        stacker.imread(tmp_path / "junk.ppm")
    (tmp_path / "short.ppm").write_bytes(b"P6\n4 4\n255\n\x00\x01")
    with pytest.raises(OpenCvError):
        stacker.imread(tmp_path / "short.ppm")
    with pytest.raises(OpenCvError):
        stacker.imread(tmp_path / "photo.jpg")                     # missing file: empty Mat -> cvtColor raises
    with pytest.raises(NotImplementedYet):
        stacker.imread(tmp_path / "photo.exr")                     # no codec in this build


def test_path_based_entry_points_equal_frame_based(stacker, tmp_path):
    frames, _ = synth.make_stack(4, 320, 240)
    fr = frames.numpy()
    paths = []
    for i, f in enumerate(fr):
        paths.append(tmp_path / f"frame_{i:02d}.ppm")
        write_pnm(paths[-1], f)
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    d_f, out_f = stacker.keypoint_match_files(paths, kp)
    d_a, out_a = stacker.keypoint_match(list(fr), kp)
    assert d_f == d_a == 0 and np.array_equal(out_f, out_a)
    assert np.array_equal(stacker.ecc_match_files(paths, ecc), stacker.ecc_match(list(fr), ecc))
    with pytest.raises(NotEnoughFiles):
        stacker.ecc_match_files([], ecc)
    # a grey file in the list: cvtColor(BGR2GRAY) on a 1-channel Mat raises in the reference (utils.rs:136)
    write_pnm(tmp_path / "grey.pgm", fr[0][..., 0])
    with pytest.raises(OpenCvError):
        stacker.ecc_match_files([tmp_path / "grey.pgm", tmp_path / "grey.pgm"], ecc)
    # 16-bit colour files: ORB asserts 8-bit, findTransformECC rejects 16UC1
    write_pnm(tmp_path / "a16.ppm", fr[0].astype(np.uint16) * 257)
    write_pnm(tmp_path / "b16.ppm", fr[1].astype(np.uint16) * 257)
    with pytest.raises(OpenCvError):
        stacker.keypoint_match_files([tmp_path / "a16.ppm", tmp_path / "b16.ppm"], kp)
    with pytest.raises(OpenCvError):
        stacker.ecc_match_files([tmp_path / "a16.ppm", tmp_path / "b16.ppm"], ecc)


def test_png_paths_equal_frame_based(stacker, tmp_path, write_png):
    # 8-bit RGB PNG through the run-time libpng (where it can be loaded): same result as handing the frames over
    import ctypes
    try:
        ctypes.CDLL("libpng16.so.16")
    except OSError:
        pytest.skip("libpng16.so.16 is not installed here")
    frames, _ = synth.make_stack(3, 320, 240)
    fr = frames.numpy()
    paths = []
    for i, f in enumerate(fr):
        paths.append(tmp_path / f"frame_{i}.png")
        write_png(paths[-1], f)
    assert np.array_equal(stacker.imread(paths[1]), fr[1])
    ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    assert np.array_equal(stacker.ecc_match_files(paths, ecc), stacker.ecc_match(list(fr), ecc))


def test_webp_paths_equal_frame_based(stacker, tmp_path):
    # lossless WebP through the run-time libwebp (round 4): the decoded frames are the encoded ones, so the path-based call equals
    # the frame-based one; a stack with alpha gives a four-channel result
    import ctypes as C
    try:
        wl = C.CDLL("libwebp.so.7")
    except OSError:
        pytest.skip("libwebp.so.7 is not installed here")
    wl.WebPFree.argtypes = [C.c_void_p]

    def write_webp(path, img):
        fn = wl.WebPEncodeLosslessBGRA if img.shape[2] == 4 else wl.WebPEncodeLosslessBGR
        fn.restype = C.c_size_t
        img = np.ascontiguousarray(img)
        out = C.c_void_p()
        n = fn(C.c_void_p(img.ctypes.data), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.c_int(img.strides[0]), C.byref(out))
        assert n > 0
        path.write_bytes(C.string_at(out, n))
        wl.WebPFree(out)

    frames, _ = synth.make_stack(3, 320, 240)
    fr = frames.numpy()
    paths = []
    for i, f in enumerate(fr):
        paths.append(tmp_path / f"frame_{i}.webp")
        write_webp(paths[-1], f)
    assert np.array_equal(stacker.imread(paths[1]), fr[1])
    ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    assert np.array_equal(stacker.ecc_match_files(paths, ecc), stacker.ecc_match(list(fr), ecc))
    d_f, out_f = stacker.keypoint_match_files(paths, kp)
    d_a, out_a = stacker.keypoint_match(list(fr), kp)
    assert d_f == d_a and np.array_equal(out_f, out_a)
    alpha = np.full(fr.shape[:3] + (1,), 200, np.uint8)
    fa = np.concatenate([fr, alpha], axis=3)
    pa = []
    for i, f in enumerate(fa):
        pa.append(tmp_path / f"alpha_{i}.webp")
        write_webp(pa[-1], f)
    out4 = stacker.ecc_match_files(pa, ecc)
    assert out4.shape == (240, 320, 4) and np.array_equal(out4, stacker.ecc_match(list(fa), ecc))


def test_16bit_tiff_stack_through_hybrid_match_files(stacker, tmp_path, write_tiff):
    # BASELINE configs[4] end to end: a 16-bit TIFF stack on disk -> ORB-seeded ECC -> f32 image
    import ctypes
    try:
        ctypes.CDLL("libtiff.so.5")
    except OSError:
        pytest.skip("libtiff.so.5 is not installed here")
    frames, _ = synth.make_stack(3, 320, 240)
    f16 = frames.numpy().astype(np.uint16) * 257 + 3
    paths = []
    for i, f in enumerate(f16):
        paths.append(tmp_path / f"frame_{i}.tif")
        write_tiff(paths[-1], f)
    assert np.array_equal(stacker.imread(paths[2]), f16[2])
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    ecc = EccMatchParameters(MotionType.Homography, 200, 1e-5, 5)
    assert np.array_equal(stacker.hybrid_match_files(paths, kp, ecc), stacker.hybrid_match(list(f16), kp, ecc))
    with pytest.raises(OpenCvError):                        # the reference's own entry points reject 16-bit stacks
        stacker.ecc_match_files(paths, ecc)


def test_host_fed_pipeline_equals_device_resident(stacker):
    """Frames handed over in host memory cross PCIe in batches while earlier batches are prepared and aligned (copy
    stream -> prep stream -> gated ECC queue / batched ORB); results must not depend on where the frames were or on how
    the batches fell: bit-identical to the device-resident run, pinned or pageable, any batch size."""
    import torch
    from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, synth
    ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    frames, _ = synth.make_stack(21, 320, 240)
    dev = frames.cuda()
    ref_e, st_e = stacker.ecc_match(dev, ecc, return_stats=True)
    ref_k = stacker.keypoint_match(dev[:8], kp)
    pinned = frames.pin_memory()
    try:
        for batch in (1, 3, 8, 64):
            stacker.set_option("upload_batch", batch)
            for src in (list(frames.numpy()), pinned):
                out, st = stacker.ecc_match(src, ecc, return_stats=True)
                assert np.array_equal(out if isinstance(out, np.ndarray) else out.cpu().numpy(), ref_e.cpu().numpy())
                assert [s["iterations"] for s in st] == [s["iterations"] for s in st_e]
            d, outk = stacker.keypoint_match(list(frames.numpy()[:8]), kp)
            assert d == ref_k[0] and np.array_equal(outk, ref_k[1].cpu().numpy())
        t = stacker.timing()
        assert t["h2d_bytes"] == 8 * 320 * 240 * 3 and t["h2d_ms"] > 0
        # scale-down variant and a single frame through the same route
        a = stacker.ecc_match(list(frames.numpy()[:5]), ecc, scale_down_width=200.0)
        b = stacker.ecc_match(dev[:5], ecc, scale_down_width=200.0).cpu().numpy()
        assert np.array_equal(a, b)
        assert np.array_equal(stacker.ecc_match([frames.numpy()[0]], ecc), stacker.ecc_match(dev[:1], ecc).cpu().numpy())
        # eps > 0.5: OpenCV's loop test fails before the first iteration (rho = -1, last_rho = -eps): identity warps
        o0, s0 = stacker.ecc_match(list(frames.numpy()[:3]), EccMatchParameters(MotionType.Homography, 50, 2.0, 5), return_stats=True)
        assert all(s["iterations"] == 0 for s in s0) and np.array_equal(s0[1]["warp"], np.eye(3))
    finally:
        stacker.set_option("upload_batch", 8)


def test_files_are_decoded_in_parallel_into_pinned_memory(stacker, tmp_path):
    """*_match_files decodes frames 1..n-1 on a pool of host threads straight into one page-locked block (the reference
    decodes inside its Rayon fold, lib.rs:200, 756): same image as the frame-based call on the same pixels; with several
    unreadable files the one reported is the FIRST in list order, whichever thread met it; a JPEG stack (the reference's
    own data set is JPEG) goes through the same route."""
    import os, subprocess
    frames, _ = synth.make_stack(24, 320, 240)
    fr = frames.numpy()
    paths = []
    for i, f in enumerate(fr):
        paths.append(tmp_path / f"f{i:03d}.ppm")
        write_pnm(paths[-1], f)
    ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    assert np.array_equal(stacker.ecc_match_files(paths, ecc), stacker.ecc_match(list(fr), ecc))
    bad = list(paths)
    bad[17] = tmp_path / "missing_17.ppm"
    bad[5] = tmp_path / "missing_05.ppm"
    with pytest.raises(OpenCvError) as ei:
        stacker.ecc_match_files(bad, ecc)
    assert "missing_05" in str(ei.value)
    small = tmp_path / "small.ppm"
    write_pnm(small, fr[0][:100, :100].copy())
    with pytest.raises(Exception) as ei:
        stacker.ecc_match_files(paths[:3] + [small] + paths[3:], ecc)
    assert "differs in size" in str(ei.value)
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        return
    np.save(tmp_path / "stack.npy", fr[:6])
    # (round 4: the snippet used to carry literal backslash-n sequences, was a SyntaxError for the side interpreter, and this
    # half of the test returned early without ever running)
    code = ("import numpy as np, sys\nfrom PIL import Image\nd = sys.argv[1]\na = np.load(d + '/stack.npy')\n"
            "for i, f in enumerate(a):\n    Image.fromarray(f[..., ::-1]).save(d + '/j%02d.jpg' % i, quality=95, subsampling=0)\n")
    r = subprocess.run([py, "-c", code, str(tmp_path)], capture_output=True)
    if r.returncode != 0:
        assert b"No module named" in r.stderr, r.stderr          # no Pillow in that interpreter: nothing to write JPEGs with
        return
    jpaths = [tmp_path / ("j%02d.jpg" % i) for i in range(6)]
    decoded = [stacker.imread(p) for p in jpaths]
    assert decoded[0].shape == (240, 320, 3) and np.abs(decoded[1].astype(int) - fr[1].astype(int)).mean() < 3
    assert np.array_equal(stacker.ecc_match_files(jpaths, ecc), stacker.ecc_match(decoded, ecc))
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    d_f, out_f = stacker.keypoint_match_files(jpaths, kp)
    d_a, out_a = stacker.keypoint_match(decoded, kp)
    assert d_f == d_a and np.array_equal(out_f, out_a)
