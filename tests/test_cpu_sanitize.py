"""CPU sanitizer builds: the oracle under ASan + UBSan over its own test file, the product's host-side
parsers (stk_imread: hand-written PNM parser, hand-declared libpng / libjpeg structs) under ASan + UBSan over golden,
truncated and garbage files, the host half of findHomography under ASan + UBSan, and the keypoint path's thread pool under
TSan. CPU builds only — GPU AddressSanitizer is not available on the pool."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
INC = ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include")]


def _build(src, out, flags):
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer"] + flags + INC + [os.path.join(SAN, src), "-o", out, "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.timeout(600)
def test_imread_under_asan_ubsan(tmp_path, write_png, write_tiff, write_tiff_tiled, write_bmp):
    exe = str(tmp_path / "imread_harness")
    _build("imread_harness.cpp", exe, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    files = []
    good = tmp_path / "ok.ppm"
    good.write_bytes(b"P6\n# comment\n23 17\n255\n" + img.tobytes())
    files.append(good)
    g16 = tmp_path / "ok16.pgm"
    g16.write_bytes(b"P5 5 4 65535\n" + rng.integers(0, 65536, (4, 5), dtype=np.uint16).astype(">u2").tobytes())
    files.append(g16)
    write_png(tmp_path / "ok.png", img); files.append(tmp_path / "ok.png")
    write_tiff(tmp_path / "ok.tif", img); files.append(tmp_path / "ok.tif")
    write_tiff(tmp_path / "ok16.tif", rng.integers(0, 65536, (6, 7, 3), dtype=np.uint16)); files.append(tmp_path / "ok16.tif")
    # round 4: tiled and RGBA TIFF, RGBA PNG, BMP (24-bit, 32-bit, palette)
    write_tiff_tiled(tmp_path / "tiled.tif", img); files.append(tmp_path / "tiled.tif")
    write_tiff_tiled(tmp_path / "tiled_a.tif", rng.integers(0, 256, (9, 21, 4), dtype=np.uint8)); files.append(tmp_path / "tiled_a.tif")
    write_png(tmp_path / "rgba.png", rng.integers(0, 256, (5, 6, 4), dtype=np.uint8)); files.append(tmp_path / "rgba.png")
    write_bmp(tmp_path / "ok.bmp", img); files.append(tmp_path / "ok.bmp")
    write_bmp(tmp_path / "ok32.bmp", rng.integers(0, 256, (4, 5, 4), dtype=np.uint8)); files.append(tmp_path / "ok32.bmp")
    write_bmp(tmp_path / "okp.bmp", rng.integers(0, 5, (6, 9), dtype=np.uint8), palette=rng.integers(0, 256, (5, 3), dtype=np.uint8)); files.append(tmp_path / "okp.bmp")
    try:                                                   # still WebP (lossless colour, lossless with alpha, lossy), when the machine has the library
        import ctypes as C
        wl = C.CDLL("libwebp.so.7")
        for name, arr, fn, extra in (("ok.webp", img, "WebPEncodeLosslessBGR", ()), ("oka.webp", rng.integers(1, 256, (7, 9, 4), dtype=np.uint8), "WebPEncodeLosslessBGRA", ()),
                                     ("okl.webp", img, "WebPEncodeBGR", (C.c_float(70.0),))):
            f = getattr(wl, fn); f.restype = C.c_size_t
            arr = np.ascontiguousarray(arr)
            out = C.c_void_p()
            n = f(C.c_void_p(arr.ctypes.data), C.c_int(arr.shape[1]), C.c_int(arr.shape[0]), C.c_int(arr.strides[0]), *extra, C.byref(out))
            assert n > 0
            (tmp_path / name).write_bytes(C.string_at(out, n)); files.append(tmp_path / name)
            wl.WebPFree.argtypes = [C.c_void_p]; wl.WebPFree(out)
    except OSError:
        pass
    from test_cpu_host import _cmyk_jpeg                   # a four-component JPEG (hand-made: Adobe CMYK, DC-only blocks)
    (tmp_path / "cmyk.jpg").write_bytes(_cmyk_jpeg(rng.integers(0, 256, (2, 3, 4), dtype=np.uint8))); files.append(tmp_path / "cmyk.jpg")
    n_good = len(files)
    # truncations at every interesting place and byte-level garbage of each format
    for src in list(files):
        data = src.read_bytes()
        for cut in (0, 1, 2, 3, 7, 11, 15, len(data) // 3, len(data) // 2, len(data) - 1):
            p = tmp_path / f"cut{cut}_{src.name}"
            p.write_bytes(data[:cut]); files.append(p)
        for k in range(6):
            b = bytearray(data)
            for pos in rng.integers(0, min(len(b), 64), 8):
                b[pos] = int(rng.integers(0, 256))
            p = tmp_path / f"fuzz{k}_{src.name}"
            p.write_bytes(bytes(b)); files.append(p)
    for name, blob in (("huge.ppm", b"P6\n99999999 99999999\n255\n"), ("neg.ppm", b"P6\n-4 4\n255\n" + b"0" * 64),
                       ("maxval.ppm", b"P6\n2 2\n1023\n" + b"0" * 64), ("nohdr.ppm", b"P6"), ("cmt.ppm", b"P6\n#" + b"x" * 300),
                       ("big_dims_small_data.pgm", b"P5\n1000 1000\n255\n" + b"1" * 10), ("noise.jpg", bytes(rng.integers(0, 256, 500, dtype=np.uint8))),
                       ("soi_only.jpg", b"\xff\xd8"), ("sof_lies.jpg", b"\xff\xd8\xff\xc0\x00\x11\x08\xff\xff\xff\xff\x03" + b"\x01\x11\x00" * 3 + b"\xff\xda")):
        p = tmp_path / name
        p.write_bytes(blob); files.append(p)
    jp = "/opt/conda/bin/python3.9"
    if os.path.exists(jp):
        r = subprocess.run([jp, "-c", "import numpy as np, sys\nfrom PIL import Image\nImage.fromarray((np.arange(64 * 48 * 3) % 251).astype(np.uint8).reshape(48, 64, 3)).save(sys.argv[1], quality=90)", str(tmp_path / "ok.jpg")])
        if r.returncode == 0:
            data = (tmp_path / "ok.jpg").read_bytes()
            files.append(tmp_path / "ok.jpg")
            for cut in (2, 4, 20, 100, len(data) // 2, len(data) - 2):
                p = tmp_path / f"cut{cut}_ok.jpg"
                p.write_bytes(data[:cut]); files.append(p)
            for k in range(12):
                b = bytearray(data)
                for pos in rng.integers(0, len(b), 6):
                    b[pos] = int(rng.integers(0, 256))
                p = tmp_path / f"fuzz{k}_ok.jpg"
                p.write_bytes(bytes(b)); files.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + [str(f) for f in files], capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    assert all(l.split()[0] == "0" for l in lines[:n_good])                   # the intact files decode
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.timeout(600)
def test_host_pool_under_tsan(tmp_path):
    exe = str(tmp_path / "hostpool_tsan")
    _build("hostpool_tsan.cpp", exe, ["-fsanitize=thread"])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), timeout=500)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-500:] + r.stderr[-3000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr


@pytest.mark.timeout(600)
def test_async_upload_under_tsan(tmp_path):
    exe = str(tmp_path / "upload_tsan")
    _build("upload_tsan.cpp", exe, ["-fsanitize=thread"])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), timeout=500)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-500:] + r.stderr[-3000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr


@pytest.mark.timeout(600)
def test_ransac_host_half_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "ransac_host")
    _build("ransac_host_harness.cpp", exe, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), timeout=500)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-500:] + r.stderr[-3000:]
    # cv::RNG(-1): the first two outputs modulo 1000, from the recurrence state = lo * 4164903690 + hi
    st = 0xFFFFFFFFFFFFFFFF
    exp = []
    for _ in range(2):
        st = ((st & 0xFFFFFFFF) * 4164903690 + (st >> 32)) & 0xFFFFFFFFFFFFFFFF
        exp.append((st & 0xFFFFFFFF) % 1000)
    assert [int(x) for x in r.stdout.split()[2:4]] == exp


@pytest.mark.timeout(900)
def test_oracle_under_asan_ubsan(tmp_path):
    """`make -C oracle asan` and the oracle's own CPU test file run against that build (LD_PRELOAD of libasan, the
    interpreter itself is not instrumented)."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "-s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=libasan + (":" + libubsan if os.path.exists(libubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", STACKER_ORACLE_LIB=os.path.join(ROOT, "oracle", "_asan", "liboracle.so"),
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_cpu_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr
