"""One context over several devices (stk_create_multi): shard + per-device host threads + reduce + finalize inside the
library. The box has one GPU, so the devices are [0, 0] (members share the card, the accumulators are added locally);
the RCCL leg is exercised by stk_rccl_selftest (library load, communicator, a verified ncclReduce on a 1-rank group).
The real multi-GPU reduce has not run anywhere yet — unmeasured."""
import numpy as np
import pytest
import torch

from libstacker_rs_amd import (EccMatchParameters, KeyPointMatchParameters, MotionType, OpenCvError, RANSAC, Stacker, synth)

pytestmark = pytest.mark.gpu
ECC = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
KP = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)


@pytest.fixture(scope="module")
def multi():
    s = Stacker(devices=[0, 0])
    yield s
    s.close()


def test_rccl_selftest(stacker, multi):
    stacker.rccl_selftest(1 << 20)                       # dlopen(librccl.so.1) + 1-rank communicator + checked ncclReduce
    Stacker(devices=[0]).rccl_selftest(4096)             # n_devices == 1 is the plain context


def test_multi_ecc_equals_single(stacker, multi):
    # every frame is summed over a fixed workgroup partition (ecc_plan), whatever shares the launch with it: its warp is
    # bit-identical to the single-device run for any split of the stack
    frames, _ = synth.make_stack(33, 320, 240)
    fr = list(frames.numpy())
    one, s1 = stacker.ecc_match(fr, ECC, return_stats=True)
    two, s2 = multi.ecc_match(fr, ECC, return_stats=True)
    for a, b in zip(s1, s2):
        assert np.array_equal(a["warp"], b["warp"]) and a["iterations"] == b["iterations"]
    assert np.max(np.abs(one - two)) <= 1e-6             # the order of the f32 adds
    dev = multi.ecc_match(torch.from_numpy(frames.numpy()).cuda(), ECC)      # device-resident frames
    assert np.array_equal(dev.cpu().numpy(), two)


def test_multi_keypoint_and_hybrid_equal_single(stacker, multi):
    frames, _ = synth.make_stack(6, 640, 480)
    fr = list(frames.numpy())
    fr[3] = np.full_like(fr[3], 128)                     # dropped by the member that owns it
    d1, one, s1 = stacker.keypoint_match(fr, KP, return_stats=True)
    d2, two, s2 = multi.keypoint_match(fr, KP, return_stats=True)
    assert d1 == d2 == 1 and [s["status"] for s in s1] == [s["status"] for s in s2]
    for a, b in zip(s1, s2):
        assert np.array_equal(a["warp"], b["warp"])
    assert np.max(np.abs(one - two)) <= 1e-6             # divisor n - dropped on both sides
    good = [f for i, f in enumerate(fr) if i != 3]
    h1 = stacker.hybrid_match(good, KP, EccMatchParameters(MotionType.Homography, 200, 1e-5, 5))
    h2 = multi.hybrid_match(good, KP, EccMatchParameters(MotionType.Homography, 200, 1e-5, 5))
    assert np.max(np.abs(h1 - h2)) <= 1e-6               # per-frame results identical; the order of the f32 adds differs


def test_multi_context_takes_frames_of_differing_size_on_its_first_device(stacker, multi):
    # what the Rust shim's one shared context (rust/src/amd.rs) does with a stack of mixed sizes: the frame-by-frame route on member 0
    frames, _ = synth.make_stack(4, 640, 480)
    fr = frames.numpy()
    mixed = [fr[0], np.ascontiguousarray(fr[1][:400, :600]), np.ascontiguousarray(fr[2][:470, :520]), np.ascontiguousarray(fr[3][:333, :639])]
    d1, one = stacker.keypoint_match(mixed, KP)
    d2, two = multi.keypoint_match(mixed, KP)
    assert d1 == d2 and np.array_equal(one, two)


def test_multi_more_devices_than_frames_and_errors(stacker, multi):
    frames, _ = synth.make_stack(2, 320, 240)
    fr = list(frames.numpy())
    assert np.max(np.abs(multi.ecc_match(fr, ECC) - stacker.ecc_match(fr, ECC))) <= 1e-6      # member 1 has nothing to do
    assert np.array_equal(multi.ecc_match(fr[:1], ECC), stacker.ecc_match(fr[:1], ECC))
    bad = [fr[0], np.full_like(fr[0], 7)]                # constant frame: ECC cannot correlate -> the whole call fails
    with pytest.raises(OpenCvError):
        multi.ecc_match(bad, ECC)
    multi.set_option("ecc_chunk", 4)                     # options reach every member


def test_bound_and_unbound_contexts_side_by_side(stacker):
    # one context on torch's stream, one on its own: both must see finished inputs (the binding is per instance)
    frames, _ = synth.make_stack(3, 320, 240)
    ref = stacker.ecc_match(list(frames.numpy()), ECC)
    bound = Stacker(0)
    bound.use_torch_stream()
    try:
        for _ in range(3):
            dev = torch.from_numpy(frames.numpy()).cuda(non_blocking=True) + 0      # produced on torch's current stream
            a = bound.ecc_match(dev, ECC)
            b = stacker.ecc_match(dev, ECC)                                         # unbound: drains the producer first
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):                                           # bound stream != current stream now
                dev2 = torch.from_numpy(frames.numpy()).cuda(non_blocking=True) + 0
                c = bound.ecc_match(dev2, ECC)
            side.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(a.cpu().numpy(), ref) and np.array_equal(b.cpu().numpy(), ref) and np.array_equal(c.cpu().numpy(), ref)
    finally:
        bound.close()


def test_multi_context_on_files(stacker, multi, tmp_path):
    # the path-based entry point on a multi-device context: frames are still being decoded by the thread pool when the
    # members' uploaders ask for them (one gate shared by every member)
    frames, _ = synth.make_stack(13, 320, 240)
    paths = []
    for i, f in enumerate(frames.numpy()):
        p = tmp_path / f"m{i:02d}.ppm"
        p.write_bytes(b"P6\n320 240\n255\n" + np.ascontiguousarray(f[..., ::-1]).tobytes())
        paths.append(p)
    one = stacker.ecc_match_files(paths, ECC)
    two = multi.ecc_match_files(paths, ECC)
    assert np.max(np.abs(one - two)) <= 1e-6
    d1, k1 = stacker.keypoint_match_files(paths, KP)
    d2, k2 = multi.keypoint_match_files(paths, KP)
    assert d1 == d2 and np.max(np.abs(k1 - k2)) <= 1e-6


def test_create_multi_with_a_device_the_node_does_not_have():
    # member creation failing half-way used to index the not-yet-sized accumulator vectors in the clean-up
    from libstacker_rs_amd import HipError
    for ids in ([0, 9999], [0, 0, 9999], [9999, 0]):
        with pytest.raises(HipError):
            Stacker(devices=ids)
    s = Stacker(devices=[0, 0])                          # and the library is still usable afterwards
    s.close()
