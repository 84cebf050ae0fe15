"""Dataset-record decode and checkpoint-prefix handling.

CPU: the oracle restatement (oracle/records.py) against tests/golden/records.npz, which holds the outputs of the
reference's own `CustomImageDataset._deserialize_datapoint` (customDatasets/datasets.py:92-135); host validation;
checkpoint key normalisation.  GPU: `hipseg_decode_records` through the C ABI, bit-exact against the oracle."""
import numpy as np
import pytest
import torch

from oracle import records


@pytest.fixture(scope="module")
def rec_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "records.npz"))


def test_oracle_matches_reference_outputs(rec_golden):
    images, masks = records.make_records()
    assert np.array_equal(masks, rec_golden["mask_in"])  # same synthetic records as the fixture generator saw
    lut = rec_golden["image_lut"]
    assert np.array_equal(lut, (np.arange(256, dtype=np.float32) / np.float32(255.0)))  # reference value per byte
    got_i, got_m = records.decode_records(images, masks)
    for i in range(images.shape[0]):
        assert str(rec_golden[f"mask_dtype_{i}"]) == "int64" and got_m[i].dtype == np.int64
        assert np.array_equal(got_m[i], rec_golden[f"mask_out_{i}"].astype(np.int64)), f"mask {i}"
        assert np.array_equal(got_i[i], np.transpose(lut[images[i]], (2, 0, 1))), f"image {i}"  # bit-exact
        assert np.array_equal(got_i[i][:, ::37, ::41], rec_golden[f"image_samples_{i}"])
        np.testing.assert_allclose(got_i[i].astype(np.float64).sum((1, 2)), rec_golden[f"image_sum_{i}"], rtol=0, atol=1e-6)
    # branch coverage of the mask rule: cat record -> {0,1}; dog / neither -> {0,2}; cat AND dog -> dog pixels dropped
    assert set(np.unique(got_m[0])) == {0, 1} and set(np.unique(got_m[1])) == {0, 2}
    assert set(np.unique(got_m[2])) == {0, 1} and (got_m[2][masks[2] == 75] == 0).all()
    assert set(np.unique(got_m[3])) == {0, 2} and (got_m[4] == 2).all()


def test_ckpt_prefix_roundtrip():
    import hipseg.ckpt as ck
    from models.UNet import UNet
    net = UNet()
    sd = net.state_dict()
    for compiled in (False, True):
        for ddp in (False, True):
            ref = ck.reference_state_dict(net, compiled=compiled, ddp=ddp)
            pre = ("module." if ddp else "") + ("_orig_mod." if compiled else "")
            assert all(k.startswith(pre) for k in ref) and len(ref) == len(sd) == 124
            back = ck.strip_wrapper_prefixes(ref)
            assert list(back) == list(sd)
            other = UNet()
            assert ck.load_reference_checkpoint(other, ref).missing_keys == []
            assert all(torch.equal(a, b) for a, b in zip(other.state_dict().values(), sd.values()))
    with pytest.raises(KeyError):
        ck.strip_wrapper_prefixes({"module.a": 1, "a": 2})
    # `module.` is a leading prefix only: a submodule whose name ends in "module" is not mangled
    got = ck.strip_wrapper_prefixes({"module._orig_mod.fusion_module.weight": 1, "_orig_mod.module.x.module.y": 2,
                                     "enc._orig_mod.z": 3})
    assert list(got) == ["fusion_module.weight", "x.module.y", "enc.z"]


def test_decode_host_validation():
    import hipseg.data as D
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D.decode_records(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), torch.zeros(1, 4, 4, dtype=torch.uint8))


@pytest.mark.gpu
def test_decode_records_gpu_bit_exact():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hipseg.data as D
    images, masks = records.make_records()
    want_i, want_m = records.decode_records(images, masks)
    got_i, got_m = D.decode_records(torch.from_numpy(images).cuda(), torch.from_numpy(masks).cuda())
    torch.cuda.synchronize()
    assert got_i.dtype == torch.float32 and got_m.dtype == torch.int64
    assert np.array_equal(got_i.cpu().numpy(), want_i) and np.array_equal(got_m.cpu().numpy(), want_m)
    # per-record dict interface of the reference, other geometries (H*W % 4 == 0), every byte value
    i0, m0 = D.decode_record({"image": images[0].tobytes(), "mask": masks[0].tobytes()})
    assert np.array_equal(i0.cpu().numpy(), want_i[0]) and np.array_equal(m0.cpu().numpy(), want_m[0])
    small_i = np.arange(2 * 6 * 10 * 3, dtype=np.uint8).reshape(2, 6, 10, 3)
    small_m = np.array([38, 75, 255, 0, 7], np.uint8)[np.arange(2 * 6 * 10).reshape(2, 6, 10) % 5]
    small_m[1][small_m[1] == 38] = 1
    gi, gm = D.decode_records(torch.from_numpy(small_i).cuda(), torch.from_numpy(small_m).cuda())
    wi = np.transpose(small_i, (0, 3, 1, 2)).astype(np.float32) / np.float32(255.0)
    wm0 = (small_m[0] == 38).astype(np.int64) + (small_m[0] == 255)
    wm1 = 2 * (small_m[1] == 75).astype(np.int64) + 2 * (small_m[1] == 255)
    assert np.array_equal(gi.cpu().numpy(), wi)
    assert np.array_equal(gm[0].cpu().numpy(), wm0) and np.array_equal(gm[1].cpu().numpy(), wm1)
    with pytest.raises(Exception):
        D.decode_records(torch.zeros(1, 3, 3, 3, dtype=torch.uint8).cuda(), torch.zeros(1, 3, 3, dtype=torch.uint8).cuda())
