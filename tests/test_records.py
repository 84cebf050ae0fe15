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


def test_pt_dataset_cache_format(tmp_path, rec_golden):
    """`.pt` cache of customDatasets/datasets.py:64-83: a list of (image float32 CHW, mask int64 HW) tuples written by
    torch.save and read with weights_only=True -- files written the reference's way load here, files written here
    load the reference's way, entry for entry (entries = the pinned oracle's decode of the synthetic records)."""
    import hipseg.data as D

    images, masks = records.make_records()
    # the reference's loop: dataset_cache.append(self._deserialize_datapoint(datapoint)); torch.save(dataset_cache, f)
    ref_cache = []
    for i in range(images.shape[0]):
        img, msk = records.decode_record(images[i].tobytes(), masks[i].tobytes())
        assert np.array_equal(msk, rec_golden[f"mask_out_{i}"].astype(np.int64))  # == the reference's own output
        ref_cache.append((torch.from_numpy(img), torch.tensor(msk)))
    ref_file = tmp_path / "train_dataset.pt"
    assert D.cache_path(str(tmp_path), "train") == str(ref_file)
    torch.save(ref_cache, ref_file)
    got = D.load_dataset_cache(ref_file)
    assert len(got) == len(ref_cache)
    for (gi, gm), (ri, rm) in zip(got, ref_cache):
        assert gi.dtype == torch.float32 and gm.dtype == torch.int64 and torch.equal(gi, ri) and torch.equal(gm, rm)
    si, sm = D.load_dataset_cache(ref_file, device="cpu")
    assert si.shape == (6, 3, 256, 256) and sm.shape == (6, 256, 256) and torch.equal(si[2], ref_cache[2][0])
    # and the other way round: our writer -> the reference's reader (torch.load(cache_file, weights_only=True))
    ours = tmp_path / "validation_dataset.pt"
    D.save_dataset_cache(ours, got)
    back = torch.load(ours, weights_only=True)
    assert isinstance(back, list) and all(isinstance(e, tuple) and len(e) == 2 for e in back)
    for (bi, bm), (ri, rm) in zip(back, ref_cache):
        assert torch.equal(bi, ri) and torch.equal(bm, rm)
    with pytest.raises(TypeError):
        D.save_dataset_cache(tmp_path / "bad.pt", [(torch.zeros(3, 4, 4), torch.zeros(4, 4))])
    torch.save({"a": torch.zeros(1)}, tmp_path / "notcache.pt")
    with pytest.raises(ValueError):
        D.load_dataset_cache(tmp_path / "notcache.pt")


@pytest.mark.gpu
def test_build_dataset_cache_on_gpu():
    """records -> cache entries through the HIP decode kernel == the oracle's per-record decode, bit for bit."""
    import hipseg.data as D

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    images, masks = records.make_records()
    cache = D.build_dataset_cache(torch.from_numpy(images), torch.from_numpy(masks), chunk=4)
    assert len(cache) == images.shape[0]
    for i, (img, msk) in enumerate(cache):
        ri, rm = records.decode_record(images[i].tobytes(), masks[i].tobytes())
        assert img.dtype == torch.float32 and msk.dtype == torch.int64
        assert np.array_equal(img.numpy(), ri) and np.array_equal(msk.numpy(), rm)
