"""Device-side dataset-record decode (reference: customDatasets/datasets.py:92-135).

`decode_records(images_u8, masks_u8)` is the batched, on-GPU form of `CustomImageDataset._deserialize_datapoint`:
raw HWC uint8 images and uint8 masks in, `(float32 (n,3,H,W), int64 (n,H,W))` out -- bit-identical to the reference's
per-record numpy/torch arithmetic.  `decode_record(datapoint)` keeps the reference's per-record dict interface.
No CPU fallback: CPU tensors raise."""
import numpy as np
import torch

from . import _lib as L


def decode_records(images_u8, masks_u8):
    if not (images_u8.is_cuda and masks_u8.is_cuda):
        raise RuntimeError("hipseg: decode_records needs CUDA/HIP tensors (no CPU fallback)")
    if images_u8.dtype != torch.uint8 or masks_u8.dtype != torch.uint8:
        raise TypeError("hipseg: decode_records takes uint8 tensors")
    if images_u8.dim() != 4 or images_u8.shape[3] != 3 or masks_u8.shape != images_u8.shape[:3]:
        raise ValueError(f"hipseg: expected images (n,H,W,3) and masks (n,H,W); got {tuple(images_u8.shape)}, "
                         f"{tuple(masks_u8.shape)}")
    n, H, W = masks_u8.shape
    images_u8, masks_u8 = images_u8.contiguous(), masks_u8.contiguous()
    out_i = torch.empty((n, 3, H, W), dtype=torch.float32, device=images_u8.device)
    out_m = torch.empty((n, H, W), dtype=torch.int64, device=images_u8.device)
    flags = torch.empty(n, dtype=torch.int32, device=images_u8.device)
    L.decode_records(images_u8.data_ptr(), masks_u8.data_ptr(), out_i.data_ptr(), out_m.data_ptr(), flags.data_ptr(), n, H, W,
                     torch.cuda.current_stream().cuda_stream)
    return out_i, out_m


def decode_record(datapoint, device="cuda", shape=(256, 256)):
    """{'image': bytes, 'mask': bytes} -> (image (3,H,W) float32, mask (H,W) int64), as datasets.py:92-131."""
    H, W = shape
    img = torch.from_numpy(np.frombuffer(datapoint["image"], dtype=np.uint8).reshape(1, H, W, 3).copy()).to(device)
    msk = torch.from_numpy(np.frombuffer(datapoint["mask"], dtype=np.uint8).reshape(1, H, W).copy()).to(device)
    i, m = decode_records(img, msk)
    return i[0], m[0]


# ---------------------------------------------------------------------------------------------------------------
# The reference's `.pt` dataset cache (customDatasets/datasets.py:64-83): `torch.save` of a Python LIST with one
# `(image float32 (3,H,W), mask int64 (H,W))` tuple per record -- the outputs of `_deserialize_datapoint` -- stored as
# `<dataset_loc>/<split>_dataset.pt` and read back with `torch.load(..., weights_only=True)`.
def cache_path(dataset_loc, split):
    """file name the reference uses (datasets.py:66)."""
    import os

    return os.path.join(dataset_loc, f"{split}_dataset.pt")


def build_dataset_cache(images_u8, masks_u8, chunk=1024):
    """decode raw records on the GPU (chunks of `chunk` records) into the reference's cache structure: a list of
    `(image, mask)` CPU tensor tuples, bit-identical to the reference's per-record loop (datasets.py:74-76)."""
    out = []
    for i in range(0, images_u8.shape[0], chunk):
        im, mk = decode_records(images_u8[i:i + chunk].cuda(non_blocking=True), masks_u8[i:i + chunk].cuda(non_blocking=True))
        im, mk = im.cpu(), mk.cpu()
        out.extend((im[j].clone(), mk[j].clone()) for j in range(im.shape[0]))
    return out


def save_dataset_cache(path, cache):
    """write a cache list in the reference's on-disk format (plain `torch.save` of the list of tuples)."""
    for img, msk in cache:
        if img.dtype != torch.float32 or img.dim() != 3 or msk.dtype != torch.int64 or msk.shape != img.shape[1:]:
            raise TypeError("dataset cache entries are (float32 (3,H,W), int64 (H,W)) tuples")
    torch.save([(img.cpu().contiguous(), msk.cpu().contiguous()) for img, msk in cache], path)


def load_dataset_cache(path, device=None):
    """read a reference-format cache with `weights_only=True` (nothing from the file is executed).  Returns the list of
    tuples as the reference holds it, or -- with `device` -- two stacked tensors `(n,3,H,W) float32`, `(n,H,W) int64`
    resident on that device (the whole training set of the reference fits in HBM many times over)."""
    cache = torch.load(path, weights_only=True)
    if not isinstance(cache, list) or not all(isinstance(e, (tuple, list)) and len(e) == 2 for e in cache):
        raise ValueError(f"{path}: not a dataset cache (expected a list of (image, mask) tuples)")
    if device is None:
        return [tuple(e) for e in cache]
    return (torch.stack([e[0] for e in cache]).to(device), torch.stack([e[1] for e in cache]).to(device))
