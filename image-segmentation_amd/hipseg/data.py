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
