"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's dataset-record decode.

Reference: customDatasets/datasets.py:92-135 (`CustomImageDataset._deserialize_datapoint` / `_deserialize_numpy`):
a record holds the raw bytes of a 256x256x3 uint8 HWC image and of a 256x256 uint8 trimap-derived mask
(38 = cat, 75 = dog, 255 = uncertain border, anything else = background).
  image -> float32 CHW, value / 255.0
  mask  -> int64:  if the record has ANY cat pixel: (mask == 38) + (mask == 255)        (cat 1, border 1)
                   else:                            2 * (mask == 75) + 2 * (mask == 255) (dog 2, border 2)
Pinned by tests/golden/records.npz (outputs of the reference function itself on the inputs of `make_records`).
"""
import numpy as np

from . import fill

H = W = 256  # the reference hard-codes the record geometry (datasets.py:96,133)


def decode_record(image_bytes, mask_bytes):
    """bytes, bytes -> (float32 (3,H,W), int64 (H,W)); datasets.py:92-131."""
    image = np.frombuffer(image_bytes, dtype=np.uint8).reshape(H, W, 3)
    mask = np.frombuffer(mask_bytes, dtype=np.uint8).reshape(H, W)
    img = np.transpose(image, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    cat = np.where(mask == 38, 1, 0)
    dog = np.where(mask == 75, 2, 0)
    unc = np.where(mask == 255, 1, 0)
    out = cat + unc if cat.sum() > 0 else dog + 2 * unc
    return img, out.astype(np.int64)


def decode_records(images_u8, masks_u8):
    """(n,H,W,3) uint8, (n,H,W) uint8 -> (n,3,H,W) float32, (n,H,W) int64: the batched form the HIP kernel mirrors."""
    imgs, msks = [], []
    for i in range(images_u8.shape[0]):
        a, b = decode_record(images_u8[i].tobytes(), masks_u8[i].tobytes())
        imgs.append(a)
        msks.append(b)
    return np.stack(imgs), np.stack(msks)


def make_records():
    """Deterministic synthetic records covering every branch of the mask rule.  Returns (images (6,H,W,3) uint8,
    masks (6,H,W) uint8): 0 cat + border, 1 dog + border, 2 cat AND dog (cat wins, dog pixels drop to 0),
    3 neither (border only -> 2), 4 all border, 5 arbitrary byte values."""
    n = 6
    images = fill.randint("records.images", (n, H, W, 3), 256).astype(np.uint8)
    masks = np.zeros((n, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]

    def blob(cy, cx, r):
        return (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r

    def ring(cy, cx, r):
        d = (yy - cy) ** 2 + (xx - cx) ** 2
        return (d > r * r) & (d <= (r + 4) ** 2)

    masks[0][ring(100, 120, 60)] = 255
    masks[0][blob(100, 120, 60)] = 38
    masks[1][ring(140, 90, 50)] = 255
    masks[1][blob(140, 90, 50)] = 75
    masks[2][blob(70, 70, 40)] = 38
    masks[2][blob(180, 180, 45)] = 75
    masks[2][ring(180, 180, 45)] = 255
    masks[3][ring(128, 128, 30)] = 255
    masks[4][:] = 255
    masks[5] = fill.randint("records.mask5", (H, W), 256).astype(np.uint8)
    masks[5][masks[5] == 38] = 37  # no cat pixel: exercises the dog branch on arbitrary bytes
    return images, masks
