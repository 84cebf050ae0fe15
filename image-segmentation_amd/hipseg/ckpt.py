"""Checkpoint exchange with the reference.

The reference saves `model.state_dict()` of a `torch.compile`d (and, under DDP, wrapped) model, so keys carry the
prefixes `_orig_mod.` and `module.` (models/model_wrappers.py:249,1047); its loaders strip `_orig_mod.` before
`load_state_dict` (models/model_wrappers.py:323-332).  Our modules register the same names and shapes, so a
reference checkpoint loads here -- and ours loads there -- once the wrapper prefixes are normalised."""
from collections import OrderedDict

import torch

PREFIXES = ("_orig_mod.", "module.")


def _strip_key(k):
    """`module.` (DDP) is only ever a LEADING prefix, possibly interleaved with `_orig_mod.` (compile inside/outside
    DDP); `_orig_mod.` is removed as a dot-delimited path component anywhere (model_wrappers.py:326-329 uses
    `k.replace("_orig_mod.", "")`).  A submodule whose name merely ENDS in "module" (`fusion_module.weight`) is kept."""
    changed = True
    while changed:
        changed = False
        for p in PREFIXES:
            if k.startswith(p):
                k, changed = k[len(p):], True
    return ".".join(c for c in k.split(".") if c != "_orig_mod")


def strip_wrapper_prefixes(state_dict):
    """Remove the `_orig_mod.` / `module.` wrapper prefixes of a reference-format checkpoint."""
    out = OrderedDict()
    for k, v in state_dict.items():
        nk = _strip_key(k)
        if nk in out:
            raise KeyError(f"hipseg.ckpt: keys collide after prefix removal: {k!r} -> {nk!r}")
        out[nk] = v
    return out


def load_reference_checkpoint(model, path_or_state, strict=True):
    """Load a reference-format checkpoint (path or state_dict) into one of our drop-in models.  Files are opened
    with `weights_only=True` only (nothing from the file is executed)."""
    sd = path_or_state
    if not isinstance(sd, dict):
        sd = torch.load(path_or_state, map_location="cpu", weights_only=True)
    return model.load_state_dict(strip_wrapper_prefixes(sd), strict=strict)


def reference_state_dict(model, compiled=False, ddp=False):
    """Our state_dict under the key names the reference's wrappers would have produced."""
    prefix = ("module." if ddp else "") + ("_orig_mod." if compiled else "")
    return OrderedDict((prefix + k, v) for k, v in model.state_dict().items())
