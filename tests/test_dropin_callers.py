"""The drop-in claim, checked against the reference's OWN callers: with this repo's `models` package first on
sys.path, the reference's `models/model_wrappers.py` and `models/helperFunctions.py` are loaded IN PLACE from
/root/reference (nothing is copied into the repo) as `models.model_wrappers` / `models.helperFunctions`, the way a
maintainer would keep them next to the drop-in modules.  Everything the wrappers import by name must resolve
(model_wrappers.py:1-14), their default arguments must bind to the drop-in classes, and the constructor-time
introspection (`save_training_info`, helperFunctions.py:10-125) must work on the drop-in model.

CPU only; skipped where /root/reference does not exist (the GPU box).  Third-party packages the reference imports
at module level and this image lacks (matplotlib, torchvision) are stubbed exactly as SURVEY.md section 8c prescribes;
they are not touched by the code exercised here."""
import importlib.util
import json
import os
import sys
from unittest.mock import MagicMock

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")


def _load_in_place(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ref_callers():
    import transformers, datasets  # noqa: F401,E401  (real packages first)

    stubbed = []
    for n in ("matplotlib", "matplotlib.pyplot", "matplotlib.patches", "torchvision", "torchvision.transforms",
              "torchvision.transforms.v2", "torchvision.models"):
        if n not in sys.modules:
            try:
                importlib.import_module(n)
            except ImportError:
                sys.modules[n] = MagicMock()
                stubbed.append(n)
    import models  # the drop-in package (tests/conftest.py puts image-segmentation_amd first on sys.path)

    assert os.path.realpath(models.__file__).startswith(os.path.realpath(os.path.join(os.path.dirname(__file__), "..")))
    sys.path.append(REF)  # AFTER the drop-in: `customDatasets` / `scripts` resolve to the reference, `models` to ours
    sys.dont_write_bytecode = True
    try:
        hf = _load_in_place("models.helperFunctions", os.path.join(REF, "models", "helperFunctions.py"))
        mw = _load_in_place("models.model_wrappers", os.path.join(REF, "models", "model_wrappers.py"))
        yield mw, hf
    finally:
        sys.path.remove(REF)
        for n in stubbed + ["models.helperFunctions", "models.model_wrappers", "customDatasets", "customDatasets.datasets",
                            "scripts", "scripts.dataset_downloader"]:
            sys.modules.pop(n, None)


def test_reference_wrappers_import_against_the_dropin_package(ref_callers):
    mw, _ = ref_callers
    import models.losses as ls
    import models.processing_blocks as pb
    import models.UNet as un

    # names bound by `from models... import` in model_wrappers.py:1-10 are OUR classes
    assert mw.UNet is un.UNet
    assert mw.HybridLoss is ls.HybridLoss and mw.IoU is ls.IoU and mw.PixelAccuracy is ls.PixelAccuracy
    assert mw.Dice is ls.Dice
    for n in ("DataAugmentor", "DataAugmentorPrompt", "GaussianPixelNoise", "RepeatedBlur", "ContrastChange",
              "BrightnessChange", "Occlusion", "SaltAndPepper", "ConvBlock", "ClipFeatureExtractor",
              "ResNet34FeatureExtractor", "CrossAttentionFusion"):
        assert getattr(mw, n) is getattr(pb, n), n  # star import, model_wrappers.py:2
    # default constructor arguments of the wrappers bind to the drop-in classes (model_wrappers.py:32-45)
    import inspect

    sig = inspect.signature(mw.TrainingWrapper.__init__).parameters
    assert sig["model_class"].default is un.UNet
    assert sig["criterion_class"].default is ls.HybridLoss
    assert sig["data_augmentor_class"].default is pb.DataAugmentor
    assert inspect.signature(mw.DistributedTrainingWrapper.__init__).parameters["criterion_class"].default is ls.HybridLoss
    for cls in (mw.TrainingWrapper, mw.TestWrapper, mw.DistributedTrainingWrapper):
        assert inspect.isclass(cls)


def test_distributed_wrapper_constructs_on_the_dropin_model(ref_callers, tmp_path):
    """DistributedTrainingWrapper.__init__ (model_wrappers.py:827-900): `model.module.__class__.__name__`, optimizer
    over `model.parameters()`, criterion, and `save_training_info` over `named_modules()` -- with the drop-in UNet
    behind a DDP-shaped wrapper, the drop-in DataAugmentor and synthetic data.  No forward pass (no GPU here)."""
    mw, hf = ref_callers
    from torch.utils.data import DataLoader, TensorDataset

    from models.processing_blocks import DataAugmentor
    from models.UNet import UNet

    class DDPShaped(torch.nn.Module):  # what `DDP(model)` / HipDDP expose: `.module` + parameter passthrough
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, x):
            return self.module(x)

    ds = TensorDataset(torch.rand(4, 3, 16, 16), torch.zeros(4, 16, 16, dtype=torch.long))
    dl = DataLoader(ds, batch_size=2)
    save = str(tmp_path / "run") + "/"
    os.makedirs(save)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        w = mw.DistributedTrainingWrapper(0, DDPShaped(UNet()), dl, dl, DataAugmentor(4), batch_size=2,
                                          save_location=str(tmp_path / "saved" / "UNet"))
    finally:
        os.chdir(cwd)
    assert isinstance(w.criterion, mw.HybridLoss) and isinstance(w.optimizer, torch.optim.Adam)
    info = json.load(open(os.path.join(w.save_location, "model_settings.json")))
    st = info["model_structure"]
    assert st["module.enc1.block.0.conv.0"] == {"type": "Conv2d", "in_channels": 32, "out_channels": 64,
                                                 "kernel_size": [3, 3], "padding": [1, 1]}
    assert st["module.dec1.up"]["type"] == "ConvTranspose2d" and st["module.dec1.up"]["kernel_size"] == [2, 2]
    assert info["loss_function"] == "HybridLoss" and info["extra_params"]["num_params"] == 7_755_907
    # the same dump straight through helperFunctions on ClipUnet (MultiheadAttention attributes must stay JSON-able)
    from models.CLIP_models import ClipUnet

    m = ClipUnet(clip_feature_extractor=torch.nn.Identity())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    hf.save_training_info(m, opt, mw.HybridLoss(), dl, dl, save)
    st = json.load(open(save + "model_settings.json"))["model_structure"]
    assert st["cross_attention_fusion.cross_attn"]["embed_dim"] == 512
    assert st["cross_attention_fusion.cross_attn"]["num_heads"] == 1


def test_train_distributed_script_names_resolve(ref_callers):
    """scripts/train_distributed.py:6-9 imports, resolved against the drop-in package; HipDDP accepts the script's
    `DDP(model, device_ids=[rank])` call shape (scripts/train_distributed.py:35)."""
    import inspect

    from hipseg.ddp import HipDDP
    from models.model_wrappers import DistributedTrainingWrapper  # noqa: F401  (the in-place reference module)
    from models.processing_blocks import DataAugmentor
    from models.UNet import UNet  # noqa: F401

    assert DataAugmentor(4).augmentations_per_datapoint == 4
    params = list(inspect.signature(HipDDP.__init__).parameters)
    assert params[1] == "module" and params[2] == "device_ids"
