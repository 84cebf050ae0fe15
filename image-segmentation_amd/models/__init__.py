"""Drop-in replacements for the reference's `models` package on the training hot path:
models.processing_blocks, models.UNet, models.CLIP_models, models.prompt_segmentation, models.losses -- same class names,
constructor / forward signatures and state_dict layout, backed by hand-written HIP kernels."""
