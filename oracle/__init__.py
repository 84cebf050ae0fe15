"""CPU oracle for the U-Net / ClipUnet training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the thing measured or shipped.  The product path
(``image-segmentation_amd/``) never imports this package and raises when its
HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's
own Python modules (``/root/reference/models``) in the build container and
dumps the fixtures under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those fixtures.  The one exception is the
Dice term of ``HybridLossBinary`` (third-party ``segmentation_models_pytorch
==0.4.0`` arithmetic, package absent, no reference fixture): that term is
"parity unpinned" and is checked against hand-derived known answers only.
"""
