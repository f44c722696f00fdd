"""CPU oracle for the Latent2im walk-training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch-on-CPU restatement of
the reference's algorithm for the path named in BASELINE.json (sample z -> W+ ->
linear walk -> StyleGAN2 synthesis x2 -> ResNet-50 / VGG-19 / discriminator
losses -> gradient into the walk -> Adam).  Each function cites the reference
``file:line`` it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``latent2im_amd`` never does (tests/test_host_logic.py::test_product_never_imports_the_oracle_or_the_reference
enforces that).

Pinning status
--------------
* First-party arithmetic (generator, discriminator, fused_bias_act, upfirdn2d,
  walk, losses, optimizer step, PGGAN-256 generator): PINNED against the
  reference's own Python imported in the build container through import shims
  (tests/golden/make_golden.py; outputs committed under tests/golden/*.npz and
  re-checked by tests/test_oracle_golden.py, tests/test_oracle_pggan.py and
  tests/test_next_rows_cpu.py).
* Third-party arithmetic (torchvision ResNet-50 v0.5.0 and VGG-19 ``features``):
  source is neither vendored in the reference nor installed in this image, and
  the reference holds no test or golden vector at that boundary => PARITY
  UNPINNED for those two networks; they are restated from the published
  architecture on top of torch's own conv / batch-norm
  primitives, and the product is checked against this restatement.
"""
