"""CPU oracle for the ae_combined hot path -- TEST INFRASTRUCTURE ONLY.

This package is a PyTorch-CPU fp32 restatement of the reference's arithmetic for the
hot path (auto-encoder fwd/bwd, latent lerp, MSE + LPIPS loss, the train step and the
slice-synthesis inference loop).  It exists to CHECK the HIP product path; it is never
shipped, never imported by ``superresolution_aniso_mri_amd`` and never the thing that is
measured as the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Pinning: the reference has no tests/golden vectors of its own, so the
oracle is pinned against outputs of the reference's own modules run in the build
container (``oracle/make_golden.py`` imports ``/root/reference`` with stubs for the
unused, missing third-party imports and writes ``tests/golden/*.npz``); the not-gpu test
suite replays those vectors through this oracle.  Parts of the path whose arithmetic
lives in an absent third-party dependency (the pretrained torchvision VGG16 backbone
weights) are pinned only with a deterministic synthetic backbone -- "parity unpinned"
for real ImageNet weights, see DESIGN.md.
"""
