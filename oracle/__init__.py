"""CPU oracle for the SYN3R hot path — TEST INFRASTRUCTURE ONLY.

Plain numpy / torch-CPU restatements of the reference algorithms (each function cites the
reference file:line it follows).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg may import this package; the product package
`syn3r_amd` never does and has no CPU fallback.

Pinning status:
  geometry (warp_oracle)      pinned by tests/golden/warp_*.npz   (generated from the reference, oracle/gen_golden.py)
  scheduler (scheduler_oracle) pinned by tests/golden/sched_*.npz  (generated from the reference)
  UNet (unet_oracle)          pinned by tests/golden/unet_small.npz (outputs of the reference module itself)
  VAE                         no restatement: tests compare with tests/golden/vae_small.npz (reference module outputs)
  pipelines / orchestrator    no restatement: tests/golden/pipeline_mock.npz, pipeline_unet.npz, pipeline_unet_vae.npz (the
                              reference pipeline classes' own __call__: with mock modules / with the reference UNet /
                              with the reference UNet and VAE), orchestrator.npz
  rasteriser (raster_oracle)  PARITY UNPINNED — the reference's CUDA rasteriser source is an
                              un-vendored submodule; restates the published 3DGS algorithm
"""
