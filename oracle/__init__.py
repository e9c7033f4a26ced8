"""CPU oracle for the DiffMusic hot path -- TEST INFRASTRUCTURE ONLY.

A pure-PyTorch fp32 restatement of the reference's per-step guided-diffusion loop
(U-Net -> VAE decode -> HiFi-GAN -> measurement operator -> mel -> L2 -> autograd ->
DDIM/DPS/MPGD/DSG/DiffMusic update).  Every function cites the reference file:line (paths
into the upstream repo, jwliao1209/DiffMusic @ 2025-06-13) or the public third-party
semantics it restates.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package.  The product (`diffmusic_amd/`) never does: it fails loudly when the HIP
library is missing.

Parity pinning status
---------------------
* reference-owned glue (scheduler step bodies, masks, rect-STFT operator, reverb operator,
  randn_tensor): pinned by golden vectors generated from the reference's own sources under a
  name shim (`tests/golden/gen_golden.py`, fixtures in `tests/golden/*.npz`).
* third-party arithmetic (diffusers 0.31.0 U-Net/VAE/DDIM parent, torchaudio mel/resample):
  packages are absent from this image and the reference has no tests for them, so those parts
  are **parity unpinned** against the reference; they are cross-checked against independent
  implementations available here (`torch.stft`, `transformers.SpeechT5HifiGan`,
  `transformers.audio_utils.mel_filter_bank`).
"""
