"""CPU oracle for the chunk -> latent hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package (`amphibian_vae_latent_detector_amd/`) imports
this directory.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker / the
reported CPU baseline, never as the thing that is measured or shipped.

Pinning status (see DESIGN.md "Oracle"):

* `rms_normalize`, `crop_or_pad_time`, the encoder-output adapter, the radial
  fit and the radial decision are PINNED against the reference's own functions
  (imported in the build container by `tests/golden/make_golden.py`; the
  captured vectors are committed under `tests/golden/`).
* the Gaussian-MAP detector (`oracle.map_score`: estimate_cov, inv_and_logdet,
  gaussian_logpdf_from_precision, the 08b fit block, priors, the 10b decision) is
  PINNED the same way (`tests/golden/make_golden_map.py` -> `map_fit.npz`,
  `map_decide*.npz`).
* the resampler (`oracle.resample`: resampy 0.4.3 kaiser_best + librosa's
  fix_length, used by librosa.load only when a file's rate differs from `sr`) is
  **parity unpinned**: neither package nor its filter table is available and the
  reference holds no resampled fixture; anchored on signal-level known answers.
* the mel front end (librosa==0.9.2 arithmetic) and the encoder architecture
  (un-vendored `soundscape_vae`) are **parity unpinned**: librosa is not
  installable here and the reference holds no fixtures for it, so
  `oracle.frontend` restates librosa 0.9.2's published algorithm and is
  anchored only by known-answer checks of the Slaney mel scale and an
  independent scipy.signal.stft cross-check.
"""
