"""CPU oracle for the vdm4cdm variational-diffusion denoising hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / the timed CPU baseline.  Nothing under ``vdm4cdm_amd/`` imports it.

What it restates
----------------
* ``unet_oracle``  - the 3D/2D conditional UNet score network (``mltools.networks.networks.CUNet``,
  called from /root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:116-127 and
  /root/reference/src/utils.py:451-462) as plain fp32 ``torch.nn.functional`` ops on the CPU
  (conv3d / group_norm / silu / linear / gelu / interpolate) in NCDHW.
* ``vdm_oracle``   - the VDM noise schedule, ELBO loss and ancestral sampler
  (``mltools.models.vdm_model.VDM``; the only in-tree fragments are the notebook traceback
  lines ``vdm_model.py:318-324,370-378,429-442`` quoted in SURVEY.md section 3.2, and the call sites
  /root/reference/src/utils.py:286-299).
* ``pk_oracle``    - the isotropic power spectrum / cross-correlation estimators
  /root/reference/src/utils.py:16-128 (``power``, ``pk``, ``get_ccs``) in numpy.

Pinning status
--------------
* ``pk_oracle``: PINNED.  Checked against golden vectors produced by importing the reference's
  own ``src/utils.py`` in the build container (``tests/golden/make_pk_golden.py`` ->
  ``tests/golden/pk_golden.npz``).
* ``unet_oracle`` / ``vdm_oracle``: **PARITY UNPINNED.**  The network and VDM arithmetic live in
  the third-party package ``mltools`` (github cfpark00/MLtools, used as an editable ``~/MLtools``
  checkout, no version pinned anywhere in the reference), which is not vendored under
  /root/reference, not installed, and not fetchable (no network).  The reference has no tests,
  no checkpoints and no golden tensors for this path.  The oracle therefore restates the
  published VDM formulation (Kingma et al. 2021) under the spec decisions D1-D12 of SURVEY.md
  section 8 and is anchored only on the reference's call signatures and the traceback fragments.
"""
