"""vdm4cdm_amd - MI355X-native (gfx950) implementation of the vdm4cdm variational-diffusion denoising path.

Public modules mirror what the reference scripts import from ``mltools``:
``networks`` (CUNet), ``vdm_model`` (VDM, LightVDM).  Compute kernels: libvdm4cdm_hip.so (C-ABI in include/).
"""
__version__ = "0.1.0"
