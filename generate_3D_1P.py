"""Sampling entry point for the CAMELS 1P set.  Same command line as the reference script of this name:
    python generate_3D_1P.py <model_name> <save_path> <runtype>      (runtype: 1P_24 | 1P_128)"""
from vdm4cdm_amd.entry import generate_3d_1p

if __name__ == "__main__":
    generate_3d_1p()
