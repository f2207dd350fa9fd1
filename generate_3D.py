"""Sampling entry point.  Same command line as the reference script of this name:
    python generate_3D.py <model_name> <save_path> <runtype>      (runtype: CV_12_12 | CV_1_128)"""
from vdm4cdm_amd.entry import generate_3d

if __name__ == "__main__":
    generate_3d()
