"""2D parameter-conditioned VDM (CPU PyTorch plumbing).  Same command line as the reference script of this name:
    python train_uc_c_from_field_name.py <field_name>"""
from vdm4cdm_amd.entry import train_uc_c

if __name__ == "__main__":
    train_uc_c()
