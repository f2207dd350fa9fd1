"""2D unconditional VDM (BASELINE config C1: CPU PyTorch plumbing).  Same command line as the reference script of this name:
    python train_uc_uc_from_field_name.py <field_name>"""
from vdm4cdm_amd.entry import train_uc_uc

if __name__ == "__main__":
    train_uc_uc()
