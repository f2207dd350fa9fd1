"""2D flow matching (SFM) with mid-level attention (CPU plumbing, like BASELINE config C1).  Same command line as the reference script:
    python trainSFM_c_uc_from_field_name.py <field_in> <field_out>"""
from vdm4cdm_amd.entry import train_sfm_c_uc_2d

if __name__ == "__main__":
    train_sfm_c_uc_2d()
