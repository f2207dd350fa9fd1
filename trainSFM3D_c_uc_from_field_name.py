"""256^3 flow matching (SFM) without parameter conditioning, chs 12/36/64/128.  Same command line as the reference script of this name:
    python trainSFM3D_c_uc_from_field_name.py <field_in> <field_out> <cropsize>"""
from vdm4cdm_amd.entry import train_sfm3d

if __name__ == "__main__":
    train_sfm3d("256_c_uc")
