"""256^3 flow matching (SFM) training, chs 16..128.  Same command line as the reference script of this name:
    python trainSFM3D_c_c_from_field_name_thick_lowbatch.py <field_in> <field_out> <cropsize>"""
from vdm4cdm_amd.entry import train_sfm3d

if __name__ == "__main__":
    train_sfm3d("256")
