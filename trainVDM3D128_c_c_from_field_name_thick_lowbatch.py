"""128^3 conditional VDM training.  Same command line as the reference script of this name:
    python trainVDM3D128_c_c_from_field_name_thick_lowbatch.py <field_in> <field_out> <cropsize>"""
from vdm4cdm_amd.entry import train_vdm3d

if __name__ == "__main__":
    train_vdm3d("128")
