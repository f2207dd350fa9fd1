"""Summary statistics of generated cubes.  Same command line as the reference script of this name:
    python calc_SS.py <model_name>      (reads <VDM4CDM_GEN_DIR or ./data/ICML_v2>/<model_name>/{CV_1_128,CV_12_12,1P_24,1P_128}/)"""
from vdm4cdm_amd.calc_ss import main

if __name__ == "__main__":
    main()
