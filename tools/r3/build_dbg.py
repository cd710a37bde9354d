"""Side builds of libdisco_hip for the round-3 bound experiments (never loaded by the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from disco_amd.build import build_lib
for name, flags in (("libdisco_dbg_allhit.so", ["-DDISCO_DBG_COLMASK=4095"]),
                    ("libdisco_dbg_normw.so", ["-DDISCO_DBG_NORMW=1"]),
                    ("libdisco_dbg_nogather.so", ["-DDISCO_DBG_NOGATHER=1"]),
                    ("libdisco_dbg_allhit_normw.so", ["-DDISCO_DBG_COLMASK=4095", "-DDISCO_DBG_NORMW=1"])):
    print(build_lib(force=True, extra_flags=flags, name=name))
