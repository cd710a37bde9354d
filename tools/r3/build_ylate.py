"""Side build: windowed kernel with the running sum read at the end of a visit (experiment)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from disco_amd.build import build_lib
print(build_lib(force=True, extra_flags=["-DDISCO_WIN_YLATE=1"], name="libdisco_dbg_ylate.so"))
