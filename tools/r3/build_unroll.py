"""Side builds for the in-flight-depth / occupancy sensitivity of the windowed kernel (round 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from disco_amd.build import build_lib
for u, w in ((4, 8), (4, 6), (12, 5), (16, 4), (16, 3), (8, 5), (8, 4)):
    print(build_lib(force=True, extra_flags=[f"-DDISCO_U1={u}", f"-DDISCO_WIN_WAVES={w}"], name=f"libdisco_dbg_u{u}_w{w}.so"))
