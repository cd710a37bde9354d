import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from disco_amd.build import build_lib
for r, w in ((4, 6), (4, 5), (8, 5)):
    print(build_lib(force=True, extra_flags=[f"-DDISCO_WIN_MAXR={r}", f"-DDISCO_WIN_WAVES={w}"], name=f"libdisco_dbg_r{r}_w{w}.so"))
