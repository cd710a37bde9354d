"""Side builds for the gather cache-policy experiment (never loaded by the product)."""
import sys, os
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from disco_amd.build import build_lib
variants = [("libdisco_dbg_aux0.so", ["-DDISCO_GATHER_AUX=0"]), ("libdisco_dbg_aux1.so", ["-DDISCO_GATHER_AUX=1"]),
            ("libdisco_dbg_aux2.so", ["-DDISCO_GATHER_AUX=2"]), ("libdisco_dbg_aux16.so", ["-DDISCO_GATHER_AUX=16"]),
            ("libdisco_dbg_aux17.so", ["-DDISCO_GATHER_AUX=17"])]
with ThreadPoolExecutor(max_workers=5) as ex:
    for r in ex.map(lambda v: build_lib(force=True, extra_flags=v[1], name=v[0]), variants):
        print(r)
