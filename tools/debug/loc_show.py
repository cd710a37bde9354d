import json
for l in open("gpurun_out/loc_bench.log"):
    if l.startswith("=="):
        print(l.strip()); continue
    try:
        d = json.loads(l); print("   ms/step", round(d["ms_per_step"], 3), d["config"].get("locality_pass"))
    except Exception:
        print("   ??", l[:300])
