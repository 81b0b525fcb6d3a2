#!/bin/bash
# HBM-side bytes of full-frame forward-only rendering (tools/inference_rate.py: 409,600 rays x 64 samples, lean_inference, the frame
# rendered twice): FETCH_SIZE / WRITE_SIZE in passes of their own, summed over all kernels of the process; counter units from the
# calibration of tools/pmc_traffic.sh (<pmc dir>/pmc_hbm_traffic.json).   usage: pmc_inference.sh <out dir under gpurun_out> <pmc dir> [mode]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# <pmc dir>: a directory under gpurun_out/ of the same call, or a committed one (profiles/rNN: the units are constants of the counter)
O=gpurun_out/$1; if [ -f $2/pmc_hbm_traffic.json ]; then CAL=$2/pmc_hbm_traffic.json; else CAL=gpurun_out/$2/pmc_hbm_traffic.json; fi; MODE=${3:-f16x2}; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o t -- python3 tools/inference_rate.py 40960 lean $MODE > $O/$c.log 2>&1 || { tail -5 $O/$c.log; exit 1; }
done
python3 - <<PY
import csv, collections, json
cal = json.load(open("$CAL"))["calibration"]
unit = {"FETCH_SIZE": cal["fetch_bytes_per_unit"], "WRITE_SIZE": cal["write_bytes_per_unit"]}
tot, per = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open("$O/%s/t_counter_collection.csv" % c)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("snerf::bsp::", "").replace("snerf::", "")
        agg[k] += float(r["Counter_Value"]) * unit[c]; n[k] += 1
    tot[c] = sum(agg.values()); per[c] = {k: {"launches": n[k], "bytes": v} for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:8]}
rays, frames = 640 * 640, 2
out = {"mode": "$MODE", "rays_per_frame": rays, "samples": 64, "frames_rendered": frames, "render_chunk_size": 40960,
       "read_bytes": tot["FETCH_SIZE"], "written_bytes": tot["WRITE_SIZE"],
       "lean": {"hbm_bytes_per_frame": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / frames, "hbm_bytes_per_ray": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / frames / rays},
       "top_kernels": per,
       "how": "tools/pmc_inference.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (passes of their own) over lean_inference of a 409,600-ray frame x 64 samples rendered "
              "twice; all kernels of the process; units calibrated on a known copy (pmc_hbm_traffic.json of the same session)"}
json.dump(out, open("$O/pmc_inference.json", "w"), indent=1)
print(json.dumps(out["lean"]), "read %.1f GB written %.1f GB" % (tot["FETCH_SIZE"] / 1e9, tot["WRITE_SIZE"] / 1e9))
PY
