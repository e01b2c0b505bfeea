"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into profiles/<round>/<name>_pmc.json and update
profiles/traffic.json (HBM bytes per launch of the fused MLP kernels, FETCH_SIZE doubled per the gfx950
correction in MI355X_MICROARCH.md section HBM).  usage: pmc_summary.py <out_json> <rays> <samples> <csv>..."""
import collections, csv, json, os, sys

out_json, rays, samples = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(os.environ.get("PMC_STEPS", "0"))      # render_rays calls in each profiled run (timed + warm-up): for the per-step table
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[4:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k[:120]: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
           for k, cs in agg.items()}
json.dump(summary, open(out_json, "w"), indent=1)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tpath = os.path.join(root, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
step_total = 0.0
per_step = {}
for k, cs in agg.items():
    if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024      # counter unit: KB
    wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
    if steps:
        b = (2 * f + wr) * len(cs["FETCH_SIZE"]) / steps
        per_step[k[:100]] = {"launches_per_step": len(cs["FETCH_SIZE"]) / steps, "hbm_bytes_per_step": b}
        step_total += b
    if "mlp_fused_kernel" in k:
        args = [t.strip() for t in k.split("mlp_fused_kernel<")[1].split(">")[0].split(",")]
        w = args[0]
        prec = "bf16" if "BF16" in k else ("f16" if "PrecF16" in k else "f32")
        tag = f"mlp_warp_W{w}_{prec}" if len(args) >= 7 and args[6] == "true" else f"mlp_fused_W{w}_{prec}"
        traffic[tag] = {"hbm_bytes_per_launch": 2 * f + wr, "fetch_bytes_raw": f, "write_bytes": wr,
                        "rays": rays, "samples": samples,
                        "note": "FETCH_SIZE x2 (gfx950 wide-stream correction) + WRITE_SIZE, separate --pmc passes"}
if steps:
    mode = os.environ.get("PMC_MODE", "fp16")        # the render mode of the profiled run (bench.py --precision)
    traffic["render_step_total" + ("" if mode == "fp16" else "_" + mode)] = {
                                    "hbm_bytes_per_step": step_total, "rays": rays, "samples": samples,
                                    "note": f"sum over every kernel of one render_rays call ({mode} mode), same correction",
                                    "kernels": dict(sorted(per_step.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:12])}
sys.path.insert(0, root)
from moda_amd.build import source_hash      # noqa: E402
traffic["_stamp"] = {"kernel_source_sha16": source_hash(),
                     "note": "moda_amd.build.source_hash() of the tree these counters were collected on; bench.py prints "
                             "roofline.traffic only when it equals the hash of the tree it runs from"}
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps({k: v for k, v in traffic.items() if k not in ("render_step_total", "_stamp")}, indent=1))
if steps:
    print("render step total HBM bytes:", step_total)
