"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into profiles/<round>/<name>_pmc.json and update
profiles/traffic.json (HBM bytes per launch of the fused MLP kernels, FETCH_SIZE doubled per the gfx950
correction in MI355X_MICROARCH.md section HBM).  usage: pmc_summary.py <out_json> <rays> <samples> <csv>..."""
import collections, csv, json, os, sys

out_json, rays, samples = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
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
for k, cs in agg.items():
    if "mlp_fused_kernel" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        w = k.split("mlp_fused_kernel<")[1].split(",")[0]
        prec = "bf16" if "BF16" in k else "f32"
        f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024      # counter unit: KB
        wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
        traffic[f"mlp_fused_W{w}_{prec}"] = {"hbm_bytes_per_launch": 2 * f + wr, "fetch_bytes_raw": f, "write_bytes": wr,
                                              "rays": rays, "samples": samples,
                                              "note": "FETCH_SIZE x2 (gfx950 wide-stream correction) + WRITE_SIZE, separate --pmc passes"}
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps(traffic, indent=1))
