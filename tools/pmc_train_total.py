"""Sum FETCH_SIZE / WRITE_SIZE over every dispatch of a profiled eager training run -> HBM bytes per step
(2 x FETCH + WRITE: MI355X_MICROARCH.md's gfx950 correction; counter unit KB), the ten largest kernels, and
profiles/traffic.json["train_step_<precision>"] stamped with the kernel-source hash.
usage: pmc_train_total.py <precision>[:<key suffix>:<what>] <steps in the run> <csv>..."""
import collections, csv, json, os, sys

prec, steps = sys.argv[1], int(sys.argv[2])
suffix, what = "", "cfg4 training step (2048 rays x 128 samples)"
if ":" in prec:                      # e.g. bf16:_cfg4_8192x256:cfg4 training step (8192 rays x 256 samples)
    prec, suffix, what = prec.split(":", 2)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]) * 1024
per = {k: (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) / steps for k, c in agg.items()}
total = sum(per.values())
fetch = sum(c.get("FETCH_SIZE", 0.0) for c in agg.values()) / steps
write = sum(c.get("WRITE_SIZE", 0.0) for c in agg.values()) / steps
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from moda_amd.build import source_hash  # noqa: E402
tpath = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
t["train_step_" + prec + suffix] = {"hbm_bytes_per_step": total, "fetch_bytes_raw": fetch, "write_bytes": write, "steps_in_run": steps,
                            "kernel_source_sha16": source_hash(),
                            "note": f"sum over every kernel of one eagerly launched {what}, "
                                    "2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes (tools/pmc_train_step.sh)",
                            "largest": {k[:90]: v for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:10]}}
json.dump(t, open(tpath, "w"), indent=1)
print(f"train step [{prec}{suffix}]: {total / 1e9:.3f} GB per step (fetch raw {fetch / 1e9:.3f}, write {write / 1e9:.3f})")
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:10]:
    print(f"   {v / 1e6:9.1f} MB  {k[:110]}")
