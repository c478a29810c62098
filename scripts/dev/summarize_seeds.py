"""dev: gpurun_out/<tag>/train_*_seed*.json (scripts/gpu/r04s5.sh) -> profiles/<tag>_seed_records.json and the
round-4 block of profiles/latest_accuracy.json (seeds_summary_round4)."""
import glob, json, os, statistics as st, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04s5"
src = os.path.join(root, "gpurun_out", tag)
groups = {"joint (configs[1])": "train_cfg2_fp32_seed*.json", "sequential": "train_cfg2_seq_seed*.json",
          "oscillator L=32 B=512 (configs[2] per GPU shape, 100000 steps)": "train_osc_B512_seed*.json",
          "joint, bf16x3 path": "train_cfg2_bf16x3_seed*.json", "joint, exact Laplacian": "train_cfg2_exact_seed*.json"}
rec, summ = {}, {}
for name, pat in groups.items():
    vals = []
    for f in sorted(glob.glob(os.path.join(src, pat))):
        d = json.load(open(f))
        last = d["evals"][-1] if "evals" in d else d
        rec[os.path.basename(f)] = {k: last.get(k) for k in ("step", "rel_err_mean", "rel_err_max", "steps_per_s", "train_seconds")}
        vals.append(last["rel_err_mean"])
    if vals:
        summ[name] = dict(n=len(vals), mean=st.mean(vals), stdev=st.stdev(vals) if len(vals) > 1 else 0.0, min=min(vals), max=max(vals))
json.dump(dict(what="full-schedule eigenvalue errors on the round-4 tree (stencil in even / odd form), scripts/gpu/r04s5.sh",
               records=rec, summary=summ), open(os.path.join(root, "profiles", f"{tag}_seed_records.json"), "w"), indent=1)
acc = json.load(open(os.path.join(root, "profiles", "latest_accuracy.json")))
acc["seeds_summary_round4"] = dict(summ, source=f"profiles/{tag}_seed_records.json",
                                   note="round-4 tree: every path carries the stencil in even / odd form (Tf at 1e-6 of the float64 stencil); "
                                        "seeds_summary above is the round-3 tree (point-wise float32 stencil)")
json.dump(acc, open(os.path.join(root, "profiles", "latest_accuracy.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
