"""Post-process gpurun_out/<tag>/pmc_*_counter_collection.csv into profiles/<name>_pmc_traffic_cfg2.txt and
profiles/latest_traffic.json (per-launch HBM-side bytes per kernel, gfx950 corrections of MI355X_MICROARCH.md:
FETCH_SIZE under-reports wide coalesced reads by 2x, WRITE_SIZE is exact; both in KiB)."""
import csv, glob, json, os, sys
from collections import defaultdict

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
SHORT = [("pmlp_fused_fwd_kernel<5, 0, 0>", "pmlp_fused_fwd"),  # the headline forward (stencil, native fp32 MFMA)
         ("pmlp_fused_fwd_kernel<5, 0, 1>", "pmlp_fused_fwd_bf16x3"),  # bench.py's side measurement
         ("w0_split", "w0_split_bf16x3"), ("pmlp_fused_wgrad", "pmlp_fused_wgrad"),
         ("pmlp_fused_bwd_chain", "pmlp_fused_bwd_chain"), ("fourier_stencil", "fourier_stencil"),
         ("evd_partial", "evd_partial"), ("rmsprop_ema", "rmsprop_ema"), ("distribution_elementwise", "torch_randn")]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        short = next((s for pat, s in SHORT if pat in k), None)
        if short is None:
            continue
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [f"# rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events",
         "# separate passes for FETCH_SIZE, WRITE_SIZE, (TCC_HIT_sum TCC_MISS_sum); MI355X, cfg2; per-launch averages",
         "# units: FETCH_SIZE / WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 correction",
         "# (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads; WRITE_SIZE exact for 16-B stores)",
         f"{'kernel':<28}{'launches':>9}{'FETCH_KiB':>13}{'WRITE_KiB':>13}{'L2 hit':>9}{'hbm_MB(corrected)':>20}"]
out = {"workload": "cfg2", "source": f"profiles/{name}_pmc_traffic_cfg2.txt", "kernels": {}}
avg = lambda v: sum(v) / len(v) if v else float("nan")
for _, s in SHORT:
    if s not in vals:
        continue
    c = vals[s]
    fetch, write = avg(c.get("FETCH_SIZE", [])), avg(c.get("WRITE_SIZE", []))
    hit, miss = avg(c.get("TCC_HIT_sum", [])), avg(c.get("TCC_MISS_sum", []))
    rate = hit / (hit + miss) if hit == hit and hit + miss > 0 else float("nan")
    hbm = (2 * fetch + write) * 1024
    lines.append(f"{s:<28}{len(c.get('FETCH_SIZE', [])):>9}{fetch:>13.1f}{write:>13.1f}{rate:>9.3f}{hbm / 1e6:>20.1f}")
    out["kernels"][s] = dict(fetch_kib=fetch, write_kib=write, l2_hit=rate, hbm_bytes_corrected=hbm)
open(os.path.join(root, "profiles", f"{name}_pmc_traffic_cfg2.txt"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(root, "profiles", "latest_traffic.json"), "w"), indent=1)
print("\n".join(lines))
