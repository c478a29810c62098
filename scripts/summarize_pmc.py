"""summarize_pmc.py <tag> <name> [workload = cfg2]
Post-process gpurun_out/<tag>/pmc_*_counter_collection.csv into profiles/<name>_pmc_traffic_<workload>.txt and - for
the headline workload cfg2 only - profiles/latest_traffic.json (per-launch HBM-side bytes per kernel, gfx950 corrections of MI355X_MICROARCH.md:
FETCH_SIZE under-reports wide coalesced reads by 2x, WRITE_SIZE is exact; both in KiB) and, from the SQ pass,
the MFMA utilisation per kernel:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (64 cycles per v_mfma_f32_32x32x2_f32: the forward
kernel reads exactly 64 x its MFMA count); SQ_BUSY_CYCLES is summed over the 32 shader engines, so
SQ_BUSY_CYCLES / 32 is the dispatch's length in shader cycles (it agrees with End - Start timestamps at the
~2.1 GHz the chip holds under a profiled run) and 1024 * that the SIMD-cycles on offer."""
import csv, glob, json, os, sys
from collections import defaultdict

tag, name = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
SHORT = [("pmlp_fused_fwd_kernel<5, 0, 0", "pmlp_fused_fwd"),  # the headline forward (stencil, native fp32 MFMA)
         ("pmlp_fused_fwd_kernel<5, 0, 1", "pmlp_fused_fwd_bf16x3"),  # bench.py's side measurement
         ("w0_split", "w0_split_bf16x3"), ("pmlp_fused_wgrad", "pmlp_fused_wgrad"),
         ("pmlp_fused_bwd_chain", "pmlp_fused_bwd_chain"), ("fourier_stencil", "fourier_stencil"),
         ("evd_partial", "evd_partial"), ("rmsprop_ema", "rmsprop_ema"), ("distribution_elementwise", "torch_randn"),
         ("pmlp_fused_fwd_kernel<1, 0, 0", "pmlp_fused_fwd_E1"), ("pmlp_fused_fwd_kernel<4, 0, 0, 1>", "pmlp_fused_fwd_plain4"), ("wgrad_reduce", "wgrad_reduce"), ("pmlp_stream_bwd", "pmlp_stream_bwd"), ("pmlp_plain_stream_fwd", "pmlp_plain_stream_fwd"), ("ka_gemm_dma", "ka_gemm_dma"),
         ("tower_col_kernel<false", "tower_col_fwd"), ("tower_col_kernel<true", "tower_col_bwd"),
         ("tower_gemm_nt", "tower_gemm_nt"), ("gemm16b_kernel<true, true", "gemm16b_wgrad"), ("gemm16b_kernel<false, false, true", "gemm16b_fwd1"), ("gemm16b_kernel<false, true", "gemm16b_dA1"),
         ("gemm16_kernel<true, true", "gemm16_wgrad"), ("gemm16_kernel<false, false, true", "gemm16_fwd1"),
         ("gemm16_kernel<false, true", "gemm16_dA1"), ("gemm16_kernel<false, false, false", "gemm16_fwd2"),
         ("tower_bn16_forward", "tower_bn16_forward"), ("tower_bn16_backward", "tower_bn16_backward"), ("narrow_", "narrow_end_kernels"),
         ("cdk_sgd", "cdk_sgd"), ("tower_bn_forward", "tower_bn_forward"),
         ("tower_bn_backward", "tower_bn_backward"), ("tower_transpose", "tower_transpose"),
         ("tower_sum_slices", "tower_sum_slices"), ("cdk_", "cdk_loss_kernels"), ("ka_gemm", "ka_gemm"),
         ("ka_scatter", "ka_scatter"), ("ka_zero", "ka_zero"), ("ka_reduce", "ka_reduce"), ("row_normalize", "row_normalize")]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        short = next((s for pat, s in SHORT if pat in k), None)
        if short is None:
            continue
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
cfg_arg = "" if workload == "cfg2" else f" --config {workload}"
lines = [f"# rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events{cfg_arg}",
         f"# separate passes for FETCH_SIZE, WRITE_SIZE, (TCC_HIT_sum TCC_MISS_sum); MI355X, {workload}; per-launch averages",
         "# units: FETCH_SIZE / WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 correction",
         "# (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads; WRITE_SIZE exact for 16-B stores)",
         f"{'kernel':<28}{'launches':>9}{'FETCH_KiB':>13}{'WRITE_KiB':>13}{'L2 hit':>9}{'hbm_MB(corrected)':>20}"]
lines2 = ["", "# SQ pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE",
          "# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES) (MFMA-pipe-busy SIMD cycles over all 1024 SIMDs x dispatch cycles);",
          "# wait_inst / wait_any / active: shares of SQ_WAVE_CYCLES (quad-cycles: issue stall, parked at waitcnt/barrier, issuing)",
          f"{'kernel':<28}{'launches':>9}{'MFMA_BUSY':>14}{'SQ_BUSY/32':>12}{'mfma_busy_frac':>16}{'wait_inst':>11}{'wait_any':>10}{'active':>8}"]
out = {"workload": workload, "source": f"profiles/{name}_pmc_traffic_{workload}.txt", "kernels": {}}
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
    if c.get("SQ_BUSY_CYCLES"):
        mb, sb, wc = avg(c["SQ_VALU_MFMA_BUSY_CYCLES"]), avg(c["SQ_BUSY_CYCLES"]), avg(c.get("SQ_WAVE_CYCLES", []))
        fr = mb / (32.0 * sb) if sb > 0 else float("nan")
        sh = [avg(c.get(k, [])) / wc if wc == wc and wc > 0 else float("nan")
              for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")]
        lines2.append(f"{s:<28}{len(c['SQ_BUSY_CYCLES']):>9}{mb:>14.0f}{sb / 32:>12.0f}{fr:>16.3f}{sh[0]:>11.3f}{sh[1]:>10.3f}{sh[2]:>8.3f}")
        out["kernels"][s].update(mfma_busy_cycles=mb, sq_busy_cycles_per_se=sb / 32, mfma_busy_frac=fr,
                                 grbm_gui_active=avg(c.get("GRBM_GUI_ACTIVE", [])))
lines += lines2
open(os.path.join(root, "profiles", f"{name}_pmc_traffic_{workload}.txt"), "w").write("\n".join(lines) + "\n")
if workload == "cfg2":
    json.dump(out, open(os.path.join(root, "profiles", "latest_traffic.json"), "w"), indent=1)
print("\n".join(lines))
