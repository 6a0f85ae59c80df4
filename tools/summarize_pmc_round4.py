#!/usr/bin/env python3
"""Condenses the --pmc passes of tools/pmc_round4.sh (last dispatch of the named kernel in each pass; counters summed over the device)."""
import collections
import csv
import glob
import os
import sys


def load(d, match):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    vals = collections.OrderedDict()
    if not f:
        return vals
    rows = [r for r in csv.DictReader(open(f[0])) if match in r["Kernel_Name"]]
    if not rows:
        return vals
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    vals["_res"] = (rows[0].get("VGPR_Count"), rows[0].get("Accum_VGPR_Count"), rows[0].get("LDS_Block_Size"), rows[0].get("Scratch_Size"))
    return vals


def main():
    out, tag = sys.argv[1], sys.argv[2]
    g = lambda d, k: d.get(k, float("nan"))
    P = {p: load(os.path.join(out, f"{tag}_pmc_af8_{p}"), "attn_fp8_kernel") for p in "ABCD"}
    a, b, c, d = P["A"], P["B"], P["C"], P["D"]
    print("== attn_fp8_kernel<1> (fp8 self-attention, B=2 H=40 L=32760; 43.96 TFLOP algorithmic per launch) ==")
    if "_res" in a:
        print(f"   VGPRs {a['_res'][0]} (+{a['_res'][1]} acc), LDS {a['_res'][2]} B / workgroup, scratch {a['_res'][3]} B / lane")
    for dd in (a, b, c, d):
        for k, v in dd.items():
            if not k.startswith("_"):
                print(f"   {k:28s} {v:.5g}")
    wc = g(a, "SQ_WAVE_CYCLES")
    cyc = g(b, "GRBM_GUI_ACTIVE") / 8.0
    print(f"   -> kernel cycles (GRBM_GUI_ACTIVE / 8)                 : {cyc:.4g}")
    print(f"   -> MFMA pipe occupancy = MFMA_BUSY / (1024 x cycles)   : {g(a, 'SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc):.3f}")
    print(f"   -> VALU instructions per MFMA                          : {g(a, 'SQ_INSTS_VALU') / g(a, 'SQ_INSTS_MFMA'):.2f}")
    print(f"   -> wave cycles waiting (SQ_WAIT_INST_ANY / WAVE_CYCLES): {g(a, 'SQ_WAIT_INST_ANY') / wc:.3f}")
    print(f"   -> wave cycles parked  (SQ_WAIT_ANY / WAVE_CYCLES)     : {g(a, 'SQ_WAIT_ANY') / wc:.3f}")
    print(f"   -> wave cycles issuing (SQ_ACTIVE_INST_ANY / WAVE_CYC) : {g(a, 'SQ_ACTIVE_INST_ANY') / wc:.3f}")
    print(f"   -> LDS bank-conflict cycles / LDS active cycles        : {g(b, 'SQ_LDS_BANK_CONFLICT') / g(b, 'SQ_LDS_IDX_ACTIVE'):.4f}")
    print(f"   -> LDS array active / (256 CUs x cycles)               : {g(b, 'SQ_LDS_IDX_ACTIVE') / (256 * cyc):.3f}")
    hit, miss = g(c, "TCC_HIT_sum"), g(d, "TCC_MISS_sum")
    print(f"   -> L2 hit rate                                          : {hit / (hit + miss):.3f}")
    print(f"   -> fabric-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB counters): {(2 * g(c, 'FETCH_SIZE') + g(d, 'WRITE_SIZE')) * 1024 / 1e9:.2f} GB")
    print()
    print("== gemm_pp_kernel, M = 65520: production tile map (tile 4: eight XCD bands) vs all XCDs on one 64-row super-band (tile 6) ==")
    for n, k in ((5120, 5120), (13824, 5120)):
        for t in (4, 6):
            cc = load(os.path.join(out, f"{tag}_pmc_gm_{t}_{n}_C"), "gemm_pp_kernel")
            dd = load(os.path.join(out, f"{tag}_pmc_gm_{t}_{n}_D"), "gemm_pp_kernel")
            hit, miss = g(cc, "TCC_HIT_sum"), g(dd, "TCC_MISS_sum")
            cyc = g(cc, "GRBM_GUI_ACTIVE") / 8.0
            print(f"   N={n:5d} K={k} tile {t}: FETCH_SIZE x2 {2 * g(cc, 'FETCH_SIZE') * 1024 / 1e9:6.2f} GB  WRITE_SIZE {g(dd, 'WRITE_SIZE') * 1024 / 1e9:5.2f} GB  "
                  f"L2 hit {hit / (hit + miss):.3f}  kernel cycles {cyc:.4g}")


if __name__ == "__main__":
    main()
