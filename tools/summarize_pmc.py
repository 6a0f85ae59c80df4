#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes of the hot-kernel micro-drivers (tools/pmc_kernels.sh) into derived utilisation figures.
Counters are summed over the device by rocprofv3; ratios below are per kernel launch (last launch of the kernel in the pass).
  MFMA pipe busy   = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES          (fraction of SIMD-busy cycles with the matrix pipe busy;
                     both are accumulated per SIMD / per SE as the guide describes -- treat as relative, compare builds)
  VALU active      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES * 4 (waves per SIMD share one issue port: reported raw as well)
  LDS active       = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES ; bank conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait on LDS      = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES ; waiting on anything = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"""
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
    vals["_vgpr"] = rows[0].get("VGPR_Count"), rows[0].get("Accum_VGPR_Count"), rows[0].get("LDS_Block_Size"), rows[0].get("Scratch_Size")
    return vals


def main():
    for name, match, da, db in (("attn_fwd_pipe_kernel (self-attention, B=2 H=40 L=32760)", "attn_fwd_pipe_kernel", sys.argv[1], sys.argv[2]),
                                ("gemm_pp_kernel (M=65520 N=K=5120)", "gemm_pp_kernel", sys.argv[3], sys.argv[4])):
        a, b = load(da, match), load(db, match)
        print(f"== {name} ==")
        if "_vgpr" in a:
            print(f"   VGPRs {a['_vgpr'][0]} (+{a['_vgpr'][1]} acc), LDS {a['_vgpr'][2]} B / workgroup, scratch {a['_vgpr'][3]} B / lane")
        for k, v in list(a.items()) + list(b.items()):
            if not k.startswith("_"):
                print(f"   {k:28s} {v:.4g}")
        g = lambda d, k: d.get(k, float("nan"))
        wc = g(a, "SQ_WAVE_CYCLES")
        cyc = g(b, "GRBM_GUI_ACTIVE") / 8.0            # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
        # SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipes are occupied, summed over the 1024 SIMDs (32 per 32x32x16 bf16 MFMA,
        # 16 per 16x16x32): divided by 1024 x the kernel's cycles it is the fraction of the dense MFMA peak at the clock the kernel held
        print(f"   -> kernel cycles (GRBM_GUI_ACTIVE / 8)             : {cyc:.4g}")
        print(f"   -> MFMA pipe occupancy = MFMA_BUSY / (1024 x cycles) : {g(a, 'SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc):.3f}")
        print(f"   -> MFMA instructions per VALU instruction            : {g(a, 'SQ_INSTS_MFMA') / g(a, 'SQ_INSTS_VALU'):.3f}")
        print(f"   -> wave cycles waiting on anything / wave cycles     : {g(a, 'SQ_WAIT_INST_ANY') / wc:.3f}")
        print(f"   -> wave cycles issuing / wave cycles                 : {g(a, 'SQ_ACTIVE_INST_ANY') / wc:.3f}")
        print(f"   -> LDS bank-conflict cycles / LDS active cycles      : {g(b, 'SQ_LDS_BANK_CONFLICT') / g(b, 'SQ_LDS_IDX_ACTIVE'):.4f}")
        print(f"   -> LDS array active / (256 CUs x cycles)             : {g(b, 'SQ_LDS_IDX_ACTIVE') / (256 * cyc):.3f}")
        print()


if __name__ == "__main__":
    main()
