#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel stats + PMC passes) into the small text summaries kept
under profiles/.  Usage: summarize_profile.py <stats_dir> [<fetch_dir> <write_dir>] > profiles/rNN_xxx.txt

PMC notes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so the fetch side is doubled here ("corrected")."""
import collections
import csv
import re
import glob
import os
import sys


def find(d, suffix):
    fs = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None


def short(name):
    for k in ("attn_fwd_pipe_kernel", "attn_short_kernel", "qkv_front_kernel", "gemm_sw_kernel", "attn_fwd_kernel", "conv_igemm_kernel", "gemm_pp_kernel", "gemm_bf16_kernel", "t5_attention_kernel", "geoada_context_kernel", "layernorm_kernel", "rmsnorm_rope_kernel", "patchify_kernel",
              "unpatchify_kernel", "small_linear_kernel", "modulation_kernel", "axpy_kernel", "copy_strided_kernel"):
        if k in name:
            if k == "gemm_bf16_kernel":
                return k + ("<256x256>" if "256, 256" in name else "<128x128>")
            return k
    return name[:70]


def main():
    stats_dir = sys.argv[1]
    f = find(stats_dir, "_kernel_stats.csv")
    print("== rocprofv3 --kernel-trace --stats : kernel_stats (top 12 by total time) ==")
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':45s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>11s} {'pct':>7s}")
    # template instantiations of one kernel (the GEMM's epilogue kinds) are one line: the bench line's per-kernel averages are
    # over the family as well
    fam = {}
    for r in rows:
        k = short(r["Name"])
        e = fam.setdefault(k, [0, 0, 0.0, []])
        e[0] += int(r["Calls"]); e[1] += int(r["TotalDurationNs"]); e[2] += float(r["Percentage"])
        e[3].append(r)
    for k, e in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:45s} {e[0]:7d} {e[1] / 1e6:11.2f} {e[1] / e[0] / 1e3:11.1f} {e[2]:7.2f}")
        if len(e[3]) > 1 and k == "gemm_pp_kernel":      # the epilogue kinds of the production GEMM
            for r in e[3]:
                m = re.search(r"_kernel<([^>]*)>", r["Name"])
                print(f"{'    <' + (m.group(1) if m else '?') + '>':45s} {int(r['Calls']):7d} {int(r['TotalDurationNs']) / 1e6:11.2f} "
                      f"{float(r['AverageNs']) / 1e3:11.1f} {float(r['Percentage']):7.2f}")
    tr = find(stats_dir, "_kernel_trace.csv")
    if tr:
        att = [r for r in csv.DictReader(open(tr)) if "attn_fwd" in r["Kernel_Name"]]
        att.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in att]
        big = [d for d in dur if d > 5e6]
        small = [d for d in dur if d <= 5e6]
        if big:
            print(f"attention self-attention launches : {len(big)} avg {sum(big) / len(big) / 1e6:.3f} ms")
        if small:
            print(f"attention cross-attention launches: {len(small)} avg {sum(small) / len(small) / 1e6:.3f} ms")
    if len(sys.argv) >= 4:
        print()
        print("== PMC passes (separate runs): HBM-side traffic per launch, GiB ==")
        agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
        for d, ctr in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
            rows = list(csv.DictReader(open(find(d, "_counter_collection.csv"))))
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            seen = collections.Counter()
            # cross-attention has its own kernel (attn_short_kernel) when the folded prompt fits in LDS: then every launch of
            # the pipelined kernel is a self-attention; otherwise the two alternate inside every DiT block
            alternate = not any("attn_short_kernel" in r["Kernel_Name"] for r in rows)
            for r in rows:
                if r["Counter_Name"] != ctr:
                    continue
                k = short(r["Kernel_Name"])
                if k.startswith("attn_fwd"):
                    base = k
                    k += "[self]" if (not alternate or seen[base] % 2 == 0) else "[cross]"
                    seen[base] += 1
                agg[k][ctr] += float(r["Counter_Value"])
                if ctr == "FETCH_SIZE":
                    agg[k]["n"] += 1
        print(f"{'kernel':40s} {'launches':>8s} {'fetch_raw':>10s} {'fetch_corr(x2)':>14s} {'write':>10s}")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])[:8]:
            n = max(v["n"], 1)
            fr = v["FETCH_SIZE"] * 1024 / n / 2 ** 30
            wr = v["WRITE_SIZE"] * 1024 / n / 2 ** 30
            print(f"{k:40s} {n:8d} {fr:10.3f} {2 * fr:14.3f} {wr:10.3f}")


if __name__ == "__main__":
    main()
