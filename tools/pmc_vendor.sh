#!/bin/bash
# Run ON THE GPU BOX: the same counters as tools/pmc_kernels.sh for the vendor GEMM (torch F.linear -> hipBLASLt) next to ours on
# the three engine shapes -- reference point only: at what clock and matrix-pipe occupancy does the vendor kernel run?
set -e
tag=${1:-pmcv}
root=$(pwd)
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
B="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
rocprofv3 --pmc $A --output-format csv -d $out/${tag}_a -o run -- python3 $root/tools/bench_vendor_gemm.py nosdpa > /dev/null 2> $out/${tag}_a.err
rocprofv3 --pmc $B --output-format csv -d $out/${tag}_b -o run -- python3 $root/tools/bench_vendor_gemm.py nosdpa > /dev/null 2> $out/${tag}_b.err
cd $root
python3 - $out/${tag}_a $out/${tag}_b > $out/${tag}_vendor_gemm_pmc.txt <<'PY'
import collections, csv, glob, os, sys
def load(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_pp_kernel" in n: k = "ours: gemm_pp_kernel"
        elif n.startswith("Cijk") or "Cijk_" in n: k = "vendor: " + n[:60]
        else: continue
        key = (k, r["Grid_Size"])
        per.setdefault(key, collections.defaultdict(float))
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        per[key]["_n_" + r["Counter_Name"]] += 1
    return per
a, b = load(sys.argv[1]), load(sys.argv[2])
print("kernel | grid | launches | cycles/launch (GRBM_GUI_ACTIVE/8) | MFMA pipe occupancy at the held clock | LDS array active | waiting/wave cycles")
for key in a:
    if key not in b: continue
    n = a[key]["_n_SQ_WAVE_CYCLES"]; nb = b[key]["_n_GRBM_GUI_ACTIVE"]
    cyc = b[key]["GRBM_GUI_ACTIVE"] / 8.0 / nb
    occ = a[key]["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024 * cyc)
    lds = b[key]["SQ_LDS_IDX_ACTIVE"] / nb / (256 * cyc)
    wait = a[key]["SQ_WAIT_INST_ANY"] / a[key]["SQ_WAVE_CYCLES"]
    print(f"{key[0]} | {key[1]} | {int(n)} | {cyc:.4g} | {occ:.3f} | {lds:.3f} | {wait:.3f}")
PY
rm -rf $out/${tag}_a $out/${tag}_b
cat $out/${tag}_vendor_gemm_pmc.txt
