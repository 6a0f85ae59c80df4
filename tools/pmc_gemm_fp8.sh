#!/bin/bash
# Run ON THE GPU BOX (via gpurun): counters of gemm_pp_kernel<FP8> at the three cfg-3 shapes (separate --pmc passes) + the K sweep.
set -e
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
for shape in "5120 5120" "13824 5120" "5120 13824"; do
  n=${shape% *}; k=${shape#* }
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $out/${tag}_pmc_g8_${n}_${k}_A -o run -- python3 $root/tools/prof_gemm_fp8.py $n $k > /dev/null 2> $out/${tag}_pmc_g8_${n}_${k}_A.err
  echo "pass A $n $k done" >&2
  # FETCH_SIZE + TCC_HIT_sum + TCC_MISS_sum in ONE pass do not fit the TCC's counters (the run then sat in counter replay until the
  # silence guard killed it): two passes, as tools/pmc_round4.sh does
  rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/${tag}_pmc_g8_${n}_${k}_C -o run -- python3 $root/tools/prof_gemm_fp8.py $n $k > /dev/null 2> $out/${tag}_pmc_g8_${n}_${k}_C.err
  echo "pass C $n $k done" >&2
  rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_MISS_sum --output-format csv -d $out/${tag}_pmc_g8_${n}_${k}_D -o run -- python3 $root/tools/prof_gemm_fp8.py $n $k > /dev/null 2> $out/${tag}_pmc_g8_${n}_${k}_D.err
  echo "pass D $n $k done" >&2
done
cd $root
python3 - <<PY > $out/${tag}_gemm_fp8_pmc.txt
import csv, collections, glob, os
print("gemm_pp_kernel<FP8>, M = 65520, bias epilogue, gaussian operands quantised per row; 3 launches per pass, per-launch figures")
for n, k in ((5120, 5120), (13824, 5120), (5120, 13824)):
    c = collections.defaultdict(float); nl = collections.defaultdict(int)
    for p in "ACD":
        for r in csv.DictReader(open(f"$out/${tag}_pmc_g8_{n}_{k}_{p}/run_counter_collection.csv")):
            if "gemm_pp_kernel" in r["Kernel_Name"]:
                c[(p, r["Counter_Name"])] += float(r["Counter_Value"]); nl[(p, r["Counter_Name"])] += 1
    g = lambda p, name: c[(p, name)] / max(1, nl[(p, name)])
    cyc = g("A", "GRBM_GUI_ACTIVE") / 8
    flop = 2.0 * 65520 * n * k
    print(f"  N={n:6d} K={k:6d}: kernel cycles {cyc:.4g}  MFMA pipe occupancy {g('A', 'SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc):.3f}  "
          f"MFMA instr {g('A', 'SQ_INSTS_MFMA'):.4g}  FETCH x2 {2 * g('C', 'FETCH_SIZE') * 1024 / 1e9:.2f} GB  "
          f"L2 hit {g('C', 'TCC_HIT_sum') / max(1.0, g('C', 'TCC_HIT_sum') + g('D', 'TCC_MISS_sum')):.3f}  ({flop / 1e12:.2f} TFLOP per launch)")
PY
cat $out/${tag}_gemm_fp8_pmc.txt
for shape in "5120 5120" "13824 5120" "5120 13824"; do python3 tools/prof_gemm_fp8.py $shape ksweep 2>&1 | grep -v amdgpu.ids; done | tee $out/${tag}_gemm_fp8_ksweep.txt
