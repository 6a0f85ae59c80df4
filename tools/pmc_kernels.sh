#!/bin/bash
# Run ON THE GPU BOX (via gpurun): hardware counters of the two hot kernels in isolation (cfg-3 shapes), two separate --pmc passes
# each (no trace domains combined with --pmc), condensed into gpurun_out/<tag>_pmc_hot_kernels.txt.   tools/pmc_kernels.sh r02
set -e
tag=${1:-pmc}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
B="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
for k in attn gemm; do
  rocprofv3 --pmc $A --output-format csv -d $out/${tag}_pmc_${k}_a -o run -- python3 $root/tools/prof_$k.py > /dev/null 2> $out/${tag}_pmc_${k}_a.err
  rocprofv3 --pmc $B --output-format csv -d $out/${tag}_pmc_${k}_b -o run -- python3 $root/tools/prof_$k.py > /dev/null 2> $out/${tag}_pmc_${k}_b.err
  echo "$k passes done" >&2
done
cd $root
python3 tools/summarize_pmc.py $out/${tag}_pmc_attn_a $out/${tag}_pmc_attn_b $out/${tag}_pmc_gemm_a $out/${tag}_pmc_gemm_b > $out/${tag}_pmc_hot_kernels.txt
rm -rf $out/${tag}_pmc_attn_a $out/${tag}_pmc_attn_b $out/${tag}_pmc_gemm_a $out/${tag}_pmc_gemm_b
cat $out/${tag}_pmc_hot_kernels.txt
