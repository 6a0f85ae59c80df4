#!/bin/bash
# Run ON THE GPU BOX (via gpurun): counters of the round-4 kernels, separate --pmc passes (never combined with trace domains).
#   1. attn_fp8_kernel at the cfg-3 shape: issue / wait / MFMA / LDS / L2 counters
#   2. gemm_pp_kernel under the two tile maps (tile 4 production, tile 6 shared super-band) at 5120^2 and 5120->13824: FETCH_SIZE, L2 hits, clock
set -e
tag=${1:-r04}      # tools/pmc_round4.sh TAG [attn-only]
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
B="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
C="GRBM_GUI_ACTIVE FETCH_SIZE TCC_HIT_sum"
D="GRBM_GUI_ACTIVE WRITE_SIZE TCC_MISS_sum"
for p in A B C D; do
  eval cs=\$$p
  rocprofv3 --pmc $cs --output-format csv -d $out/${tag}_pmc_af8_$p -o run -- python3 $root/tools/prof_attn_fp8.py 1 > /dev/null 2> $out/${tag}_pmc_af8_$p.err
done
echo "attention passes done" >&2
[ "$2" = "attn-only" ] && { cd $root; python3 tools/summarize_pmc_round4.py $out $tag > $out/${tag}_pmc_round4.txt || true; cat $out/${tag}_pmc_round4.txt; exit 0; }
for t in 4 6; do
  for shape in "5120 5120" "13824 5120"; do
    n=${shape% *}
    for p in C D; do
      eval cs=\$$p
      rocprofv3 --pmc $cs --output-format csv -d $out/${tag}_pmc_gm_${t}_${n}_$p -o run -- python3 $root/tools/prof_gemm_map.py $t $shape > /dev/null 2> $out/${tag}_pmc_gm_${t}_${n}_$p.err
    done
  done
done
echo "gemm passes done" >&2
cd $root
python3 tools/summarize_pmc_round4.py $out $tag > $out/${tag}_pmc_round4.txt
cat $out/${tag}_pmc_round4.txt
