#!/bin/bash
# Run ON THE GPU BOX (via gpurun): where the hot kernels' traffic beyond the CUs is served from -- L2 hit rate (TCC_HIT / TCC_MISS),
# fabric-side read requests (TCC_EA0_RDREQ, 32-byte share) and the part of them that goes to DRAM rather than the Infinity Cache
# (TCC_EA0_RDREQ_DRAM), the same for writes.  Three separate --pmc passes per kernel (TCC has 4 slots; no trace domains).
#   tools/pmc_l2.sh r03   ->  gpurun_out/r03_pmc_l2.txt
set -e
tag=${1:-pmc}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
P2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum"
P3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
for k in gemm attn; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $out/${tag}_l2_${k}_$i -o run -- python3 $root/tools/prof_$k.py > /dev/null 2> $out/${tag}_l2_${k}_$i.err
  done
  echo "$k passes done" >&2
done
cd $root
python3 - $out $tag > $out/${tag}_pmc_l2.txt <<'PY'
import collections, csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
def load(d, match):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if match in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    v = collections.OrderedDict()
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            v[r["Counter_Name"]] = v.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return v
for k, match, what, operands in (("gemm", "gemm_pp_kernel", "gemm_pp_kernel, M=65520 N=K=5120 (one launch)", 65520 * 5120 * 2 + 5120 * 5120 * 2 + 65520 * 5120 * 2),
                                 ("attn", "attn_fwd_pipe_kernel", "attn_fwd_pipe_kernel, B=2 H=40 L=32760 (one launch)", 4 * 2 * 40 * 32760 * 128 * 2)):
    v = collections.OrderedDict()
    for i in (1, 2, 3):
        v.update(load(os.path.join(out, f"{tag}_l2_{k}_{i}"), match))
    print(f"== {what} ==")
    for n, x in v.items():
        print(f"   {n:28s} {x:.5g}")
    hit, miss = v["TCC_HIT_sum"], v["TCC_MISS_sum"]
    rd, rd32, rdd = v["TCC_EA0_RDREQ_sum"], v["TCC_EA0_RDREQ_32B_sum"], v["TCC_EA0_RDREQ_DRAM_sum"]
    wr, wr64, wrd = v["TCC_EA0_WRREQ_sum"], v["TCC_EA0_WRREQ_64B_sum"], v["TCC_EA0_WRREQ_DRAM_sum"]
    rd_bytes = rd32 * 32 + (rd - rd32) * 64
    print(f"   -> L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS)                 : {hit / (hit + miss):.3f}")
    print(f"   -> fabric-side read bytes (32-B x RDREQ_32B + 64-B x the rest)   : {rd_bytes / 1e9:.3f} GB  (x2 per MI355X_MICROARCH.md if these are 128-B requests tallied at 64 B: {2 * rd_bytes / 1e9:.3f} GB)")
    print(f"   -> share of read requests that go to DRAM (rest: Infinity Cache) : {rdd / rd:.3f}")
    print(f"   -> fabric-side write bytes (64-B x WRREQ_64B + 32-B x the rest)  : {(wr64 * 64 + (wr - wr64) * 32) / 1e9:.3f} GB ; share to DRAM {wrd / max(wr, 1):.3f}")
    print(f"   -> operand + result bytes of the launch (algorithmic)             : {operands / 1e9:.3f} GB")
    print()
PY
for k in gemm attn; do for i in 1 2 3; do rm -rf $out/${tag}_l2_${k}_$i; done; done
cat $out/${tag}_pmc_l2.txt
