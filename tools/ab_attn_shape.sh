#!/bin/bash
# Run ON THE GPU BOX (via gpurun): MFMA-shape A/B of the self-attention kernel -- wall time (interleaved rounds in one process,
# random data) and hardware counters (two separate --pmc passes, no trace domains) of both variants -> gpurun_out/<tag>_attn_shape_ab.txt
set -e
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
res=$out/${tag}_attn_shape_ab.txt
echo "== wall: python tools/ab_attn_shape.py 10 (HIP events, interleaved rounds, one process, gaussian q/k/v) ==" > $res
python3 tools/ab_attn_shape.py 10 >> $res 2>/dev/null
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
B="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
rocprofv3 --pmc $A --output-format csv -d $out/${tag}_shape_a -o run -- python3 $root/tools/prof_attn_shapes.py > /dev/null 2> $out/${tag}_shape_a.err
rocprofv3 --pmc $B --output-format csv -d $out/${tag}_shape_b -o run -- python3 $root/tools/prof_attn_shapes.py > /dev/null 2> $out/${tag}_shape_b.err
cd $root
python3 - $out/${tag}_shape_a $out/${tag}_shape_b >> $res <<'PY'
import collections, csv, glob, os, sys
def load(d, match):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if match in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    vals = collections.OrderedDict()
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    vals["_regs"] = (rows[0].get("VGPR_Count"), rows[0].get("Accum_VGPR_Count"), rows[0].get("Scratch_Size"))
    return vals
print()
print("== counters (rocprofv3 --pmc, two passes; last launch of each kernel; B=2 H=40 L=32760) ==")
res = {}
for name, match in (("32x32x16", "attn_fwd_pipe_kernel"), ("16x16x32", "attn_fwd_pipe16_kernel")):
    a, b = load(sys.argv[1], match), load(sys.argv[2], match)
    cyc = b["GRBM_GUI_ACTIVE"] / 8.0
    res[name] = dict(wave_cycles=a["SQ_WAVE_CYCLES"], cycles=cyc, mfma_occ=a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                     valu=a["SQ_INSTS_VALU"], mfma=a["SQ_INSTS_MFMA"], wait=a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
                     lds_active=b["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), conflicts=b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"],
                     regs=a["_regs"])
for k, v in res.items():
    print(f"{k}: SQ_WAVE_CYCLES {v['wave_cycles']:.4g}  kernel cycles (GRBM_GUI_ACTIVE/8) {v['cycles']:.4g}  MFMA pipe occupancy {v['mfma_occ']:.3f}  "
          f"VALU insts {v['valu']:.4g}  MFMA insts {v['mfma']:.4g}  waiting/wave-cycles {v['wait']:.3f}  LDS array active {v['lds_active']:.3f}  "
          f"bank conflicts {v['conflicts']:.4f}  VGPR/AGPR/scratch {v['regs']}")
a, b = res["32x32x16"], res["16x16x32"]
print(f"16x16x32 / 32x32x16: SQ_WAVE_CYCLES x{b['wave_cycles'] / a['wave_cycles']:.3f}, kernel cycles x{b['cycles'] / a['cycles']:.3f}, "
      f"VALU instructions x{b['valu'] / a['valu']:.3f}")
PY
rm -rf $out/${tag}_shape_a $out/${tag}_shape_b
cat $res
