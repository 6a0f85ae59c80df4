#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of ONE rank's share of a P-way run (tools/sim_sp_rank.py P; exchanges simulated).
#   tools/profile_sim_rank.sh 8 r02_sim8   ->  gpurun_out/r02_sim8_rocprof_summary.txt
set -e
P=${1:-8}; tag=${2:-sim$P}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o run -- python3 $root/tools/sim_sp_rank.py $P > $out/${tag}.json 2> $out/${tag}.err
cd $root
python3 - "$out/${tag}_stats" "$P" > $out/${tag}_rocprof_summary.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"== rocprofv3 --kernel-trace --stats of tools/sim_sp_rank.py {sys.argv[2]} (ONE rank of a {sys.argv[2]}-way run; exchanges = local copy + delay kernel) ==")
print("%-64s %7s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in rows[:16]:
    print("%-64s %7d %10.2f %10.1f %6.2f" % (r["Name"][:64], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $out/${tag}_stats
cat $out/${tag}_rocprof_summary.txt
