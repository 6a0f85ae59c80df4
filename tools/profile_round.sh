#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats pass + two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# of the default bench command, condensed into gpurun_out/<tag>_rocprof_summary.txt.   tools/profile_round.sh r01_v6 [extra bench.py flags, e.g. --fp8-linear --fp8-attn 1]
set -e
tag=${1:-prof}
extra="${@:2}"
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cmd="python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-teacache-line $extra"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o run -- $cmd > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
echo "stats pass done" >&2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o run -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-teacache-line --no-profile $extra > /dev/null 2> $out/${tag}_fetch.err
echo "fetch pass done" >&2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -o run -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-teacache-line --no-profile $extra > /dev/null 2> $out/${tag}_write.err
echo "write pass done" >&2
cd $root
python3 tools/summarize_profile.py $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write > $out/${tag}_rocprof_summary.txt
# raw rocprof directories are large: keep only the summary and the bench line
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
cat $out/${tag}_rocprof_summary.txt
