# usage: run_prof_one.sh <tag> <bench args...>
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python bench.py "$@" > $out/trace.log 2>&1
python tools/prof_summary.py $out/trace/${tag}_kernel_stats.csv 14
find $out -name "*kernel_trace.csv" -delete
