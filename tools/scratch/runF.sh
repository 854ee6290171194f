#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
echo "== driver command"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/F_default.json 2> $O/F_default.err; echo "rc=$?"; tail -c 400 $O/F_default.json
cd /tmp && export TMPDIR=/tmp
echo "== trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-full-launch > $O/F_trace.json 2> $O/F_trace.err && python3 $R/tools/kernel_stats.py /tmp/kt > $O/F_kernel_stats.csv && python3 $R/tools/timeline.py /tmp/kt 16 > $O/F_timeline.txt; head -9 $O/F_kernel_stats.csv
