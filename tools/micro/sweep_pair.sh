set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_nets.py -m gpu -x -q -k "scheduling" > gpurun_out/sched_test.log 2>&1; tail -5 gpurun_out/sched_test.log
grep -q " passed" gpurun_out/sched_test.log || exit 1
for cfg in "2 0" "2 2048" "2 3072" "2 1536" "3 2048" "1 2048"; do
  set -- $cfg
  AR_GROUPS=$1 AR_GATHER_GRID=$2 timeout -k 10 200 python3 bench.py --steps 2 --warmup 4 --batch-steps 512 --no-cpu-baseline --no-full-launch 2> gpurun_out/sw_$1_$2.err > gpurun_out/sw_$1_$2.json || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/sw_$1_$2.json')); r=d['roofline']; print('groups $1 grid $2', round(d['value']/1e6,1), 'M sims/s  launch', round(r['avg_launch_ms'],3), 'step', round(r['step']['avg_step_ms'],3), 'desc/s', round(d['descents_per_sec']/1e6,1))" | tee -a gpurun_out/sweep.txt
done
