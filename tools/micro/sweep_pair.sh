set -o pipefail
for cfg in "quad" "octet3" "quad" "octet3"; do
  AR_GATHER=$cfg timeout -k 10 300 python3 bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-full-launch 2> gpurun_out/cf_$cfg.err > gpurun_out/cf_$cfg.json || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/cf_$cfg.json')); r=d['roofline']; print('$cfg', round(d['value']/1e6,1), 'M sims/s  launch', round(r['avg_launch_ms'],3), 'step', round(r['step']['avg_step_ms'],3), 'desc/s', round(d['descents_per_sec']/1e6,2))" | tee -a gpurun_out/sweep.txt
done
