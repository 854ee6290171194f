#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 700 python -m pytest tests -x -q -m gpu --durations=6 > $O/D_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/D_pytest.log
echo "== cnn 16384 resident"
timeout -k 10 400 python bench.py --evaluator cnn --warmup-batch-steps 6000 --batch-steps 64 --steps 4 --no-cpu-baseline --no-full-launch > $O/D_cnn16k.json 2> $O/D_cnn16k.err; echo "rc=$?"; python - <<PY
import json
d=json.loads(open("$O/D_cnn16k.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), d["config"]["resident_games_per_gpu"], d.get("window"))
PY
