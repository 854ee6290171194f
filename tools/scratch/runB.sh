#!/bin/bash
# scratch: MLP tile A/B + steady-state lines of configs 4 and 5
cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python bench.py --resident 131072 --batch-steps 512 --steps 3 --warmup 8 --no-cpu-baseline --no-full-launch"
run() { name=$1; cmd=$2; shift; shift; echo "== $name"; env "$@" timeout -k 10 400 $cmd > $O/B_$name.json 2> $O/B_$name.err || { echo "FAILED $name"; tail -5 $O/B_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/B_$name.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$name", round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), "games/s", round(d.get("games_per_sec",0),1), "gather frac", r.get("frac"), d.get("window"))
PY
}
run mt2 "$B" AR_X=0 && run mt1 "$B" AR_MLP_MT=1 && run sym "python bench.py --evaluator symmetric --warmup-batch-steps 3400 --batch-steps 256 --steps 4 --no-cpu-baseline --no-full-launch" AR_X=0 && run cnn "python bench.py --evaluator cnn --resident 4096 --warmup-batch-steps 8000 --batch-steps 64 --steps 6 --no-cpu-baseline --no-full-launch" AR_X=0
