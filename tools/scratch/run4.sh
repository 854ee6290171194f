#!/bin/bash
# scratch: batch 4 (full GPU suite on the new defaults, pass-limit sweep)
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=8 > $O/r5_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/r5_pytest.log
B="python bench.py --resident 131072 --batch-steps 512 --steps 3 --warmup 8 --no-cpu-baseline --no-full-launch"
run() { name=$1; cmd=$2; shift; shift; echo "== $name"; env "$@" timeout -k 10 240 $cmd > $O/r5_$name.json 2> $O/r5_$name.err || { echo "FAILED $name"; tail -5 $O/r5_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/r5_$name.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$name", round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), "gather frac", r.get("frac"))
PY
}
run p64 "$B" AR_X=0 && run p48 "$B" AR_GW_PASSES=48 && run p40 "$B" AR_GW_PASSES=40 && run p32 "$B" AR_GW_PASSES=32 && run p24 "$B" AR_GW_PASSES=24
