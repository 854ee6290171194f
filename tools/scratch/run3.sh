#!/bin/bash
# scratch: experiment batch 3 (CNN / MLP correctness, first-layer variants, late advance + pass limit)
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_nets.py -x -q -m gpu --durations=5 > $O/r3_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/r3_pytest.log
B="python bench.py --resident 131072 --batch-steps 512 --steps 3 --warmup 8 --no-cpu-baseline --no-full-launch"
run() { name=$1; cmd=$2; shift; shift; echo "== $name"; env "$@" timeout -k 10 240 $cmd > $O/r3_$name.json 2> $O/r3_$name.err || { echo "FAILED $name"; tail -5 $O/r3_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/r3_$name.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$name", round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), "gather frac", r.get("frac"))
PY
}
run fl2 "$B" AR_X=0 && run fl2_late80 "$B" AR_ADV_LATE=1 AR_GW_PASSES=80 && run fl0_late80 "$B" AR_MLP_FL=0 AR_ADV_LATE=1 AR_GW_PASSES=80 && run fl2_late64 "$B" AR_ADV_LATE=1 AR_GW_PASSES=64 && run fl2_late72 "$B" AR_ADV_LATE=1 AR_GW_PASSES=72 && run fl2_late80_w2048 "$B" AR_ADV_LATE=1 AR_GW_PASSES=80 AR_GW_WAVES=2048
SY="python bench.py --evaluator symmetric --warmup-batch-steps 600 --batch-steps 64 --steps 3 --no-cpu-baseline --no-full-launch"
run sym "$SY" AR_X=0
