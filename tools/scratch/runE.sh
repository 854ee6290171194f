#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_nets.py -x -q -m gpu > $O/E_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/E_pytest.log
B="python bench.py --resident 131072 --batch-steps 512 --steps 3 --warmup 8 --no-cpu-baseline --no-full-launch"
run() { name=$1; cmd=$2; shift; shift; echo "== $name"; env "$@" timeout -k 10 300 $cmd > $O/E_$name.json 2> $O/E_$name.err || { echo "FAILED $name"; tail -5 $O/E_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/E_$name.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$name", round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), "gather frac", r.get("frac"))
PY
}
run new1 "$B" AR_X=0 && run prev1 "$B" AR_LIB=$GRAFT_REPO_ROOT/alpharat_amd/libalpharat_hip_prev.so && run new2 "$B" AR_X=0 && run prev2 "$B" AR_LIB=$GRAFT_REPO_ROOT/alpharat_amd/libalpharat_hip_prev.so
