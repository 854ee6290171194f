#!/bin/bash
# scratch: experiment batch 1 (evaluator staging, CU masks, advance overlap)
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline_parity.py -x -q -m gpu --durations=5 > $O/r1_pytest.log 2>&1 || { tail -30 $O/r1_pytest.log; exit 1; }
tail -8 $O/r1_pytest.log
B="python bench.py --resident 131072 --batch-steps 512 --steps 3 --warmup 8 --no-cpu-baseline --no-full-launch"
run() { name=$1; shift; echo "== $name"; env "$@" timeout -k 10 200 $B > $O/r1_$name.json 2> $O/r1_$name.err || { echo "FAILED $name"; tail -5 $O/r1_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/r1_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], json.dumps(d.get("roofline",{}))[:300])
PY
}
run default AR_X=0 && run xregs AR_MLP_XREGS=1 && run lohi AR_CUMASK=lohi AR_GW_WAVES=896 && run evenodd AR_CUMASK=evenodd AR_GW_WAVES=896 && run lohi192 AR_CUMASK=lohi AR_GW_WAVES=896 AR_GW_PASSES=192 && run noadvoverlap AR_NO_ADVANCE_OVERLAP=1
