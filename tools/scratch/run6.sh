#!/bin/bash
# scratch: batch 6: long windows (several generations of games) for the pass limit
cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-launch"
run() { name=$1; cmd=$2; shift; shift; echo "== $name"; env "$@" timeout -k 10 400 $cmd > $O/r6_$name.json 2> $O/r6_$name.err || { echo "FAILED $name"; tail -5 $O/r6_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/r6_$name.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$name", round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), "games/s", round(d.get("games_per_sec",0),1), "ratio", round(d["value"]/max(d.get("nn_evals_per_sec",1),1),3), "gather frac", r.get("frac"), d.get("window"))
PY
}
run p64 "$B" AR_X=0 && run p48 "$B" AR_GW_PASSES=48 && run p96 "$B" AR_GW_PASSES=96 && run p32 "$B" AR_GW_PASSES=32
