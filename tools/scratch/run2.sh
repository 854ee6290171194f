#!/bin/bash
# scratch: experiment batch 2 (CNN register kernel, late advance, kernel trace)
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "device_net_matches or cnn_with_trunk or mlp_first_layer" > $O/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/r2_pytest.log
B="python bench.py --resident 131072 --batch-steps 512 --steps 3 --warmup 8 --no-cpu-baseline --no-full-launch"
CN="python bench.py --evaluator cnn --resident 4096 --warmup-batch-steps 200 --batch-steps 32 --steps 3 --no-cpu-baseline --no-full-launch"
run() { name=$1; cmd=$2; shift; shift; echo "== $name"; env "$@" timeout -k 10 240 $cmd > $O/r2_$name.json 2> $O/r2_$name.err || { echo "FAILED $name"; tail -5 $O/r2_$name.err; return 1; }; python - <<PY
import json
d=json.loads(open("$O/r2_$name.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$name", round(d["value"]/1e6,2), "Msims/s", round(d["ms_per_step"],1), "ms/step evals/s", round(d.get("nn_evals_per_sec",0)/1e6,3), "gather frac", r.get("frac"))
PY
}
run cnn_new "$CN" AR_X=0 && run cnn_old "$CN" AR_CNN_LDS=1 && run advlate "$B" AR_ADV_LATE=1 && run advlate_eo "$B" AR_ADV_LATE=1 AR_CUMASK=evenodd AR_GW_WAVES=896 && run advlate_p80 "$B" AR_ADV_LATE=1 AR_GW_PASSES=80 || exit 1
cd /tmp && export TMPDIR=/tmp
echo "== trace default"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --resident 131072 --batch-steps 256 --steps 2 --warmup 12 --no-cpu-baseline --no-full-launch > $GRAFT_REPO_ROOT/$O/r2_trace.json 2> $GRAFT_REPO_ROOT/$O/r2_trace.err && python3 $GRAFT_REPO_ROOT/tools/kernel_stats.py /tmp/kt > $GRAFT_REPO_ROOT/$O/r2_trace_stats.csv; head -12 $GRAFT_REPO_ROOT/$O/r2_trace_stats.csv
echo "== pmc cnn"
C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -k 10 200 rocprofv3 --pmc $C -d /tmp/pm_cnn -o c --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --evaluator cnn --resident 4096 --warmup-batch-steps 100 --batch-steps 8 --steps 1 --no-cpu-baseline --no-full-launch > $GRAFT_REPO_ROOT/$O/r2_pm_cnn.json 2> $GRAFT_REPO_ROOT/$O/r2_pm_cnn.err && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm_cnn > $GRAFT_REPO_ROOT/$O/r2_pmc_sq_cnn.txt; head -8 $GRAFT_REPO_ROOT/$O/r2_pmc_sq_cnn.txt
