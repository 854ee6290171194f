#!/bin/bash
# scratch: final lines of the round
cd $GRAFT_REPO_ROOT
O=gpurun_out
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== driver command"
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > $O/C_default.json 2> $O/C_default.err; echo "rc=$?"; tail -c 600 $O/C_default.json
echo "== record"
timeout -k 10 300 python bench.py --steps 3 --warmup 5 --record --no-cpu-baseline --no-full-launch > $O/C_record.json 2> $O/C_record.err; echo "rc=$?"; tail -c 300 $O/C_record.json
echo "== uniform"
timeout -k 10 300 python bench.py --evaluator uniform --steps 3 --warmup 5 --no-cpu-baseline --no-full-launch > $O/C_uniform.json 2> $O/C_uniform.err; echo "rc=$?"; tail -c 300 $O/C_uniform.json
echo "== 2-rank rehearsal"
AR_BENCH_BACKEND=gloo AR_BENCH_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 4 --batch-steps 512 --resident 49152 --no-cpu-baseline --no-full-launch > $O/C_2rank.json 2> $O/C_2rank.err; echo "rc=$?"; tail -c 300 $O/C_2rank.json
echo "== baseline rows"
timeout -k 10 300 python tools/baseline_rows.py > $O/C_baseline_rows.jsonl 2> $O/C_baseline_rows.err; echo "rc=$?"; cut -c1-200 $O/C_baseline_rows.jsonl
