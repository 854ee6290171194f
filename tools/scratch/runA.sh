#!/bin/bash
# scratch: profiles on the current build: kernel trace + timeline, HBM traffic (two PMC passes), SQ counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-full-launch > $O/A_trace.json 2> $O/A_trace.err && python3 $R/tools/kernel_stats.py /tmp/kt > $O/r03_bench_kernel_stats.csv && python3 $R/tools/timeline.py /tmp/kt 16 > $O/r03_timeline_131k.txt; head -14 $O/r03_bench_kernel_stats.csv; rm -rf /tmp/kt
echo "== pmc fetch/write"
P="python3 $R/bench.py --steps 1 --warmup 8 --batch-steps 512 --no-cpu-baseline --no-full-launch"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f --output-format csv -- $P > $O/A_pf.json 2> $O/A_pf.err && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w --output-format csv -- $P > $O/A_pw.json 2> $O/A_pw.err && python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw --evaluator mlp --resident 131072 --source "profiles/r03_pmc_hbm_traffic.txt" --out $O/traffic.json > $O/r03_pmc_hbm_traffic.txt; cat $O/r03_pmc_hbm_traffic.txt; rm -rf /tmp/pf /tmp/pw
echo "== pmc sq mlp"
C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $C -d /tmp/pm_mlp -o m --output-format csv -- python3 $R/bench.py --warmup-batch-steps 2500 --batch-steps 32 --steps 1 --no-cpu-baseline --no-full-launch > $O/A_pm_mlp.json 2> $O/A_pm_mlp.err && python3 $R/tools/pmc_summary.py /tmp/pm_mlp > $O/r03_pmc_sq_mlp_pipeline.txt; head -8 $O/r03_pmc_sq_mlp_pipeline.txt; rm -rf /tmp/pm_mlp
echo "== pmc sq sym"
timeout -k 10 300 rocprofv3 --pmc $C -d /tmp/pm_sym -o s --output-format csv -- python3 $R/bench.py --evaluator symmetric --warmup-batch-steps 400 --batch-steps 16 --steps 1 --no-cpu-baseline --no-full-launch > $O/A_pm_sym.json 2> $O/A_pm_sym.err && python3 $R/tools/pmc_summary.py /tmp/pm_sym > $O/r03_pmc_sq_symmetric.txt; head -8 $O/r03_pmc_sq_symmetric.txt; rm -rf /tmp/pm_sym
