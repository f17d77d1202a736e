# Round-6 evidence run (one gpurun call): GPU tests, smoke, PMC traffic FIRST (the stamped profiles/pmc_traffic.json the bench line
# reads), bench lines, rocprofv3 kernel stats of the default (pipelined) command and of the one-stream loop, the pipelined trace,
# the per-dispatch trace of one forward, the op-level leg.  Outputs under gpurun_out/final6/ (copied into profiles/ by hand).
# Per-kernel passes run the ONE-STREAM loop (--pipeline 0): in the pipelined region the kernels of different forwards overlap and
# a per-kernel duration means nothing.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final6
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids > $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
# ---- counters: separate --pmc passes, kernel trace only
C2="python $R/bench.py --steps 3 --warmup 2 --pipeline 0 --no-cpu-baseline --no-op-timing --no-op-leg"
mkdir -p $O/pmc_req $O/pmc_mfma $O/pmc_op
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_req -o RD --output-format csv -- $C2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_req -o WR --output-format csv -- $C2 > /dev/null 2>&1
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
python $R/scripts/pmc_request_table.py $O/pmc_req $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 -d $O/pmc_mfma -o p1 --output-format csv -- $C2 > /dev/null 2>&1
python $R/scripts/pmc_mfma_table.py $O/pmc_mfma $O/pmc_traffic.json > $O/pmc_mfma_busy.txt 2>&1
C3="python $R/bench.py --op-leg-only"
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_op -o RD --output-format csv -- $C3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_op -o WR --output-format csv -- $C3 > /dev/null 2>&1
python $R/scripts/pmc_op_leg_table.py $O/pmc_op $O/pmc_traffic.json > $O/pmc_traffic_op_leg.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json          # (the bench lines below print roofline.traffic from it)
rm -rf $O/pmc_req/*/ $O/pmc_mfma/*/ $O/pmc_op/*/ 2>/dev/null
# ---- bench lines
cd $R
timeout 600 python bench.py 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
timeout 600 python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_default_120steps.json
timeout 600 python bench.py --pipeline 0 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_pipeline0.json
timeout 600 python bench.py --pipeline 2 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_pipeline2.json
timeout 300 python bench.py --gpus 1 --spawn --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default_spawn.json
timeout 600 python bench.py --config configs3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_configs3.json
timeout 600 python bench.py --config configs4 --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_configs4_per_gpu.json
timeout 300 python bench.py --batch 32 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_b32.json
timeout 300 python bench.py --batch 1 --steps 200 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_b1.json
timeout 300 python bench.py --batch 2 --steps 100 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_b2.json
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_train.json
for b in 8 1; do timeout 600 python scripts/exp_pipeline2.py $b 0 1 2 3 4 2>&1 | grep -v amdgpu.ids >> $O/exp_pipeline.txt; done
# ---- rocprofv3 kernel stats: the default command (pipelined timed region + one-stream loop) and the one-stream command
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-leg --no-fp32-leg > $O/prof_stdout.log 2>&1
python $R/scripts/kernel_stats_table.py $O/prof 44 > $O/kernel_stats_default_command.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench -- python $R/bench.py --steps 10 --warmup 3 --pipeline 0 --no-cpu-baseline --no-op-leg --no-fp32-leg > $O/prof1_stdout.log 2>&1
python $R/scripts/kernel_stats_table.py $O/prof1 44 > $O/kernel_stats_one_stream.txt 2>&1
rm -rf /tmp/ktp /tmp/kt1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktp -o kt -- python $R/bench.py --batch 8 --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o kt -- python $R/bench.py --batch 8 --steps 12 --warmup 3 --pipeline 0 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
python $R/scripts/kernel_trace_pipeline.py /tmp/ktp /tmp/kt1 > $O/pipeline_trace_b8.txt 2>&1
python $R/scripts/kernel_trace_forward.py /tmp/kt1 > $O/forward_trace_b8.txt 2>&1
rm -rf /tmp/ktb1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktb1 -o kt -- python $R/bench.py --batch 1 --steps 4 --warmup 3 --pipeline 0 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
python $R/scripts/kernel_trace_forward.py /tmp/ktb1 > $O/forward_trace_b1.txt 2>&1
rm -rf /tmp/opleg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/opleg -o op -- python $R/bench.py --op-leg-only > /dev/null 2>&1
python $R/scripts/kernel_stats_table.py /tmp/opleg 12 > $O/kernel_stats_op_leg.txt 2>&1
rm -rf $O/prof $O/prof1
find $O -name "*.csv" -size +2M -delete
cat $O/gpu_tests.txt; cat $O/smoke.txt; cut -c1-200 $O/bench_default.json; head -12 $O/kernel_stats_one_stream.txt; tail -5 $O/pmc_traffic.txt
