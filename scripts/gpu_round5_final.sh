# Round-5 evidence run (one gpurun call): GPU tests, bench lines, rocprofv3 kernel stats, PMC traffic (L2 request-size
# counters) + MFMA busy, per-launch timeline, training step, the round's microbenchmarks.  Outputs under gpurun_out/final5/
# (copied into profiles/ by hand).  Per-kernel passes run the SINGLE-STREAM forward (--streams 1): with the default two
# sub-batches the kernels of different sub-batches overlap and a per-kernel duration means nothing.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final5
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
timeout 600 python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_default_120steps.json
timeout 600 python bench.py --streams 2 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_streams2.json
timeout 300 python bench.py --gpus 1 --spawn --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default_spawn.json
timeout 600 python bench.py --config configs3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_configs3.json
timeout 600 python bench.py --config configs4 --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_configs4_per_gpu.json
timeout 300 python bench.py --batch 32 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_b32.json
timeout 300 python bench.py --batch 1 --steps 200 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_b1.json
timeout 300 python bench.py --batch 2 --steps 100 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 > $O/bench_b2.json
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_train.json
timeout 300 python scripts/exp_timeline.py 8 > $O/timeline_batch8.txt 2>/dev/null
timeout 200 python scripts/exp_host.py 1 > $O/exp_host_issue_vs_graph_b1.txt 2>&1
for b in 8 1; do PYTHONPATH=. timeout 300 python scripts/exp_blk_ab.py $b 2>&1 | grep -v amdgpu.ids >> $O/exp_blk_ab.txt; done
for b in 8 1; do PYTHONPATH=. timeout 300 python scripts/exp_sk_ab.py $b 2>&1 | grep -v amdgpu.ids >> $O/exp_sk_ab.txt; done
PYTHONPATH=. timeout 300 python scripts/exp_t32_ab.py 8 2>&1 | grep -v amdgpu.ids > $O/exp_t32_ab.txt
timeout 300 python scripts/exp_ab_model.py thin_conv 8 2>&1 | grep -v amdgpu.ids > $O/exp_ab_thin_conv.txt
for b in 8 2 1; do timeout 300 python scripts/exp_ab_model.py small_conv $b 2>&1 | grep -v amdgpu.ids >> $O/exp_ab_small_conv.txt; done
timeout 600 python scripts/exp_ab_model.py f16x2 8 2>&1 | grep -v amdgpu.ids > $O/exp_ab_f16x2.txt
timeout 600 python scripts/exp_ab_model.py f16x2_stream_k 8 2>&1 | grep -v amdgpu.ids > $O/exp_ab_stream_k.txt
cd /tmp && export TMPDIR=/tmp
C="python $R/bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-op-leg --no-fp32-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- $C > $O/prof_stdout.log 2>&1
python $R/scripts/kernel_stats_table.py $O/prof 44 > $O/kernel_stats.txt 2>&1
for b in 8 1 2; do
  rm -rf /tmp/kt$b
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$b -o kt -- python $R/bench.py --batch $b --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
  python $R/scripts/kernel_trace_forward.py /tmp/kt$b > $O/forward_trace_b$b.txt 2>&1
done
rm -rf /tmp/opleg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/opleg -o op -- python $R/bench.py --op-leg-only > /dev/null 2>&1
python $R/scripts/kernel_stats_table.py /tmp/opleg 12 > $O/kernel_stats_op_leg.txt 2>&1
C2="python $R/bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg"
mkdir -p $O/pmc_req $O/pmc_mfma
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_req -o RD --output-format csv -- $C2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_req -o WR --output-format csv -- $C2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_req -o FS --output-format csv -- $C2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_req -o WS --output-format csv -- $C2 > /dev/null 2>&1
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
python $R/scripts/pmc_request_table.py $O/pmc_req $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 -d $O/pmc_mfma -o p1 --output-format csv -- $C2 > /dev/null 2>&1
python $R/scripts/pmc_mfma_table.py $O/pmc_mfma $O/pmc_traffic.json > $O/pmc_mfma_busy.txt 2>&1
# op-level leg of the correlation + warp launches under the request-size counters (roofline_hbm.traffic)
C3="python $R/bench.py --op-leg-only"
mkdir -p $O/pmc_op
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_op -o RD --output-format csv -- $C3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_op -o WR --output-format csv -- $C3 > /dev/null 2>&1
python $R/scripts/pmc_op_leg_table.py $O/pmc_op $O/pmc_traffic.json > $O/pmc_traffic_op_leg.txt 2>&1
rm -rf $O/pmc_op/*/ 2>/dev/null
rm -rf /tmp/tp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -o tp -- python $R/scripts/exp_train_profile.py > /dev/null 2>&1
python $R/scripts/kernel_stats_table.py /tmp/tp 30 > $O/train_kernel_stats.txt 2>&1
rm -rf $O/prof $O/pmc_req/*/ $O/pmc_mfma/*/ 2>/dev/null
find $O -name "*.csv" -size +2M -delete
cat $O/gpu_tests.txt; cut -c1-160 $O/bench_default.json; head -12 $O/kernel_stats.txt
