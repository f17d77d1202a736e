# Round 6, third GPU call: the reworked tests, the LDS-resident-weights conv in / out of the forward (A/B), forward kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6c
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "real_motion or near_the_fp16 or lazy_range or fallback or out_of_range" > $O/tests_model.txt 2>&1
tail -4 $O/tests_model.txt; grep -E "real motion|near the fp16" $O/tests_model.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "lds_resident or resize or final_flow" > $O/tests_ops.txt 2>&1
tail -2 $O/tests_ops.txt
for b in 8 1; do timeout 300 python scripts/exp_ab_model.py w32_conv $b 2>&1 | grep -v amdgpu.ids >> $O/exp_ab_w32.txt; done
cat $O/exp_ab_w32.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt8
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt8 -o kt -- python $R/bench.py --batch 8 --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
python $R/scripts/kernel_trace_forward.py /tmp/kt8 > $O/forward_trace_b8.txt 2>&1
cut -c1-110 $O/forward_trace_b8.txt | head -70
