# Round 6: kernel trace of the pipelined bench beside the one-stream one
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6h
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktp /tmp/kt1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktp -o kt -- python $R/bench.py --batch 8 --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > $O/p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o kt -- python $R/bench.py --batch 8 --steps 12 --warmup 3 --pipeline 0 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > $O/p1.log 2>&1
python $R/scripts/kernel_trace_pipeline.py /tmp/ktp /tmp/kt1 > $O/pipeline_trace_b8.txt 2>&1
head -60 $O/pipeline_trace_b8.txt
