# Round 6: forwards dealt to `depth` replicas on streams of their own hardware queues (pwcnet_amd.ForwardPipeline)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6f
rm -rf $O; mkdir -p $O
cd $R
for b in 8 1; do timeout 600 python scripts/exp_pipeline2.py $b 0 1 2 3 4 2>&1 | grep -v amdgpu.ids >> $O/exp_pipeline.txt; done
for b in 8 1; do GPU_MAX_HW_QUEUES=8 timeout 600 python scripts/exp_pipeline2.py $b 0 2 3 4 6 2>&1 | grep -v amdgpu.ids >> $O/exp_pipeline.txt; done
cat $O/exp_pipeline.txt
