# Round 6: CUs left free by the stream-K launches under the pipeline (harness build); printed margins of the real-motion parity tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6k
rm -rf $O; mkdir -p $O
cd $R
PWC_HARNESS=1 timeout 900 python scripts/exp_pipeline_reserve.py 8 3 2>&1 | grep -v amdgpu.ids > $O/exp_pipeline_reserve.txt
PWC_HARNESS=1 timeout 900 python scripts/exp_pipeline_reserve.py 8 2 2>&1 | grep -v amdgpu.ids >> $O/exp_pipeline_reserve.txt
cat $O/exp_pipeline_reserve.txt
timeout 1800 python -m pytest tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "real_motion or near_the_fp16" 2>&1 | grep -v amdgpu.ids > $O/real_motion_margins.txt
cat $O/real_motion_margins.txt | cut -c1-400
