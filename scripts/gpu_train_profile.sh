# rocprofv3 kernel stats of 4 training steps (batch 8, 448x1024) -> gpurun_out/r2_train_kernel_stats2.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -o tp -- python $R/scripts/exp_train_profile.py > /tmp/tp.log 2>&1
tail -3 /tmp/tp.log
python $R/scripts/kernel_stats_table.py /tmp/tp 26 | tee $R/gpurun_out/r2_train_kernel_stats2.txt
