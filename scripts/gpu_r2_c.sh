set -x
mkdir -p gpurun_out
timeout 300 python scripts/exp_timeline.py 8 > gpurun_out/r2c_timeline.txt 2>&1
rm -rf gpurun_out/prof; bash scripts/gpu_profile.sh > /dev/null 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof -name "*.db" | head -1) > gpurun_out/r2c_kernel_stats.txt 2>&1
rm -rf gpurun_out/pmc_traffic; bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_table.py gpurun_out/pmc_traffic gpurun_out/r2c_pmc_traffic.json > gpurun_out/r2c_pmc_traffic.txt 2>&1
bash scripts/gpu_pmc_mfma.sh > /dev/null 2>&1
python scripts/pmc_mfma_table.py gpurun_out/pmc_mfma gpurun_out/r2c_pmc_traffic.json > gpurun_out/r2c_pmc_mfma.txt 2>&1
tail -3 gpurun_out/r2c_pmc_mfma.txt
