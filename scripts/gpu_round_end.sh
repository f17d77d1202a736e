# everything the round's committed evidence comes from: GPU tests, default bench, rocprofv3 kernel stats, PMC traffic
set -x
bash scripts/gpu_check.sh > /dev/null 2>&1

rm -rf gpurun_out/prof; bash scripts/gpu_profile.sh > /dev/null 2>&1
python scripts/rocpd_stats.py $(find gpurun_out/prof -name "*.db" | head -1) > gpurun_out/kernel_stats.txt 2>&1
rm -rf gpurun_out/pmc_traffic; bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_table.py gpurun_out/pmc_traffic > gpurun_out/pmc_traffic.txt 2>&1
tail -2 gpurun_out/t3.log
