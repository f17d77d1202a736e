# copies the summaries of gpurun_out/final5 (scripts/gpu_round5_final.sh) into profiles/ under their round-5 names
O=gpurun_out/final5
for f in bench_default bench_default_120steps bench_streams2 bench_default_spawn bench_configs3 bench_configs4_per_gpu bench_b32 bench_b1 bench_b2 bench_train; do cp $O/$f.json profiles/r05_$f.json; done
cp $O/kernel_stats.txt profiles/r05_kernel_stats_final.txt
cp $O/kernel_stats_op_leg.txt profiles/r05_kernel_stats_op_leg.txt
cp $O/pmc_traffic.txt profiles/r05_pmc_traffic_final.txt
cp $O/pmc_traffic_op_leg.txt profiles/r05_pmc_traffic_op_leg.txt
cp $O/pmc_mfma_busy.txt profiles/r05_pmc_mfma_busy.txt
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cp $O/timeline_batch8.txt profiles/r05_timeline_batch8.txt
for b in 8 1 2; do cp $O/forward_trace_b$b.txt profiles/r05_forward_trace_b$b.txt; done
for f in exp_ab_f16x2 exp_ab_stream_k exp_ab_small_conv exp_ab_thin_conv exp_host_issue_vs_graph_b1; do cp $O/$f.txt profiles/r05_$f.txt; done
cp $O/train_kernel_stats.txt profiles/r05_train_kernel_stats.txt
cp $O/gpu_tests.txt profiles/r05_gpu_tests.txt
