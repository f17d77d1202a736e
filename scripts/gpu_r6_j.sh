# Round 6: whole GPU suite on the pipeline build; spawn path of the bench; infer_continuous on the pipeline
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6j
rm -rf $O; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 > $O/gpu_tests.txt
cat $O/gpu_tests.txt
timeout 600 python bench.py --gpus 1 --spawn --no-cpu-baseline --no-op-leg 2>$O/spawn_err.txt | tail -1 > $O/bench_spawn.json
python -c "
import json
d=json.load(open('$O/bench_spawn.json')); print('spawn', d['value'], d['ms_per_step'], d.get('value_one_stream'), d['config']['pipeline'], d['config']['parallelism'])"
tail -3 $O/spawn_err.txt
