# the round's record runs: bench.py (default + noise 0) and every other BASELINE configuration
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r5}
python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.log
python bench.py --noise 0 --no-cpu-baseline > gpurun_out/${R}_bench_noise0.json 2> gpurun_out/${R}_bench_noise0.log
: > gpurun_out/${R}_configs.jsonl
for c in cfg3 cfg4 cfg5 cfg4t cfg4tp cfg5t cfg2n0 cfg3n0; do
  python3 tools/bench_configs.py $c 10 >> gpurun_out/${R}_configs.jsonl 2> gpurun_out/${R}_cfg_$c.log
done
cut -c1-260 gpurun_out/${R}_configs.jsonl
python -c "
import json
d=json.load(open('gpurun_out/${R}_bench.json')); print(d['value'], d['ms_per_step'], d['frozen_ms_per_step'], d['median_ms_per_step'], d['roofline']['frac'], d['roofline']['equivalent_unfused']['frac'], d['cpu_baseline'])"
