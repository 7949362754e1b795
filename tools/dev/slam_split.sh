#!/bin/bash
# tools/dev/slam_split.sh TAG  (on the GPU box): config-5 bench + branch split of k_slam_gate_rows + kernel stats
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout 200 python3 bench.py --workload config5 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null > $o/${tag}_config5.json || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_sprof -- python3 tools/slam_branch_split.py run $o/${tag}_census.json > /dev/null 2> $o/${tag}_split.err || exit 1
python3 tools/slam_branch_split.py join $o/${tag}_sprof $o/${tag}_census.json > $o/${tag}_slam_branch_split.json
find $o/${tag}_sprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_slam_kernel_stats.csv
rm -rf $o/${tag}_sprof
python3 -c "
import json
d=json.load(open('$o/${tag}_config5.json')); print('config5', round(d['value']/1e9,1), 'G/s', round(d['ms_per_step'],4), 'ms  frac', round(d['roofline']['frac'],3), 'rows resampled', d['config']['rows_resampled_fraction'])
print(json.load(open('$o/${tag}_slam_branch_split.json'))['fit_us'])"
