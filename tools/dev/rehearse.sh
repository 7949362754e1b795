#!/bin/bash
# one-GPU rehearsal of the sharded host logic (RCCL communicator of world size 1): tools/dev/rehearse.sh TAG   (on the GPU box)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
for mode in 1 0; do
  for n in 8388608 16777216 33554432; do
    PFX_SHARD_RCCL=$mode BENCH_FORCE_SHARDED=1 timeout -k 10 120 python3 bench.py --particles $n --steps 100 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('library_rccl=$mode', d['config']['particles_per_gpu'], round(d['ms_per_step'],4), round(d['value']/1e9,1), {k:(v['launches'],round(v['avg_ms'],4)) for k,v in d['kernels'].items()})" || exit 1
  done
done > $o/${tag}_shard_rehearsal.txt
BENCH_FORCE_SHARDED=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --particles 8388608 --steps 50 --warmup 10 --no-extras --no-cpu-baseline > /dev/null 2> $o/${tag}_reh.err
find $o/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_shard_rehearsal_kernel_stats.csv
find $o/${tag}_prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $o/${tag}_shard_rehearsal_kernel_trace.csv
rm -rf $o/${tag}_prof
cat $o/${tag}_shard_rehearsal.txt
