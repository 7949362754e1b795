#!/bin/bash
# measurement set of round 3: tools/dev/r03_measure.sh TAG [loc|slam|extra]...   (run on the GPU box from the repo root)
# loc:   default bench line, short run, rocprofv3 kernel stats of the bench, PMC traffic (two passes), SQ counters
# slam:  config-5 bench, branch split of the row kernel, kernel stats, PMC traffic of the row kernel
# extra: batched Loc benches, one-GPU rehearsal of the sharded filter
set -e
tag=$1; shift
what=${@:-loc slam extra}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
for w in $what; do
case $w in
loc)
timeout -k 10 500 python3 bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/${tag}_bench_short.json 2>> $o/${tag}_bench_default.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $o/${tag}_bench_under_rocprof.json 2> $o/${tag}_prof.err
find $o/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_rocprofv3_kernel_stats.csv
rm -rf $o/${tag}_prof
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/${tag}_pmc_fetch -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2> $o/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/${tag}_pmc_write -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
python3 tools/pmc_traffic.py $o/${tag}_pmc_fetch $o/${tag}_pmc_write 16000000 > $o/${tag}_pmc_traffic.json
cp profiles/pmc_traffic.json $o/${tag}_pmc_traffic_latest.json
rm -rf $o/${tag}_pmc_fetch $o/${tag}_pmc_write
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $o/${tag}_sq1 -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $o/${tag}_sq2 -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
{ python3 tools/sq_counters.py $o/${tag}_sq1 "k_loc_deposit<true>" "SQ_INSTS_VALU>75000000"; python3 tools/sq_counters.py $o/${tag}_sq2 "k_loc_deposit<true>" "SQ_ACTIVE_INST_ANY>110000000"; } > $o/${tag}_sq_counters.txt
rm -rf $o/${tag}_sq1 $o/${tag}_sq2
;;
slam)
bash tools/dev/slam_split.sh $tag
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/${tag}_spmc_fetch -- python3 bench.py --workload config5 --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2> $o/${tag}_spmc.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/${tag}_spmc_write -- python3 bench.py --workload config5 --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2>> $o/${tag}_spmc.err
python3 tools/pmc_traffic.py $o/${tag}_spmc_fetch $o/${tag}_spmc_write slam > $o/${tag}_pmc_traffic_slam.json
cp profiles/pmc_traffic_slam.json $o/${tag}_pmc_traffic_slam_latest.json
rm -rf $o/${tag}_spmc_fetch $o/${tag}_spmc_write
# the bench line again, now that it finds fresh PMC traffic
timeout 200 python3 bench.py --workload config5 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null > $o/${tag}_config5.json
;;
extra)
timeout -k 10 200 python3 bench.py --workload loc_batch --batch-filters 1600 --particles 10000 --no-cpu-baseline > $o/${tag}_loc_batch_1600x10000.json 2> $o/${tag}_extra.err
timeout -k 10 200 python3 bench.py --workload loc_batch --batch-filters 16 --particles 1048576 --no-cpu-baseline > $o/${tag}_loc_batch_16x1M.json 2>> $o/${tag}_extra.err
bash tools/dev/rehearse.sh $tag
;;
esac
done
echo done: $what
