#!/bin/bash
# measurement set at the end of round 2: tools/dev/r02k_measure.sh TAG   (run on the GPU box from the repo root)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 400 python3 bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/${tag}_bench_short.json 2>> $o/${tag}_bench_default.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $o/${tag}_bench_under_rocprof.json 2> $o/${tag}_prof.err
find $o/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_rocprofv3_kernel_stats.csv
rm -rf $o/${tag}_prof
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/${tag}_pmc_fetch -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2> $o/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/${tag}_pmc_write -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
python3 tools/pmc_traffic.py $o/${tag}_pmc_fetch $o/${tag}_pmc_write 16000000 > $o/${tag}_pmc_traffic.json
rm -rf $o/${tag}_pmc_fetch $o/${tag}_pmc_write
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $o/${tag}_sq1 -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $o/${tag}_sq2 -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
{ python3 tools/sq_counters.py $o/${tag}_sq1 "k_loc_deposit<true>" "SQ_INSTS_VALU>75000000"; python3 tools/sq_counters.py $o/${tag}_sq2 "k_loc_deposit<true>" "SQ_ACTIVE_INST_ANY>110000000"; } > $o/${tag}_sq_counters.txt
rm -rf $o/${tag}_sq1 $o/${tag}_sq2
