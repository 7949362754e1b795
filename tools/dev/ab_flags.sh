#!/bin/bash
# usage: ab_flags.sh NAME1:"-Dflags" NAME2:"-Dflags" ...  -- same-box A/B of builds of the WORKING TREE that differ in -D flags
# only (NAME:"" = the default build).  Builds lib<NAME>.so for each, prints the register report of k_loc_deposit<true>, runs
# bench.py on all of them alternately (two rounds) in ONE gpurun call and prints the per-kernel averages.
# BENCH_ARGS (env) is appended to the bench command (e.g. --always-resample).
set -e
cd /root/repo/slam3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-unused-value -Wno-unused-function"
names=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  names+=("$name")
  /opt/rocm/bin/hipcc $F $flags -shared -o ../lib/libexp_$name.so pf_loc.hip pf_loc_batch.hip pf_slam.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/ab_$name.usage
  echo "$name [$flags]:" $(grep -A10 "k_loc_depositILb1" /tmp/ab_$name.usage | grep -E " VGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size" | head -5 | sed 's/.*remark: *//;s/\[-R.*//' | tr '\n' ' ')
done
cd /root/repo
list="${names[*]}"
/usr/local/graft/bin/gpurun --timeout 900 -- "for r in 1 2; do for v in $list; do PFX_LIB_PATH=\$PWD/slam3d_amd/lib/libexp_\$v.so timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline $BENCH_ARGS > gpurun_out/abf_\${v}_\$r.json 2> gpurun_out/abf_\${v}_\$r.err || exit 1; done; done" > /tmp/abf_gpurun.log 2>&1 || { tail -8 /tmp/abf_gpurun.log; exit 1; }
grep -E "status=" /tmp/abf_gpurun.log || true
for n in "${names[@]}"; do rm -f slam3d_amd/lib/libexp_$n.so; done
python - "$@" <<'PY'
import json, sys
for spec in sys.argv[1:]:
    v = spec.split(":")[0]
    for r in (1, 2):
        d = json.loads([l for l in open(f"gpurun_out/abf_{v}_{r}.json") if l.startswith("{")][-1])
        print(f"{v:14s} run {r}: {d['ms_per_step']:.4f} ms/step  {d['value']/1e9:6.1f} G/s ",
              {k: round(x["avg_ms"], 4) for k, x in d.get("kernels", {}).items()} or d.get("roofline", {}).get("frac"))
PY
