#!/bin/bash
# like ab_flags.sh but runs tools/dev/slam_split.sh (config-5 bench + branch split) per variant
set -e
cd /root/repo/slam3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-unused-value -Wno-unused-function"
names=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  names+=("$name")
  /opt/rocm/bin/hipcc $F $flags -shared -o ../lib/libexp_$name.so pf_loc.hip pf_loc_batch.hip pf_slam.hip 2> /tmp/ab_$name.usage
done
cd /root/repo
list="${names[*]}"
/usr/local/graft/bin/gpurun --timeout 900 -- "for r in 1 2; do for v in $list; do export PFX_LIB_PATH=\$PWD/slam3d_amd/lib/libexp_\$v.so; echo \$v; bash tools/dev/slam_split.sh abs_\${v}_\$r || exit 1; done; done" 2>&1 | grep -E "^[a-z_0-9]+$|config5|only_norm|status"
for n in "${names[@]}"; do rm -f slam3d_amd/lib/libexp_$n.so; done
