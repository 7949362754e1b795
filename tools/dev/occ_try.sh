#!/bin/bash
# usage: occ_try.sh KERNEL_MANGLED_SUBSTR WAVES [extra -D flags]: compile pf_loc.hip with amdgpu_waves_per_eu(WAVES) on k_loc_deposit
# and print the register / spill report of the kernel
cd /root/repo/slam3d_amd/csrc
k=$1; w=$2; shift 2
sed "s/__global__ __launch_bounds__(kThreads) void k_loc_deposit(LocBufs bufs/__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu($w,$w))) void k_loc_deposit(LocBufs bufs/" pf_loc_kernels.h > /tmp/pfk_test.h
sed -i 's#include "pf_common.h"#include "'$PWD'/pf_common.h"#' /tmp/pfk_test.h
sed 's#"pf_loc_kernels.h"#"/tmp/pfk_test.h"#; s#"pf_host.h"#"'$PWD'/pf_host.h"#; s#"../../include/pfx.h"#"/root/repo/include/pfx.h"#' pf_loc.hip > /tmp/pf_loc_test.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -S --cuda-device-only -o /tmp/pf_w.s /tmp/pf_loc_test.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/pf_w.usage
echo "waves $w $@:" $(grep -A10 "$k" /tmp/pf_w.usage | grep -E " VGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: *//;s/\[-R.*//' | tr '\n' ' ')
