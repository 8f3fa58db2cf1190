#!/bin/bash
# dev: registers / spills of the k_stream_spec instantiations of one translation unit under extra flags:  spill_probe.sh spec_lds.hip -DFOO
cd "$(dirname "$0")/../../rustlight_amd/csrc/kernels"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-bitwise-instead-of-logical "$@" -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/probe_$$.o 2>&1 \
  | grep -E "remark: Function Name|remark:     VGPRs:|ScratchSize|SGPRs Spill|VGPRs Spill" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - \
  | while read a b n c v d e sc f g ss h i vs; do echo "$(c++filt $n | sed -E 's/void rl:://; s/\(.*//') vgpr $v scratch $sc sgpr-spill $ss vgpr-spill $vs"; done
rm -f /tmp/probe_$$.o
