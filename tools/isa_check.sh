#!/bin/bash
# Dev tool: compile one csrc/*.hip for gfx950 to assembly and print register use, spills, scratch sites and barrier lines.
f=/root/repo/facerecognition_infrenceengine_amd/csrc/$1
out=/tmp/$(basename $1 .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  --cuda-device-only -S $f -o $out -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Spill|Scratch|VGPRs:|Function Name" | grep -v "pack" | head -${2:-5}
echo "lines $(wc -l < $out)"
echo "scratch: $(grep -n -E 'scratch_' $out | awk -F: '{printf "%s ", $1}')"
echo "barriers: $(grep -n 's_barrier' $out | awk -F: '{printf "%s ", $1}' | cut -c1-300)"
