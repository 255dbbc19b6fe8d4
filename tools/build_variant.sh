#!/bin/bash
# build_variants/libm4g_<name>.so from the working tree with extra compiler flags; prints k3a's registers
#   tools/build_variant.sh name [-DFOO=1 ...]
name=$1; shift
cd "$(dirname "$0")/../brotli_amd/csrc" || exit 1
mkdir -p /tmp/isa/$name && cd /tmp/isa/$name || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared --save-temps "$@" \
  -o /root/repo/build_variants/libm4g_$name.so /root/repo/brotli_amd/csrc/m4g.hip 2>&1 | grep -E "error|Error" 
grep -E "rare_iteration.*\.(num_vgpr|private_seg_size), [0-9]+$" m4g-hip-amdgcn-amd-amdhsa-gfx950.s | sed "s/.*\.\(num_vgpr\|private_seg_size\), /rare \1 /" | tr "\n" " "; echo
for k in k3a_fast k3a_specILb0 k3a_specILb1; do
  awk "/^_ZN3m4g8${k}/,/\.end_amdhsa_kernel/" m4g-hip-amdgcn-amd-amdhsa-gfx950.s | grep -E "^; (NumVgprs|ScratchSize|Occupancy|LDSByteSize)|next_free_vgpr|private_segment_fixed" | tr '\n' ' '; echo " <- $k"
done
