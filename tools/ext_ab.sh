#!/bin/bash
# A/B of K3_EXT_LIMIT builds (build_variants/libm4g_ext<N>.so) on the inputs of tools/kind_probe.py and on text.
out=gpurun_out/ext_ab; mkdir -p $out
for v in "$@"; do
  if [ $v = base ]; then unset M4G_LIB; else export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_ext$v.so; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/kind_probe.py 2>/dev/null | grep -E "random|paragraphs|second" || exit 1
  KP_CHUNKS=1024 timeout -k 10 200 python3 tools/kind_probe.py paragraphs 2>/dev/null | sed 's/^/  1024 chunks: /' || exit 1
done
