#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: another build of the library sources into variants/<name>.so (SNF_LIB_SO=... selects it)
name=$1; shift
mkdir -p variants
cd sniffles_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-result -Wno-unused-value "$@" \
  snf_lib.hip snf_myers.hip snf_combine.hip snf_extract.hip -o ../../variants/$name.so 2>&1 | grep -E "error|undefined" | head
