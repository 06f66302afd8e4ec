#!/bin/bash
# Builds hive_env.hip alone into build/variants/env_<name>.so, one per "name:flags" argument (tools/dev/pair_variants.py times them).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$ROOT/build/variants"
for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -shared $flags \
        -o "$ROOT/build/variants/env_$name.so" "$ROOT/hive-alphazero_amd/csrc/hive_env.hip" 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|" || true
done
