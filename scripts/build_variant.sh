#!/bin/bash
# Builds lib/variants/libnct_<name>.so: the given csrc files recompiled with extra flags, the other objects taken from build/.
# scripts/pm_modes.py then times every variant beside the default library (NCT_LIB selects the .so in the Python binding).
# usage: scripts/build_variant.sh <name> "<extra hipcc flags>" <csrc file> [...]
set -e
cd "$(dirname "$0")/../neural-color-transfer_amd"
name=$1; flags=$2; shift 2
make -s lib/libnct.so >/dev/null
mkdir -p build/variants/$name lib/variants
objs=""
for o in build/*.o; do
    b=$(basename $o .o); repl=""
    for f in "$@"; do [ "$(basename $f)" = "$b" ] && repl=$f; done
    if [ -n "$repl" ]; then
        x=""; case $repl in *.cpp) x="-x hip";; esac
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result -I../include $flags $x -c csrc/$(basename $repl) -o build/variants/$name/$b.o
        objs="$objs build/variants/$name/$b.o"
    else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libnct_$name.so $objs -lz
echo lib/variants/libnct_$name.so
