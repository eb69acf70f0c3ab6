# Build a variant of the library with extra compile flags for ONE translation unit (A/B runs inside one GPU session via MAS_LIB):
#   bash tools/build_variant.sh conv_bx "-DBX_ADMA=0" libvar_noadma.so       -> build/variants/libvar_noadma.so  (build/ is git-ignored and travels to the GPU box; _lib.load() refuses a MAS_LIB inside the package)
set -eu
unit=$1; flags=$2; out=$3
cd "$(dirname "$0")/../mulactseg_amd/csrc"
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $flags -c $unit.hip -o /tmp/variant_$unit.o
objs=$(ls *.o | grep -v "^$unit.o$" | grep -v "^test_support.o$" | tr '\n' ' ')
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$out $objs /tmp/variant_$unit.o
echo built build/variants/$out
