# Compile one translation unit of csrc/ with temporaries and print registers / spills per kernel:  bash tools/isa.sh conv_bx [extra flags]
# the ISA lands in /tmp/isa/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s
set -eu
unit=$1; shift
mkdir -p /tmp/isa && rm -f /tmp/isa/$unit-*
src="$(cd "$(dirname "$0")/../mulactseg_amd/csrc" && pwd)"
cd /tmp/isa          # (every temporary of -save-temps lands here, none beside the sources)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I"$src" "$@" -c "$src/$unit.hip" -o /tmp/isa/$unit.o -save-temps=obj
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|name):" /tmp/isa/$unit-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | sed 's/_ZN12_GLOBAL__N_1//' | awk '{print $2, "vgpr", $4, "spill", $6}'
