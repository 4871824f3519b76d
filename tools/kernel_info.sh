#!/bin/bash
# Registers, LDS, scratch and code size of the kernels of a built library.  usage: tools/kernel_info.sh [lib.so] [name regex]
L=${1:-feature_extraction_amd/lib/libfx_hip.so}; P=${2:-.}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$L" $T/fat.bin
$B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.o
$B/llvm-readelf -s $T/co.o | awk '$4=="FUNC"{print $8, $3}' > $T/sizes.txt
$B/llvm-readelf --notes $T/co.o | python3 -c "
import sys, re
sizes = dict(l.split() for l in open('$T/sizes.txt'))
txt = sys.stdin.read()
for blk in re.split(r'\n  - \.agpr_count', txt)[1:]:
    nm = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if not re.search(r'$P', nm): continue
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)
    print(f\"{nm:28s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s} spills {g('vgpr_spill_count'):>3s} code {sizes.get(nm, '?'):>7s} B\")
"
rm -rf $T
