#!/bin/bash
# Register / scratch / LDS use of every kernel in an object built for gfx950 (the code object's metadata notes):
#   scripts/kernel_resources.sh pointnet2_amd/csrc/build/train_mlp.o [name filter]
OBJ=$1; FLT=${2:-.}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "$OBJ"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/dev.co" | python3 -c "
import re, sys, subprocess
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = name.replace('pn2::', '').replace('void ', '')
    name = name[:name.find('(')] if '(' in name else name
    print('%-70s vgpr %4s sgpr %4s scratch %5s lds %6s' % (name[:70], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
" | grep -E "$FLT" | sort
rm -rf "$TMP"
