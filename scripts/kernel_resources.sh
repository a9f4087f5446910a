#!/bin/bash
# per-kernel register / LDS / scratch usage of the gfx950 code object (device-only compile, reads the .s metadata)
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d); cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -save-temps $R/open3d_slam_amd/csrc/backend.hip -o /dev/null 2>/dev/null
python3 - "$@" <<'PY'
import re,sys,glob
txt=open(glob.glob('*gfx950*.s')[0]).read()
pat=sys.argv[1] if len(sys.argv)>1 else ''
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', txt, re.S):
    blk=m.group(0)
    g=lambda k: re.search(r'\.%s:\s+(\S+)'%k, blk).group(1)
    name=g('name')
    if pat and pat not in name: continue
    print('%-100s vgpr %3s sgpr %3s lds %6s scratch %5s spill %s' % (name[:100], g('vgpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count')))
PY
rm -rf $T
