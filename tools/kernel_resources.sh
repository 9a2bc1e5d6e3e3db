#!/bin/bash
# tools/kernel_resources.sh [REGEX]: registers, scratch, LDS and occupancy of the kernels in ss_kernels.hip (hipcc's kernel-resource-usage remarks)
cd "$(dirname "$0")/../splashsurf_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC "${@:2}" -Rpass-analysis=kernel-resource-usage -c ss_kernels.hip -o /tmp/ss_kernels_res.o 2> /tmp/ss_kernels_res.txt
python3 - "${1:-k_splat_fused|k_splat_accumulate_list|k_density_sub|k_splat_certify}" <<'PY'
import re, sys, subprocess
t = open('/tmp/ss_kernels_res.txt').read()
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    except Exception:
        pass
    if not re.search(sys.argv[1], name):
        continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    print("%-60s VGPR %3s SGPR %3s scratch %3s occupancy %s LDS %s" % (name[:60], g('VGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
PY
