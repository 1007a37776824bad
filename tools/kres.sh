#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra flags]: per-kernel VGPR / SGPR / scratch / LDS / occupancy as the compiler reports them
f=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I/root/repo/include -I/root/repo/frequency-aware-inverse-consistent-octa-super-resolution_amd/csrc "$@" -c /root/repo/frequency-aware-inverse-consistent-octa-super-resolution_amd/csrc/$f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ *\[-Rpass.*//' | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' | c++filt | sed -E 's/\(float const.*\)//' | cut -c1-300
