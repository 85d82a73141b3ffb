#!/bin/bash
# Compact register / spill report of one HIP source:  tools/kres.sh <file.hip> [name-filter] [extra hipcc flags]
f=$1; pat=${2:-.}; shift; shift
cd "$(dirname "$f")" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$(basename "$f")" -o /tmp/kres.o 2>&1 |
  awk '/Function Name:/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-Rpass.*/,"",n)} / VGPRs:/{v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)} /AGPRs:/{a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)} /ScratchSize/{s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)} /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)} /LDS Size/{print substr(n,1,110), "vgpr", v, "agpr", a, "scratch", s, "occ", o}' | grep -E "$pat"
