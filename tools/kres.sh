#!/bin/bash
# kernel resource usage of one csrc/*.hip file: name, VGPRs, AGPRs, spills, scratch, occupancy, LDS
# usage: tools/kres.sh photo.hip [extra hipcc flags]
cd "$(dirname "$0")/../self-supervised-depth-estimation_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kres_$$.o 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /VGPRs:/ && !/Spill/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
       /ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /SGPRs Spill/ {ss=$0; sub(/.*: /,"",ss); sub(/ \[.*/,"",ss)}
       /VGPRs Spill/ {vs=$0; sub(/.*: /,"",vs); sub(/ \[.*/,"",vs)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-70s vgpr %-4s agpr %-4s vspill %-3s sspill %-3s scratch %-4s occ %-2s lds %s\n", name, v, a, vs, ss, sc, o, l}'
rm -f /tmp/kres_$$.o
