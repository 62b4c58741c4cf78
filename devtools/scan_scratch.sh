#!/bin/bash
# lists every kernel of csrc/*.hip that the compiler gave a private segment (scratch memory): devtools/scan_scratch.sh
# (a private segment costs every launch of the kernel; a conditional between members of two local structs is the usual cause -- gemm_split16.hip)
cd "$(dirname "$0")/../kaldi-aslp_amd"
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Innet -Iutil -Iparallel --cuda-device-only -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk -v f=$f '/Function Name:/{name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name)} /ScratchSize \[bytes\/lane\]:/{n=$0; sub(/.*: /,"",n); sub(/ .*/,"",n); if (n+0>0) print f, n, name}'
done
