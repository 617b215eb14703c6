#!/bin/bash
# register / LDS / occupancy figures of every kernel of one HIP source: profiles/tools/regs.sh spgemm.hip [filter]
HERE=$(cd "$(dirname "$0")/../../elba_amd/csrc" && pwd)
cd $HERE && /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I$HERE/../../include -c $1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {n=$5} / VGPRs: / {v=$4} /TotalSGPRs:/ {s=$4} /ScratchSize/ {sc=$5} /Occupancy/ {o=$5} /LDS Size/ {print n, "vgpr=" v, "sgpr=" s, "scratch=" sc, "occ=" o, "lds=" $6}' |
  c++filt | sed 's/elba::(anonymous namespace):://g; s/(elba::(anonymous namespace)::[A-Za-z]*, [a-z ,]*)//' | cut -c1-200 | grep -E "${2:-.}"
