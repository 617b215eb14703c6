#!/bin/bash
# usage: bash profiles/tools/build_variant.sh NAME "-DFLAG=1 ..."   -> scratch/variants/NAME/libelba_amd.so (select with ELBA_AMD_LIB; scratch/ travels to the GPU box)
set -e
NAME=$1; FLAGS=$2
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/scratch/variants/$NAME
make -s -C $R/elba_amd/csrc -j8 OUTDIR=$R/scratch/variants/$NAME OBJDIR=$R/scratch/variants/$NAME/_obj \
  CXXFLAGS="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$R/include $FLAGS"
ls -la $R/scratch/variants/$NAME/libelba_amd.so
