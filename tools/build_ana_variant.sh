#!/bin/bash
# Development: build libbkcore with the analytic-backward translation unit compiled under extra flags.
# usage: tools/build_ana_variant.sh NAME "-DBK_SWEEP_WG=2"   ->  tools/ab/libbkcore_ana_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/project-resnet-bk-an-o-n-language-model-architecture_amd/csrc
mkdir -p $ROOT/tools/ab $CS/build
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed"
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize $2 -c -o $CS/build/ana_$1.o $CS/bkcore_analytic.hip
/opt/rocm/bin/hipcc $FL -shared -o $ROOT/tools/ab/libbkcore_ana_$1.so $CS/build/bkcore_kernels.o $CS/build/bkcore_ratio.o $CS/build/bkcore_ratio_long.o $CS/build/ana_$1.o $CS/build/bkcore_layer.o $CS/build/bkcore_scan.o $CS/build/bkcore_woodbury.o
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize $2 --cuda-device-only -S -o /tmp/ana_$1.s $CS/bkcore_analytic.hip 2>/dev/null
echo "$1: $(grep -A12 '\.name:.*bk_adjoint_sweepsILi1' /tmp/ana_$1.s | grep -E 'vgpr_count|private_seg|group_segment' | tr -s ' ' | tr '\n' ' ')"
