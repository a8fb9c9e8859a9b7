#!/bin/bash
# Development: build libbkcore with the ratio-form translation unit compiled under extra flags.
# usage: tools/build_variant.sh NAME "-DBK_RATIO_T=192 -DBK_RATIO_WAVES=3"   ->  tools/ab/libbkcore_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/project-resnet-bk-an-o-n-language-model-architecture_amd/csrc
mkdir -p $ROOT/tools/ab $CS/build
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed"
/opt/rocm/bin/hipcc $FL -DBK_DEV_F32_ONLY $2 -c -o $CS/build/ratio_$1.o $CS/bkcore_ratio.hip
/opt/rocm/bin/hipcc $FL -shared -o $ROOT/tools/ab/libbkcore_$1.so $CS/build/bkcore_kernels.o $CS/build/bkcore_complex.o $CS/build/ratio_$1.o $CS/build/bkcore_ratio_long.o $CS/build/bkcore_analytic.o $CS/build/bkcore_layer.o $CS/build/bkcore_scan.o $CS/build/bkcore_woodbury.o
/opt/rocm/bin/hipcc $FL -DBK_DEV_F32_ONLY $2 --cuda-device-only -S -o /tmp/ratio_$1.s $CS/bkcore_ratio.hip 2>/dev/null
echo "$1: $(grep -A12 '\.name:.*bk_ratio_forward_kernelIfLb0ELi0ELb0EE' /tmp/ratio_$1.s | grep -E 'vgpr_count|private_seg|group_segment' | tr -s ' ' | tr '\n' ' ')"
