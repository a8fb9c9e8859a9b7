#!/bin/bash
# Development: build libbkcore with BOTH ratio-form translation units (float32 and double chains) compiled under extra
# flags.  usage: tools/build_variant_long.sh NAME "-DBK_LOOK_NOWAIT=1"   ->  tools/ab/libbkcore_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/project-resnet-bk-an-o-n-language-model-architecture_amd/csrc
mkdir -p $ROOT/tools/ab $CS/build
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed"
/opt/rocm/bin/hipcc $FL -DBK_DEV_F32_ONLY $2 -c -o $CS/build/ratio_$1.o $CS/bkcore_ratio.hip &
/opt/rocm/bin/hipcc $FL -DBK_DEV_F32_ONLY $2 -c -o $CS/build/ratio_long_$1.o $CS/bkcore_ratio_long.hip &
wait
/opt/rocm/bin/hipcc $FL -shared -o $ROOT/tools/ab/libbkcore_$1.so $CS/build/bkcore_kernels.o $CS/build/ratio_$1.o $CS/build/ratio_long_$1.o $CS/build/bkcore_analytic.o $CS/build/bkcore_layer.o $CS/build/bkcore_scan.o $CS/build/bkcore_woodbury.o
echo built tools/ab/libbkcore_$1.so
