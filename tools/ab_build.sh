#!/bin/bash
# Build a variant of libvunet_hip.so for A/B timing of kernel changes without touching the shipped library:
#     tools/ab_build.sh <tag> <source.hip> [extra hipcc flags ...]
# recompiles ONE source with the extra flags (e.g. -DH2_NO_TAP_BARRIER) and links it with the shipped objects of all the
# others into behavior_driven_video_synthesis_amd/build/libvunet_hip_<tag>.so.  Use:  VUNET_HIP_LIB=<that path> python tools/time_conv.py
set -eu
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
PKG=behavior_driven_video_synthesis_amd
python -c "import __graft_entry__ as g; g.build()" > /dev/null
OBJ=$PKG/build/ab_${TAG}_$(basename $SRC).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $PKG/csrc/$SRC -o $OBJ
OTHERS=$(ls $PKG/build/*.hip.o | grep -v "/ab_" | grep -v "/$(basename $SRC).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/build/libvunet_hip_${TAG}.so $OBJ $OTHERS
echo $PKG/build/libvunet_hip_${TAG}.so
