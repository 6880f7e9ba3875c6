#!/bin/bash
# A/B builds of the hot-path library: tools/build_variant.sh <name> "<extra hipcc flags>" [files...]
# compiles the listed kernel files (default: agent.hip rollout_fused.hip) with the extra flags and links them with the
# baseline objects of the other files into marl_amd/variants/libmarl_hip_<name>.so  (select with MARL_HIP_LIB=<path>).
set -e
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-agent.hip rollout_fused.hip}
cd "$(dirname "$0")/../marl_amd/csrc"
make -j8 > /dev/null
mkdir -p build/v_$NAME ../variants
OBJS=""
for f in gemm agent mixers optim rollout rollout_fused rollout_x6 rollout_x6_v1 qmix_fused mlp3_fused qtran_fused qmix_wide mlp3_x6 agent_x6 agent_x6p agent_bwd_x6; do
  if echo " $FILES " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC $FLAGS -c -o build/v_$NAME/$f.o $f.hip &
    OBJS="$OBJS build/v_$NAME/$f.o"
  else
    OBJS="$OBJS build/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libmarl_hip_$NAME.so $OBJS
echo built marl_amd/variants/libmarl_hip_$NAME.so
