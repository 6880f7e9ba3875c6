cd $GRAFT_REPO_ROOT
timeout 120 tools/probe/phase_probe > gpurun_out/r03_phase_probe.txt 2>&1
cat gpurun_out/r03_phase_probe.txt
for v in base noslp pad8 pad8noslp; do
  if [ $v = base ]; then L=marl_amd/libmarl_hip.so; else L=marl_amd/variants/libmarl_hip_$v.so; fi
  MARL_HIP_LIB=$PWD/$L timeout 200 python tools/ktime.py --tag $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03_ab1.txt 2>&1
cat gpurun_out/r03_ab1.txt
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r03_b2.log 2>&1; tail -c 600 gpurun_out/r03_b2.log
timeout 500 python tools/cpu_threads.py > gpurun_out/r03_cpu_threads.txt 2>&1; cat gpurun_out/r03_cpu_threads.txt
