cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resident" > gpurun_out/r03_t10.log 2>&1; tail -4 gpurun_out/r03_t10.log
rm -rf gpurun_out/r03_res; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_res -o p -- python3 tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 --warmup 3 --updates 10 --mixer-dtype bf16 > gpurun_out/r03_res.log 2>&1
grep -E "qmix_wide|Name" gpurun_out/r03_res/p_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/r03_resp; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03_resp -o p -- python3 tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 --warmup 1 --updates 3 --mixer-dtype bf16 > gpurun_out/r03_resp.log 2>&1
python3 - <<'PY'
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/r03_resp/p_counter_collection.csv')):
    if 'qmix_wide' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': d[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, 'launches',len(v),'FETCH_SIZE KB mean',sum(v)/len(v),'-> x2 MB', 2*sum(v)/len(v)/1024)
PY
