cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('headline %.4g %s  ms/step %.4f  frac %.3f  one-launch frac %.3f' % (j['value'], j['unit'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['one_launch_per_step']['frac']))
for k,v in j.get('workloads',{}).items():
    print(' ', k, ('%.4g ms/step %.4f frac %.3f one-launch %s cpu %.3g' % (v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline'].get('one_launch_per_step',{}).get('ms_per_step'), v['cpu_baseline']['value'])) if 'value' in v else v)
"
for spec in "pursuit r04_wave pursuit_wave_kernel" "pursuit_c5 r04_c5 pursuit_group_kernel" "waterworld r04_waterworld waterworld_kernel" "waterworld_std r04_waterworld_std obsnorm_pairs_kernel" "hostage r04_hostage hostage_kernel" "multiwalker r04_multiwalker mw_step_kernel" "pursuit_colocate r04_colocate pursuit_wave_kernel"; do
  set -- $spec
  echo "=== profile $1"; bash scripts/profile_workload.sh $1 $2 $3 2>&1 | tail -6
done
echo "=== profile pursuit, one launch per step"; bash scripts/profile_workload.sh pursuit r04_wave_one_launch pursuit_wave_kernel --streams 1 2>&1 | tail -5
