cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for spec in "pursuit r04_wave pursuit_wave_kernel" "pursuit_c5 r04_c5 pursuit_group_kernel" "pursuit_colocate r04_colocate pursuit_wave_kernel"; do
  set -- $spec
  echo "=== profile $1"; bash scripts/profile_workload.sh $1 $2 $3 2>&1 | tail -4
done
echo "=== profile pursuit, one launch per step"; bash scripts/profile_workload.sh pursuit r04_wave_one_launch pursuit_wave_kernel --streams 1 2>&1 | tail -4
