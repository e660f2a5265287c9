# round 6: kernel stats + PMC traffic of every workload of the bench line -> gpurun_out/profile/<name>/ (copy to profiles/)
#   scripts/profile_r06.sh [workload-name-filter]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for spec in "pursuit r06_wave pursuit_wave_kernel" "multiwalker r06_multiwalker mw_step_kernel" "multiwalker_w10 r06_multiwalker_w10 mw_step_kernel" \
            "pursuit_rollout r06_rollout pursuit_" "waterworld r06_waterworld waterworld_kernel" "pursuit_c5 r06_c5 pursuit_group_kernel" \
            "pursuit_colocate r06_colocate pursuit_wave_kernel" "waterworld_std r06_waterworld_std obsnorm_pairs" "hostage r06_hostage hostage_kernel" \
            "pursuit_authors r06_authors pursuit_group_kernel"; do
  set -- $spec
  if [ -n "$FILTER" ] && [ "$1" != "$FILTER" ]; then continue; fi
  echo "=== profile $1"; bash scripts/profile_workload.sh $1 $2 $3 2>&1 | tail -5
done
if [ -z "$FILTER" ]; then
  echo "=== profile pursuit, one launch per step"; bash scripts/profile_workload.sh pursuit r06_wave_one_launch pursuit_wave_kernel --streams 1 2>&1 | tail -4
  echo "=== profile waterworld, one launch per step"; bash scripts/profile_workload.sh waterworld r06_waterworld_one_launch waterworld_kernel --streams 1 2>&1 | tail -4
fi
