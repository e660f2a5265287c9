# round 6: kernel stats + PMC traffic of both MultiWalker workloads after the solver changes -> gpurun_out/profile/r06_multiwalker*
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for spec in "multiwalker_w10 r06_multiwalker_w10 mw_step_kernel" "multiwalker r06_multiwalker mw_step_kernel"; do
  set -- $spec
  echo "=== profile $1"; bash scripts/profile_workload.sh $1 $2 $3 2>&1 | tail -8
done
bash scripts/pmc_mix.sh multiwalker_w10 "mw_step_kernel<1>,mw_step_kernel<2>,mw_step_kernel<4>" 16384 > gpurun_out/pmc_mix_w10.log 2>&1; grep "^==\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST_VALU\|SQ_WAIT_ANY\|SQ_INSTS_VALU \|SQ_BUSY_CYCLES\|GRBM_GUI\|SQ_WAVES\|SQ_WAIT_INST_LDS\|SQ_INSTS_VMEM_RD" gpurun_out/pmc_mix_w10.log | cut -c1-150
