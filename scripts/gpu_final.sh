# end of round: refresh the MultiWalker ten-walker profile, then everything scripts/gpu_round.sh does
cd $GRAFT_REPO_ROOT
FILTER=multiwalker_w10 bash scripts/profile_r05.sh 2>&1 | tail -6
bash scripts/gpu_round.sh
