# round 6: the eight-lane class (5 .. 8 walkers, 8 envs per wavefront: 2 048 wavefronts for 16 384 envs) with the solver launch at two wavefronts
# per SIMD like the sixteen-lane class -- 9 / 14 LDS working copies per env (6 / 5 wavefronts per CU by LDS) against the shipped form; parity first
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 9 14; do
MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.c8_2w_$v.so timeout 900 python -m pytest tests/test_multiwalker_gpu.py -x -q -k "bit_for_bit and (5 or 6 or 7 or 8)" 2>&1 | tail -1
done
for w in 8 5; do for rep in 1 2; do for v in 0 9 14; do
if [ $v = 0 ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.c8_2w_$v.so; fi
echo -n "W=$w copies=$v (0 = shipped): "; MW_W=$w MW_WINDOWS=16 timeout 600 python scripts/mw_steady.py --one 2>&1 | tail -1 | cut -c100-260
done; done; done
