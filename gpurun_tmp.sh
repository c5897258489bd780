cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rop.py -x -q -m gpu 2>&1 | tail -3
python tools/phase_profile.py 1526 2>&1 | grep -v amdgpu.ids
python tools/phase_profile.py 16 2>&1 | grep -v amdgpu.ids
