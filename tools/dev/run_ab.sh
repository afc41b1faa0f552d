cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for v in 0 1 2 3 4; do
MSCOMP_AMD_XZ=$v python tools/gpu_xzprof.py mozilla dickens nci x-ray 2>&1 | grep -v amdgpu.ids | sed "s/^/XZ=$v /"
MSCOMP_AMD_XZ=$v python tools/gpu_ab.py xpress 1 config5 3 2>&1 | tail -1 | sed "s/^/XZ=$v /"
done
