cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2/pytest.log
tail -8 gpurun_out/r2/pytest.log
