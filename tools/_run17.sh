cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r17
python bench.py > gpurun_out/r17/bench.json 2> gpurun_out/r17/bench.err
tail -3 gpurun_out/r17/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r17/bench2.json 2>/dev/null
