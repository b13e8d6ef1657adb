cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_read -o read -- node napi/bench_frames.js --config c3 --form read --steps 10 > $O/prof_read.log 2>&1
ls -R $O/prof_read | head -30
for f in $(find $O/prof_read -name "*stats*.csv"); do echo "== $f"; head -12 $f; done
