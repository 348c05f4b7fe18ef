# development aid (GPU box), round 4: allocator repro with controls; finer anchors none / one launch / three launches; kernel trace of the pre-pass; wide chunks over the instantiations
mkdir -p gpurun_out/r4d
for a in "2000 1 malloc" "2000 4 malloc" "2000 1" "4000 4"; do tools/mallocasync_repro.bin $a >> gpurun_out/r4d/mallocasync.txt 2>&1; done
cat gpurun_out/r4d/mallocasync.txt
export LS=8192,16384,32768,65536,1048576 MEANS=2,4,8,32
for n in 28 30; do
  GIDDY_RUN_FINER_ANCHORS_FROM=0 TAG=none python tools/fine_prepass_sweep.py $n > gpurun_out/r4d/fine_none_$n.txt 2>/dev/null
  TAG=one_launch python tools/fine_prepass_sweep.py $n > gpurun_out/r4d/fine_one_$n.txt 2>/dev/null
  GIDDY_RUN_FINE_ONE_LAUNCH=0 GIDDY_RUN_FINE_FORCE=1 TAG=three_launches python tools/fine_prepass_sweep.py $n > gpurun_out/r4d/fine_three_$n.txt 2>/dev/null
  paste -d'\n' gpurun_out/r4d/fine_none_$n.txt gpurun_out/r4d/fine_one_$n.txt gpurun_out/r4d/fine_three_$n.txt
done
cd /tmp && export TMPDIR=/tmp
LS=65536 MEANS=32,4 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4d/trace_one -- python $GRAFT_REPO_ROOT/tools/fine_prepass_sweep.py 28 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4d/trace_one -name "*kernel_stats.csv" | head -1 | xargs head -12
bash tools/sweep_variants.sh "run_length_encoding|run_position_encoding" 30 default wide2 default wide2 > gpurun_out/r4d/sweep_wide2.txt 2>&1
python tools/compare_sweeps.py gpurun_out/r4d/sweep_wide2.txt default wide2
