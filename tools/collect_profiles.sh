#!/bin/bash
# Collects, on the GPU box, everything the files of one round under profiles/ come from: the bench lines (C2 with the
# CPU baseline, C3, C4, two ranks on one GPU), the rocprofv3 kernel statistics of the C2 bench command, the hardware
# counter passes of the four kernels' workloads (tools/pmc.sh), launch times, the interactive and time-to-RMSE
# drivers.  Writes gpurun_out/<name>/; copy what is wanted into profiles/ and condense the counter summaries with
# tools/pmc_summary.py.      bash tools/collect_profiles.sh <name>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r03}
rm -rf $O; mkdir -p $O
step() { echo "== $*"; }
step bench; timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
step rocprof-stats; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-stream-line --no-time-to-rmse --no-traffic-pass > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo "rocprof failed"; tail -5 $O/rocprof.err; exit 1; }
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
step c3; timeout -k 10 300 python3 bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/c3.err || { echo "c3 failed"; tail -5 $O/c3.err; exit 1; }
step c4; timeout -k 10 400 python3 bench.py --workload c4 --no-cpu-baseline --steps 2 > $O/bench_c4.json 2> $O/c4.err || { echo "c4 failed"; tail -5 $O/c4.err; exit 1; }
step pmc-c2; bash tools/pmc.sh $O/pmc_c2 --spp 1024 --launches 2 --kernel 5 > $O/pmc_c2.txt 2>&1 || { echo "pmc c2 failed"; tail -5 $O/pmc_c2.txt; exit 1; }
step pmc-c2-stream; bash tools/pmc.sh $O/pmc_c2s --spp 1024 --launches 2 --kernel 5 --mode stream --paths 1 > $O/pmc_c2_stream.txt 2>&1 || { echo "pmc c2 stream failed"; exit 1; }
step pmc-c2-stream-pooled; bash tools/pmc.sh $O/pmc_c2sp --spp 1024 --launches 2 --kernel 5 --mode stream --paths 2 > $O/pmc_c2_stream_pooled.txt 2>&1 || { echo "pmc c2 stream pooled failed"; exit 1; }
step pmc-c2-stream-by-class; bash tools/pmc.sh $O/pmc_c2sc --spp 1024 --launches 2 --kernel 5 --mode stream --paths 3 > $O/pmc_c2_stream_by_class.txt 2>&1 || { echo "pmc c2 stream by class failed"; exit 1; }
step pmc-c3; bash tools/pmc.sh $O/pmc_c3 --spp 128 --launches 4 --kernel 3 --scene outdoor --width 3840 --height 2160 > $O/pmc_c3.txt 2>&1 || { echo "pmc c3 failed"; exit 1; }
step pmc-c4; bash tools/pmc.sh $O/pmc_c4 --spp 2 --launches 1 --kernel 4 --scene soup:100000 > $O/pmc_c4.txt 2>&1 || { echo "pmc c4 failed"; exit 1; }
step pmc-c4-bvh; bash tools/pmc.sh $O/pmc_c4b --spp 2 --launches 2 --kernel 6 --scene soup:100000 > $O/pmc_c4_bvh.txt 2>&1 || { echo "pmc c4 bvh failed"; exit 1; }
step times; for a in "--spp 1024 --kernel 5" "--spp 1024 --kernel 5 --mode stream --paths 1" "--spp 1024 --kernel 5 --mode stream --paths 2" "--spp 1024 --kernel 5 --mode stream --paths 3" "--spp 128 --kernel 3 --scene outdoor --width 3840 --height 2160" "--spp 2 --kernel 4 --scene soup:100000" "--spp 2 --kernel 6 --scene soup:100000" "--spp 16 --kernel 6 --scene soup:100000"; do timeout -k 10 200 python3 tools/prof_run.py --launches 3 $a >> $O/times.txt 2>&1 || { echo "time failed"; exit 1; }; done
step interactive; timeout -k 10 200 python3 tools/interactive.py > $O/interactive.json 2> $O/interactive.err || { echo "interactive failed"; tail -5 $O/interactive.err; exit 1; }
step rmse; timeout -k 10 400 python3 tools/time_to_rmse.py > $O/time_to_rmse.json 2> $O/rmse.err || { echo "rmse failed"; tail -5 $O/rmse.err; exit 1; }
step gloo2; timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/gloo.err || { echo "gloo failed"; tail -5 $O/gloo.err; exit 1; }
for d in pmc_c2 pmc_c2s pmc_c2sp pmc_c2sc pmc_c3 pmc_c4 pmc_c4b; do grep -h 'scene=' $O/$d/pass1.log > $O/$d.time.txt; rm -rf $O/$d/pass*; done; rm -rf $O/stats
cat $O/times.txt
echo done
