TAG=r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
# ---- traces that contain ONLY one way of issuing the kernel (2 x 60 back-to-back forwards: the clock transient behind an idle gap, profiles/r04_clock_transient.txt, weighs ~2 % in the average): the average of the
# fused_seq32_kernel row of each CSV is the duration the bench line's roofline.frac / roofline_native_layout.frac rest on
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt_asissued -- python3 $R/tools/step_kernel_probe.py 256 32 60 > $O/probe_as_issued.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_asissued > $O/${TAG}_seq32_as_issued_only_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/kt_native -- python3 $R/tools/step_kernel_probe.py 256 32 60 native > $O/probe_native.log 2>&1
python3 $R/tools/rocprof_db_stats.py $O/kt_native > $O/${TAG}_seq32_native_only_kernel_stats.csv 2>/dev/null
rm -rf $O/kt_asissued $O/kt_native
head -3 $O/${TAG}_seq32_as_issued_only_kernel_stats.csv $O/${TAG}_seq32_native_only_kernel_stats.csv
